"""Small HOCON-subset reader with pyhocon's accessor API.

The reference parses its configs with pyhocon (train.py:82; absent in this image).  The config schema of
configs/**/*.conf uses only: nested `{}` objects, `key = value` / `key : value` / `key { ... }`, lists
`[ ... ]` with newline or comma separators, quoted or bare scalars and `#` / `//` comments.
This reader parses exactly that and exposes get_int / get_float / get_string / get_bool / get_config /
get_list / get, dotted paths and `'a.b' in conf`, coercing quoted numbers ("1." -> 1.0) like pyhocon's
typed getters.  Duplicate keys: the last one wins, objects merge (HOCON semantics).
"""
from __future__ import annotations

import re


class ConfigException(Exception):
    pass


class ConfigMissingException(ConfigException, KeyError):
    pass


_TOKEN = re.compile(r'''
      (?P<ws>[ \t\r]+)
    | (?P<comment>(\#|//)[^\n]*)
    | (?P<nl>\n)
    | (?P<str>"(?:\\.|[^"\\])*")
    | (?P<punct>[{}\[\],=:])
    | (?P<bare>[^\s{}\[\],=:"\#]+)
''', re.X)


def _tokenize(text):
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ConfigException(f"cannot tokenize at offset {pos}: {text[pos:pos + 20]!r}")
        pos = m.end()
        kind = m.lastgroup
        if kind in ("ws", "comment"):
            continue
        out.append((kind, m.group(kind)))
    return out


def _scalar(tok_kind, tok):
    if tok_kind == "str":
        return bytes(tok[1:-1], "utf-8").decode("unicode_escape")
    low = tok.lower()
    if low in ("true", "yes", "on"):
        return True
    if low in ("false", "no", "off"):
        return False
    if low == "null":
        return None
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


class ConfigTree(dict):
    # ------------------------------------------------------------------ access
    def _lookup(self, key):
        node = self
        for part in key.split("."):
            if not isinstance(node, dict) or part not in dict.keys(node):
                raise ConfigMissingException(f"No configuration setting found for key {key}")
            node = dict.__getitem__(node, part)
        return node

    def __contains__(self, key):
        try:
            self._lookup(key)
            return True
        except ConfigMissingException:
            return False

    def __getitem__(self, key):
        return self._lookup(key)

    def get(self, key, default=ConfigMissingException):
        try:
            return self._lookup(key)
        except ConfigMissingException:
            if default is ConfigMissingException:
                raise
            return default

    def get_string(self, key, default=ConfigMissingException):
        v = self.get(key, default)
        if isinstance(v, bool):
            return "true" if v else "false"
        return None if v is None else str(v)

    def get_int(self, key, default=ConfigMissingException):
        v = self.get(key, default)
        return None if v is None else int(float(v)) if isinstance(v, str) else int(v)

    def get_float(self, key, default=ConfigMissingException):
        v = self.get(key, default)
        return None if v is None else float(v)

    def get_bool(self, key, default=ConfigMissingException):
        v = self.get(key, default)
        if isinstance(v, str):
            low = v.lower()
            if low in ("true", "yes", "on"):
                return True
            if low in ("false", "no", "off"):
                return False
            raise ConfigException(f"{key} is not a boolean: {v!r}")
        return None if v is None else bool(v)

    def get_list(self, key, default=ConfigMissingException):
        v = self.get(key, default)
        if v is not None and not isinstance(v, list):
            raise ConfigException(f"{key} is not a list")
        return v

    def get_config(self, key, default=ConfigMissingException):
        v = self.get(key, default)
        if v is not None and not isinstance(v, ConfigTree):
            raise ConfigException(f"{key} is not an object")
        return v

    def put(self, key, value):
        node = self
        parts = key.split(".")
        for part in parts[:-1]:
            nxt = dict.get(node, part)
            if not isinstance(nxt, ConfigTree):
                nxt = ConfigTree()
                dict.__setitem__(node, part, nxt)
            node = nxt
        old = dict.get(node, parts[-1])
        if isinstance(old, ConfigTree) and isinstance(value, ConfigTree):
            for k, v in dict.items(value):
                old.put(k, v)
        else:
            dict.__setitem__(node, parts[-1], value)


class _Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def skip_sep(self):
        while self.peek()[0] == "nl" or self.peek() == ("punct", ","):
            self.i += 1

    def parse_object(self, braced):
        tree = ConfigTree()
        while True:
            self.skip_sep()
            kind, tok = self.peek()
            if kind is None:
                if braced:
                    raise ConfigException("unexpected end of input inside '{'")
                return tree
            if (kind, tok) == ("punct", "}"):
                if not braced:
                    raise ConfigException("unexpected '}'")
                self.i += 1
                return tree
            if kind not in ("bare", "str"):
                raise ConfigException(f"expected a key, got {tok!r}")
            self.i += 1
            key = _scalar(kind, tok) if kind == "str" else tok
            kind2, tok2 = self.peek()
            if (kind2, tok2) == ("punct", "{"):
                self.i += 1
                value = self.parse_object(True)
            elif (kind2, tok2) in (("punct", "="), ("punct", ":")):
                self.i += 1
                value = self.parse_value()
            else:
                raise ConfigException(f"expected '=', ':' or '{{' after key {key!r}, got {tok2!r}")
            tree.put(str(key), value)

    def parse_value(self):
        while self.peek()[0] == "nl":
            self.i += 1
        kind, tok = self.next()
        if (kind, tok) == ("punct", "{"):
            return self.parse_object(True)
        if (kind, tok) == ("punct", "["):
            items = []
            while True:
                self.skip_sep()
                if self.peek() == ("punct", "]"):
                    self.i += 1
                    return items
                if self.peek()[0] is None:
                    raise ConfigException("unexpected end of input inside '['")
                items.append(self.parse_value())
        if kind in ("bare", "str"):
            value = _scalar(kind, tok)
            # unquoted multi-word strings: concatenate bare tokens up to the end of the line
            while kind == "bare" and self.peek()[0] == "bare":
                value = f"{value} {self.next()[1]}"
            return value
        raise ConfigException(f"unexpected token {tok!r}")


class ConfigFactory:
    @staticmethod
    def parse_string(text):
        p = _Parser(_tokenize(text))
        while p.peek()[0] == "nl":
            p.i += 1
        if p.peek() == ("punct", "{"):
            p.i += 1
            return p.parse_object(True)
        return p.parse_object(False)

    @staticmethod
    def parse_file(path):
        with open(path, "r") as f:
            return ConfigFactory.parse_string(f.read())
