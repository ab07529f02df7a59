"""The name tables of the reference's utils/constant.py the loop and the data path are keyed on, dumped from the reference module
itself (imported from /root/reference) -> tests/golden/constant_tables.json; tests/test_entry_points.py compares
recmv/utils/constant.py with it.

    python tests/golden/make_golden_constants.py
"""
import importlib.util
import json
from pathlib import Path

HERE = Path(__file__).resolve().parent
TABLES = ('TEMPLATE_GARMENT', 'FL_INFOS', 'FL_EXTRACT', 'GARMENT_FL_MATCH', 'ZBUF_THRESHOLD', 'CURVE_AWARE', 'ATR_PARSING',
          'INI_FL_SCALE')

if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location('ref_constant', '/root/reference/utils/constant.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {name: getattr(mod, name) for name in TABLES}
    (HERE / 'constant_tables.json').write_text(json.dumps(out, indent=1, sort_keys=True) + '\n')
    print({k: len(v) for k, v in out.items()})
