"""Polyline resampling to a fixed number of points (engineer/utils/polygons.py:49-129 of the reference, `uniformsample`).

The polyline is OPEN: the closing edge (last point -> first point) is dropped, and so is the last point as a segment start.
More points than requested: the starts of the shortest segments are removed — the first and the last segment count as
shortest (their length is zeroed first), as in the reference.
Fewer: every segment gets round(length / total * n) >= 1 equally spaced points (its start included, its end not); the
rounding surplus is taken from the longest segments, a deficit given to the longest."""
import numpy as np


def uniformsample(points_px2, newpnum):
    pts = np.asarray(points_px2)
    assert pts.ndim == 2 and pts.shape[1] == 2
    nxt = pts[1:]                      # segment ends
    pts = pts[:-1]                     # segment starts
    n = pts.shape[0]
    length = np.sqrt(np.sum((nxt - pts) ** 2, axis=1))
    if n > newpnum:
        length[0] = 0.
        length[-1] = 0.
        keep = np.sort(np.argsort(length)[n - newpnum:])
        out = pts[keep]
        assert out.shape[0] == newpnum
        return out
    order = np.argsort(length)
    count = np.round(length * newpnum / np.sum(length)).astype(np.int32)
    count[count == 0] = 1
    total = int(np.sum(count))
    if total > newpnum:
        surplus, k = total - newpnum, -1
        while surplus > 0:
            e = order[k]
            if count[e] > surplus:
                count[e] -= surplus
                surplus = 0
            else:
                surplus -= count[e] - 1
                count[e] = 1
                k -= 1
    elif total < newpnum:
        count[order[-1]] += newpnum - total
    assert int(np.sum(count)) == newpnum
    pieces = []
    for i in range(n):
        w = np.arange(count[i], dtype=np.float32).reshape(-1, 1) / count[i]
        pieces.append(pts[i:i + 1] * (1 - w) + nxt[i:i + 1] * w)
    return np.concatenate(pieces, axis=0)
