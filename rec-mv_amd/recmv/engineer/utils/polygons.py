"""Polyline resampling to a fixed number of points (engineer/utils/polygons.py of the reference): `uniformsample` (:49-129,
2-D annotations of the data path) and `uniformsample3d` (:132-228, the closed 3-D template curves `align_fl` samples), with
the farthest-point subsampling the latter uses (:12-47).

`uniformsample`:

The polyline is OPEN: the closing edge (last point -> first point) is dropped, and so is the last point as a segment start.
More points than requested: the starts of the shortest segments are removed — the first and the last segment count as
shortest (their length is zeroed first), as in the reference.
Fewer: every segment gets round(length / total * n) >= 1 equally spaced points (its start included, its end not); the
rounding surplus is taken from the longest segments, a deficit given to the longest."""
import numpy as np
import torch


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


def farthest_point_sample(xyz, npoint):
    """:12-47 — indices [B,npoint] of a farthest-point subsample of xyz [B,N,3]; deterministic: the first pick is the point
    farthest from the centroid, ties go to the lowest index."""
    B, N, _ = xyz.shape
    picks = torch.zeros(B, npoint, dtype=torch.long, device=xyz.device)
    nearest = torch.full((B, N), 1e10, device=xyz.device)
    rows = torch.arange(B, dtype=torch.long, device=xyz.device)
    centre = (torch.sum(xyz, 1) / N).view(B, 1, 3)
    far = torch.max(torch.sum((xyz - centre) ** 2, -1), 1)[1]
    for i in range(npoint):
        picks[:, i] = far
        d = torch.sum((xyz - xyz[rows, far, :].view(B, 1, 3)) ** 2, -1)
        nearest = torch.where(d < nearest, d, nearest)
        far = torch.max(nearest, -1)[1]
    return picks


def uniformsample3d(points_px3, newpnum):
    """:132-228 — a CLOSED 3-D polyline resampled to about `newpnum` points.
    More points than requested: a farthest-point subsample in the original order, WITHOUT its last point (newpnum - 1 points).
    Fewer: every edge (the closing one included) gets round(length / total * n) >= 1 equally spaced points; the last point is
    dropped when the signed coordinate sum of (first - last) is below 1e-6 (the reference's closing test, kept as it is)."""
    pts = np.asarray(points_px3)
    n, c = pts.shape
    assert c == 3
    nxt = pts[(np.arange(n, dtype=np.int32) + 1) % n]
    length = np.sqrt(np.sum((nxt - pts) ** 2, axis=1))
    if n > newpnum:
        dense = torch.from_numpy(pts).float()
        keep = farthest_point_sample(dense[None], newpnum)[0].sort().values
        out = dense[keep].numpy()
        assert out.shape[0] == newpnum
        return out[:-1]
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
    out = np.concatenate(pieces, axis=0)
    return out[:-1] if sum(out[0] - out[-1]) < 1e-6 else out
