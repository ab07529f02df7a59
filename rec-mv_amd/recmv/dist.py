"""Frame-sharded data parallelism: one process per GPU, RCCL over xGMI (torch.distributed backend "nccl" is
RCCL on ROCm; "gloo" for the CPU tests).

The reference is single-process.  Video frames are independent given the shared
networks, so ranks take disjoint frames of each mini-batch and exchange ONE thing per optimiser step: the
gradients of the shared tensors (garment SDF nets, deformer MLP, colour MLP, per-frame codes / poses /
camera — ~25 MB f32) and, at the explicit-vertex SGD step, the gradients of the MC vertices (deterministic MC
gives every rank the same vertex numbering).  xGMI is point-to-point and the volume is tiny next to a
>100 ms step, so the collective is a single flattened all-reduce (latency-bound; bucketing would only add
launches) — SURVEY.md §5, §8e.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world_size); a no-op single process when WORLD_SIZE is absent or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # RECMV_DIST_BACKEND=gloo lets several ranks share one GPU (functional tests of the N>1 path on a 1-GPU box;
            # RCCL refuses two ranks on the same device)
            backend = os.environ.get("RECMV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("RECMV_SHARE_GPU0") == "1":
            local_rank = 0
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradAllReduce:
    """Average the .grad of a list of tensors over all ranks with one flattened all-reduce.
    Tensors without a gradient contribute zeros (a rank whose frames produced no valid rays must still take
    part in the collective).

    `start(tensors)` packs the gradients on the CURRENT stream and issues the collective asynchronously (RCCL runs it on the
    process group's own stream behind the current stream's work); `finish(handle)` makes the current stream wait for it and
    unpacks.  `__call__` = start + finish.  The loop issues its three exchanges from three streams (explicit vertices: main,
    curve parameters: curve stream, shared gradients: main, in two buckets around the implicit differentiation), so the staging
    buffer is per (stream, size): two exchanges in flight never share one.  Every rank issues them in the same host order."""

    def __init__(self, world_size: int):
        self.world = world_size
        self._flat = {}

    def _buffer(self, n, dev, dt):
        key = (torch.cuda.current_stream(dev).cuda_stream if dev.type == 'cuda' else None, dev, dt)
        pool = self._flat.setdefault(key, {})
        for buf in pool.values():              # (buffers in flight are marked busy until finish())
            if not buf[1] and buf[0].numel() >= n:
                buf[1] = True
                return buf
        buf = [torch.empty(n, dtype=dt, device=dev), True]
        pool[len(pool)] = buf
        return buf

    def start(self, tensors):
        if self.world <= 1:
            return None
        tensors = [t for t in tensors if t.requires_grad]
        if not tensors:
            return None
        n = sum(t.numel() for t in tensors)
        buf = self._buffer(n, tensors[0].device, tensors[0].dtype)
        flat = buf[0][:n]
        off = 0
        for t in tensors:
            k = t.numel()
            if t.grad is None:
                flat[off:off + k].zero_()
            else:
                flat[off:off + k].copy_(t.grad.reshape(-1))
            off += k
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        return (tensors, flat, work, buf)

    def finish(self, handle):
        if handle is None:
            return
        tensors, flat, work, buf = handle
        work.wait()                             # (device tensors: the current stream waits, the host does not)
        flat.div_(self.world)
        off = 0
        for t in tensors:
            k = t.numel()
            if t.grad is None:
                t.grad = flat[off:off + k].view_as(t).clone()
            else:
                t.grad.copy_(flat[off:off + k].view_as(t))
            off += k
        buf[1] = False

    def __call__(self, tensors):
        self.finish(self.start(tensors))


def broadcast_state(tensors, src=0):
    """Make every rank start from rank `src`'s values (initial state)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        with torch.no_grad():
            for t in tensors:
                # receive into a buffer and copy_ in: the collective itself does not bump the parameter's version
                # counter, and every cache keyed on `_version` (normalised weights, transposes, chain descriptors)
                # must see the new values
                buf = t.detach().clone()
                dist.broadcast(buf, src=src)
                t.copy_(buf)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
