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

import contextlib
import datetime
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
        # Timeouts (README "multi-GPU"): the training collectives wait RECMV_DIST_TIMEOUT_S (default 1800 s: a rank that crashed or
        # fell out of step takes the job down within half an hour, while a rank-0-only stage inside the loop — a mesh or
        # visualisation dump — still fits).  The START-UP rendezvous waits RECMV_STARTUP_TIMEOUT_S (default 7200 s): a first run's
        # start-up stage (train.py: skinner bake, SDF pre-fit, feature-line registration) takes rank 0 far longer while the other
        # ranks wait, and it has its own host-side gloo group (startup_gate below).
        kw = {"timeout": datetime.timedelta(seconds=float(os.environ.get("RECMV_DIST_TIMEOUT_S", "1800")))}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        global _gate_group
        try:
            _gate_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(
                seconds=float(os.environ.get("RECMV_STARTUP_TIMEOUT_S", "7200"))))
        except Exception as exc:      # noqa: BLE001 — gloo cannot resolve a host name / interface where RCCL alone works (set
            # GLOO_SOCKET_IFNAME): fall back to the default group for the gate — it then waits RECMV_DIST_TIMEOUT_S only
            import warnings
            warnings.warn("recmv.dist: no gloo side group for the start-up gate (%r); the gate uses the training group and its "
                          "timeout — raise RECMV_DIST_TIMEOUT_S for a long start-up stage, or set GLOO_SOCKET_IFNAME" % (exc,))
            _gate_group = None
    return rank, local_rank, world


_gate_group = None          # gloo group of all ranks for the start-up rendezvous (two-hour timeout), made by init_distributed


def startup_gate(ok: bool = True, what: str = "start-up stage"):
    """Rendezvous behind a stage that only rank 0 ran: every rank learns whether it succeeded.  A plain barrier leaves the waiting
    ranks hanging until the collective timeout when rank 0 raised; here rank 0 reports its outcome (MIN over one flag per rank) and
    every rank leaves with the same SystemExit when it failed.  Called a second time behind the part EVERY rank runs (each rank
    reporting its own outcome) it also catches a non-zero rank that failed after the first gate."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        if not ok:
            raise SystemExit(f"{what} failed")
        return
    if _gate_group is not None:                  # the long-timeout host group: a flag per rank, no device involved
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=_gate_group)
    else:                                        # (a process group somebody else initialised)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        raise SystemExit(f"{what} failed on another rank" if ok else f"{what} failed")


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
        try:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        except BaseException:
            buf[1] = False
            raise
        dev = tensors[0].device
        stream = torch.cuda.current_stream(dev) if dev.type == 'cuda' else None
        return (tensors, flat, work, buf, stream)

    def finish(self, handle):
        if handle is None:
            return
        tensors, flat, work, buf, stream = handle
        # the staging buffer belongs to the stream start() packed it on: wait, scale and unpack there whatever stream the caller
        # is on now, and free the buffer also when something below raises
        try:
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                work.wait()                     # (device tensors: the stream waits, the host does not)
                flat.div_(self.world)
                off = 0
                for t in tensors:
                    k = t.numel()
                    if t.grad is None:
                        t.grad = flat[off:off + k].view_as(t).clone()
                    else:
                        t.grad.copy_(flat[off:off + k].view_as(t))
                    off += k
            if stream is not None and torch.cuda.current_stream(stream.device) != stream:
                torch.cuda.current_stream(stream.device).wait_stream(stream)      # the caller consumes the gradients where it is
        finally:
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
