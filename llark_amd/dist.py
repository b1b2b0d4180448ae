"""Multi-GPU plumbing of the hot path: one process per GPU, clips sharded by rank, no data-path
collective (clips are independent units -- the reference itself shards its sorted file list with
``--batch_size/--batch_idx``, jukebox/main.py:227-232).  The only collectives are the barrier and the
max-over-ranks of the elapsed time; the backend is RCCL ("nccl") on GPUs and gloo in the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str = "nccl", device: torch.device = None) -> Tuple[int, int, int]:
    rank, world, local = env_rank_world()
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not dist.is_initialized():
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def needs_self_launch(n_procs: int) -> bool:
    """True when N > 1 ranks were requested but this process was NOT started by torch.distributed.run (no WORLD_SIZE
    in the environment): the caller must spawn its own ranks, like the reference's launcher does
    (scripts/training/train_llark.sh:20-22: ``python -m torch.distributed.launch --nproc_per_node=8 -m m2t.train``)."""
    return n_procs > 1 and "WORLD_SIZE" not in os.environ


def self_launch(n_procs: int, script: str, argv: List[str], timeout: Optional[float] = None) -> int:
    """Re-executes ``script argv`` as ``n_procs`` ranks of one node under ``python -m torch.distributed.run`` (one
    process per GPU, rendezvous on 127.0.0.1 and a free port).  torchrun sets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*; each rank binds ``cuda:LOCAL_RANK``.  Returns the launcher's exit code (non-zero if any rank failed)."""
    import subprocess
    import sys

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_procs)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_procs}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    return subprocess.call(cmd, env=env, timeout=timeout)


def gather_floats(value: float, world: int, device="cpu") -> List[float]:
    """Every rank's ``value`` on every rank (one small all_gather; off the timed path)."""
    if world <= 1:
        return [float(value)]
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous shard of ``n_items`` for ``rank`` (sizes differ by at most one; covers every item once)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def clip_indices(rank: int, clips_per_rank: int) -> List[int]:
    """Weak-scaling assignment used by bench.py: rank r processes clips [r*B, (r+1)*B)."""
    return list(range(rank * clips_per_rank, (rank + 1) * clips_per_rank))


def barrier(world: int, cuda: bool = True) -> None:
    if cuda:
        torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    if cuda:
        torch.cuda.synchronize()


def max_over_ranks(value: float, world: int, device="cpu") -> float:
    if world <= 1:
        return float(value)
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj, world: int):
    """Optional final gather of per-rank results on rank 0 (host side; not part of the timed path)."""
    if world <= 1:
        return [obj]
    import torch.distributed as dist

    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def shutdown(world: int) -> None:
    if world > 1:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


class _Exchange:
    """Handle of one in-flight gradient all-reduce; ``wait()`` also unstages a reduced-precision transport buffer."""

    def __init__(self, work, dst: torch.Tensor, staged: Optional[torch.Tensor]):
        self.work, self.dst, self.staged = work, dst, staged

    def wait(self) -> None:
        self.work.wait()
        if self.staged is not None:
            self.dst.copy_(self.staged)                 # bf16 -> fp32 back into the gradient buffer
            self.staged = None


def all_reduce_sum_async(buf: torch.Tensor, comm_dtype: torch.dtype = torch.float32, stage: Optional[torch.Tensor] = None) -> _Exchange:
    """Asynchronous SUM all-reduce of a slice of the flat fp32 gradient buffer.  ``comm_dtype=torch.bfloat16`` sends
    bf16 over the links, which is what the reference's DDP does (m2t/train.py:94-103 casts the model to bf16, so its
    gradient buckets are bf16): half the xGMI bytes of the fp32 exchange.  ``stage``: the caller's PRE-ALLOCATED transport
    buffer for this slice (same number of elements, ``comm_dtype``) -- the trainer owns one the size of the flat gradient, so the
    32 per-layer exchanges of a step neither allocate nor return 0.4 GB blocks to the caching allocator while the backward runs."""
    import torch.distributed as dist

    if comm_dtype == torch.float32:
        return _Exchange(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), buf, None)
    if stage is None:
        staged = buf.to(comm_dtype)
    else:
        assert stage.dtype == comm_dtype and stage.numel() == buf.numel(), "staging slice does not match the gradient slice"
        staged = stage.view(buf.shape)
        staged.copy_(buf)
    return _Exchange(dist.all_reduce(staged, op=dist.ReduceOp.SUM, async_op=True), buf, staged)


# ---------------------------------------------------------------------------------------------
# Model of the gradient exchange on an 8-GPU xGMI node (no such node is reachable from the build; the driver measures it)
# ---------------------------------------------------------------------------------------------
XGMI_LINKS_PER_GPU = 7          # MI355X: one link to each of the 7 peers of an 8-GPU node
XGMI_LINK_GBS = 153.0           # per link and direction (task statement: 7 links x ~153 GB/s per GPU)


def model_allreduce_ms(nbytes: float, world: int = 8, links: int = 1, link_gbs: float = XGMI_LINK_GBS) -> float:
    """Bandwidth term of a SUM all-reduce (reduce-scatter + all-gather): every GPU sends 2 (N - 1) / N of the buffer; with
    `links` of its xGMI links busy at once (1 = a single ring, 7 = every link: seven edge-disjoint rings or the direct
    algorithm on the fully connected node) that takes 2 (N - 1) / N * bytes / (links * link_gbs).  Latency terms ignored."""
    if world <= 1:
        return 0.0
    return 2.0 * (world - 1) / world * nbytes / (links * link_gbs * 1e9) * 1e3


def model_overlapped_exchange(backward_ms: float, layers: int, layer_bytes: float, rest_bytes: float, world: int = 8, links: int = 1,
                              link_gbs: float = XGMI_LINK_GBS):
    """The trainer's schedule replayed on that model: layer i's slice becomes ready when its backward completes (layers are
    differentiated last to first at an even pace over `backward_ms`), slices go over the links one after another, the rest
    (embedding rows, norms, projector) after the last layer.  Returns (total exchange ms, ms of it that ends AFTER the backward =
    what allreduce_grads() would wait for)."""
    t_link = 0.0
    per = model_allreduce_ms(layer_bytes, world, links, link_gbs)
    for k in range(layers):                               # k-th slice to become ready
        ready = backward_ms * (k + 1) / layers
        t_link = max(t_link, ready) + per
    t_link = max(t_link, backward_ms) + model_allreduce_ms(rest_bytes, world, links, link_gbs)
    return per * layers + model_allreduce_ms(rest_bytes, world, links, link_gbs), max(0.0, t_link - backward_ms)
