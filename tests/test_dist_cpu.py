"""CPU: the N>1 path (rank sharding, barrier, max-over-ranks timing, result gather) with world_size 2 over gloo."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from llark_amd import dist as D

    r, w, _ = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    clips = D.clip_indices(r, 4)                      # weak scaling: 4 clips per rank
    shard = list(D.shard_range(11, r, w))             # strong split of an 11-file list
    D.barrier(w, cuda=False)
    elapsed = D.max_over_ranks(1.0 + 0.5 * r, w)      # slowest rank defines the step time
    # every rank computes a checksum of "its" clips; rank 0 gathers them (host side, off the timed path)
    local = {i: float(i * i) for i in clips}
    gathered = D.gather_objects(local, w)
    q.put((rank, clips, shard, elapsed, gathered if rank == 0 else None))
    D.shutdown(w)


def test_two_rank_sharding_and_timing():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, clips0, shard0, e0, g0), (r1, clips1, shard1, e1, _) = res
    assert clips0 == [0, 1, 2, 3] and clips1 == [4, 5, 6, 7]           # disjoint, rank-ordered
    assert sorted(shard0 + shard1) == list(range(11)) and abs(len(shard0) - len(shard1)) <= 1
    assert e0 == e1 == 1.5                                               # MAX over ranks on every rank
    merged = {}
    for d in g0:
        merged.update(d)
    assert merged == {i: float(i * i) for i in range(8)}                # all 8 clips exactly once


def test_shard_range_properties():
    from llark_amd.dist import shard_range
    for n in (0, 1, 7, 8, 100):
        for w in (1, 2, 3, 8):
            parts = [list(shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from llark_amd import dist as D
    from llark_amd.m2t.train_engine import HipLlamaTrainer

    D.init(backend="gloo")
    tr = object.__new__(HipLlamaTrainer)                   # only the gradient-exchange logic is exercised on CPU
    tr.flat_grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    tr.allreduce_grads(world, bucket_elems=300)            # 4 buckets, async all-reduces
    first = tr.flat_grad.clone()
    # overlapped form: two "layers" are exchanged during the backward (newest layer first), the rest afterwards
    tr.flat_grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    tr._slices = {"layers.0.wqkv": (100, 50), "layers.0.ln2": (340, 10), "layers.1.wqkv": (350, 200), "layers.1.ln2": (690, 10)}
    tr._start_layer_allreduce(1)
    tr._start_layer_allreduce(0)
    tr.allreduce_grads(world, bucket_elems=128)
    assert torch.equal(tr.flat_grad, first), "overlapped exchange must reduce every element exactly once"
    # bf16 on the links (the reference's bf16 DDP buckets): same coverage, values rounded to bf16 before the sum
    tr.grad_comm = torch.bfloat16
    base = torch.arange(1000, dtype=torch.float32) * 1.001
    tr.flat_grad = base * (rank + 1)
    tr._start_layer_allreduce(1)
    tr.allreduce_grads(world, bucket_elems=128)
    want = sum((base * (r + 1)).to(torch.bfloat16) for r in range(world)).float()
    assert tr.flat_grad.dtype == torch.float32 and torch.equal(tr.flat_grad, want), "bf16 exchange: every element once, fp32 buffer"
    # the transport buffer is allocated ONCE (the size of the flat gradient) and sliced per exchange: a second step reuses it
    stage = tr._flat_stage
    assert stage.dtype == torch.bfloat16 and stage.numel() == tr.flat_grad.numel()
    tr.flat_grad = base * (rank + 1)
    tr._start_layer_allreduce(0)
    tr._start_layer_allreduce(1)
    tr.allreduce_grads(world, bucket_elems=64)
    assert tr._flat_stage is stage and torch.equal(tr.flat_grad, want)
    tr.grad_comm = torch.float32
    tr.flat_grad = first.clone()
    # by VALUE (numpy pickles its bytes): a torch tensor travels through a multiprocessing queue as a shared-memory handle, and this worker
    # may exit -- unlinking the segment -- before the parent has mapped it (seen once in round 6 as a FileNotFoundError in the parent)
    q.put((rank, tr.flat_grad.clone().numpy()))
    D.shutdown(world)


def test_gradient_allreduce_two_ranks():
    """The one exchange step of the path (config 4): bucketed SUM all-reduce of the flat gradient buffer."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = torch.arange(1000, dtype=torch.float32) * 3          # (1 + 2) x
    for _, gsum in res:
        assert torch.equal(torch.from_numpy(gsum), expect)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` started plainly (no WORLD_SIZE) must spawn its own two ranks under
    torch.distributed.run (reference launcher: scripts/training/train_llark.sh:20-22).  The
    dist-check stage runs bench.py's launcher / rendezvous / barrier / max-over-ranks / gather over gloo on CPU."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stages", "dist-check",
                          "--backend", "gloo", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                               # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 2
    assert d["clips_by_rank"] == [[0, 1], [2, 3]]
    assert len(d["per_rank_ms_per_step"]) == 2 and d["per_rank_ms_per_step"][1] > d["per_rank_ms_per_step"][0]
    assert d["ms_per_step"] >= max(d["per_rank_ms_per_step"]) - 0.5  # the slowest rank defines the step


def test_bench_rejects_world_size_mismatch():
    """Started as ONE rank of a 1-rank world but asked for --gpus 2: a clear error, not a hang."""
    import subprocess

    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stages", "dist-check",
                          "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
