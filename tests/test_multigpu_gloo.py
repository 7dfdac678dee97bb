"""N > 1 path on CPU (gloo, world_size 2): batches shard as contiguous slices with no data-path collective;
only the fixed-size proof records / timing are gathered."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    zk = importlib.import_module("zk-symmetric-crypto_amd")
    import helpers as H
    lo, hi = zk.shard_range(total, rank, world)
    # every rank derives the same global request list, proves (here: assigns witnesses for) its own slice only
    rng = H.SplitMix64(99)
    insts = [H.make_instance(0, rng) for _ in range(total)]
    recs = torch.zeros(total, 8, dtype=torch.int64)
    for i in range(lo, hi):
        _, w, out = zk.request_witness(H.request_json(0, insts[i]))
        recs[i, 0] = i
        recs[i, 1] = int(np.asarray(w, np.int64).sum())
        recs[i, 2] = int.from_bytes(out[:7], "little")
    gathered = [torch.zeros_like(recs) for _ in range(world)]
    dist.all_gather(gathered, recs)                  # end-of-run gather of fixed-size records; not on the critical path
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)         # bench.py: max over ranks
    merged = sum(gathered)
    if rank == 0:
        q.put((lo, hi, merged[:, 0].tolist(), float(t.item()), merged[:, 1].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, total = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    lo, hi, idx, tmax, sums = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (lo, hi) == (0, 4)
    assert idx == list(range(total))          # slices partition the batch: every unit proved exactly once
    assert tmax == 1.5
    assert all(s > 0 for s in sums)


def test_bench_launcher_starts_n_ranks_and_gathers():
    """`python bench.py --gpus 2` with no rendezvous in the environment must itself start two ranks (children of a parent that never
    touches a GPU), and rank 0 must see both ranks' fixed-size records after the gather.  --launcher-selftest runs exactly that
    plumbing on CPU (gloo): nothing is proved or timed."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest",
                          "--master-port", str(31000 + os.getpid() % 2000)], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                      # exactly one JSON line reaches the caller
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launched_by_bench"] is True
    assert out["gathered_records"] == 8 and out["rank_stamps"] == [1, 2]      # both ranks' slices arrived, in rank order
    assert out["max_time"] == 2.0                                             # max over ranks


def test_shard_range_partitions():
    import importlib
    sys.path.insert(0, ROOT)
    zk = importlib.import_module("zk-symmetric-crypto_amd")
    for total in (0, 1, 7, 1024, 65536):
        for world in (1, 2, 3, 8):
            spans = [zk.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
