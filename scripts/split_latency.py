#!/usr/bin/env python3
"""Latency of ONE proof split over the GPUs of a node (split mode).  Launch with torchrun, one rank per GPU:
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/split_latency.py [alg] [batch] [iters]
RCCL all_gather of the partial sums (world x batch x 384 bytes).  Prints one JSON line on rank 0."""
import importlib, json, os, sys, time
import numpy as np
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
alg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
backend = os.environ.get("SPLIT_BACKEND", "nccl")
if backend == "nccl":
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
else:
    dist.init_process_group(backend)
real_stdout = os.dup(1); os.dup2(2, 1)
zk = importlib.import_module("zk-symmetric-crypto_amd")
zk.lib.ZkpAllowUnpinnedKeys(1)
assert zk.lib.ZkpSetDevice(local if backend == "nccl" else 0) == 0
assert zk.lib.ZkpSetSplit(rank, world) == 0
H = zk.synth
r1cs = H.r1cs_bytes(alg)
pk, vk = zk.setup(r1cs, H.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs)
rng = H.SplitMix64(99)
insts = [H.make_instance(alg, rng) for _ in range(batch)]
wits = np.array([zk.request_witness(H.request_json(alg, i))[1] for i in insts], np.uint32)
rb = zk.ResidentBatch(alg, wits, [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
def once():
    mine = torch.from_numpy(rb.split_partials())
    if backend == "nccl":
        g = torch.empty((world,) + tuple(mine.shape), dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(g, mine.cuda())
        g = g.cpu()
    else:
        lst = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(lst, mine)
        g = torch.stack(lst)
    rb.split_finish(g.numpy())
once(); once()
dist.barrier()
t0 = time.perf_counter()
for _ in range(iters):
    once()
dt = (time.perf_counter() - t0) / iters
proofs, errs = rb.fetch()
if rank == 0:
    assert zk.init_verifier(alg, vk)
    ok = zk.verify(H.CIPHER[alg], zk.proof_serialize(proofs[0], alg != 0), zk.request_witness(H.request_json(alg, insts[0]))[2],
                   list(zip(insts[0]["nonces"], insts[0]["counters"])), insts[0]["data"])
    os.write(real_stdout, (json.dumps({"metric": "latency of one split Prove (ms)", "alg": alg, "batch": batch, "n_gpus": world, "backend": backend,
                                       "ms_per_call": dt * 1e3, "proof_verifies": bool(ok), "solver_ok": int((errs == 0).sum())}) + "\n").encode())
dist.barrier()
dist.destroy_process_group()
