"""Split mode (the north_star's "RCCL over xGMI only for the final bucket reduction"): ONE proof over several
GPUs.  Every rank keeps a slice of every base set of the key, multiplies its slices, the ranks all_gather the five partial sums per
proof and each finishes the proof.  Here: world size 2, both ranks on the box's single GPU, the exchange over gloo (on a multi-GPU node
the same code runs with one GPU per rank and backend nccl = RCCL, scripts/split_latency.py).  The proofs must be byte-identical to the
unsplit prover's = the oracle's."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

_RANK_SCRIPT = r"""
import importlib, json, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
zk = importlib.import_module("zk-symmetric-crypto_amd")
import helpers as H
from oracle import ref
zk.lib.ZkpAllowUnpinnedKeys(1)
assert zk.lib.ZkpSetDevice(0) == 0
assert zk.lib.ZkpSetSplit(rank, world) == 0
out = {{}}
for alg, batch in ((0, 3), (1, 2)):
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, ref.toxic_from_seed(1), raw=True)
    assert zk.init_algorithm(alg, pk, r1cs)
    rng = H.SplitMix64(4711 + alg)                      # the SAME requests and blinding on every rank
    insts = [H.make_instance(alg, rng) for _ in range(batch)]
    wits = np.array([zk.request_witness(H.request_json(alg, i))[1] for i in insts], np.uint32)
    rb = zk.ResidentBatch(alg, wits, [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
    mine = torch.from_numpy(rb.split_partials())
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)                           # the only exchange: world x batch x 384 bytes
    rb.split_finish(torch.stack(every).numpy())
    proofs, errs = rb.fetch()
    assert not errs.any()
    ser = [zk.proof_serialize(proofs[k], alg != 0).hex() for k in range(batch)]
    if rank == 0:
        exp = [H.oracle_proof_bytes(alg, i)[0].hex() for i in insts]
        out[str(alg)] = dict(match_oracle=(ser == exp), partial_differs=bool((every[0] != every[1]).any().item()))
    gathered = [None] * world
    dist.all_gather_object(gathered, ser)
    if rank == 0:
        out[str(alg)]["ranks_agree"] = all(g == ser for g in gathered)
if rank == 0:
    print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_one_proof_split_over_two_ranks_matches_the_unsplit_prover():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(30500 + os.getpid() % 2000), WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, "-c", _RANK_SCRIPT.format(root=H.ROOT)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    for alg in ("0", "1"):
        assert res[alg]["match_oracle"], res          # byte-identical to the oracle's (= the unsplit prover's) proofs
        assert res[alg]["ranks_agree"], res           # every rank finishes the same proof
        assert res[alg]["partial_differs"], res       # and the ranks really contributed different partial sums
