#!/usr/bin/env python3
"""rocprofv3 target: repeated single-proof passes (B = 1) to see which kernels make up the latency of one Prove() call."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
zk = importlib.import_module("zk-symmetric-crypto_amd")
zk.lib.ZkpAllowUnpinnedKeys(1)
alg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r1cs = zk.synth.r1cs_bytes(alg)
pk, _ = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs)
rng = zk.synth.SplitMix64(5)
insts = [zk.synth.make_instance(alg, rng) for _ in range(B)]
wits = np.array([zk.request_witness(zk.synth.request_json(alg, i))[1] for i in insts], np.uint32)
rb = zk.ResidentBatch(alg, wits, [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
for _ in range(20):
    t = rb.step()
print("PHASES", t)
