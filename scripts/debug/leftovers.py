import importlib, sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
tl = zk.testlib          # the debug entries live in the test library (same objects, own registry)
tl.ZkpAllowUnpinnedKeys(1)
H = zk.synth
alg, B = int(sys.argv[1]), 1024
r1cs = H.r1cs_bytes(alg)
pk, vk = zk.setup(r1cs, H.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs, lib=tl)
rng = H.SplitMix64(0x5EED0000 + 2)
insts = [H.make_instance(alg, rng) for _ in range(B)]
wits = np.array([zk.request_witness(H.request_json(alg, i))[1] for i in insts], np.uint32)
rb = zk.ResidentBatch(alg, wits, [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts], lib=tl)
rb.step()
out = np.zeros(3, np.int64)
tl.zkp_debug_leftovers(alg, out.ctypes.data)
print("leftovers of the last main-stream wire MSM: sum %d max %d instances %d of %d" % (out[0], out[1], out[2], B))
