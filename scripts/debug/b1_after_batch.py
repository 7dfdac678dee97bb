"""single Prove() latency of one circuit alone: fresh, and again after a 2048-request ProveBatch has reserved the large-batch state"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zk = importlib.import_module("zk-symmetric-crypto_amd")
alg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
zk.lib.ZkpAllowUnpinnedKeys(1)
r1cs = zk.synth.r1cs_bytes(alg)
pk, _ = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs)
rng = zk.synth.SplitMix64(5)
rq = zk.synth.request_json(alg, zk.synth.make_instance(alg, rng))
def lat(tag):
    zk.prove_raw(rq)
    v = []
    for _ in range(15):
        t0 = time.perf_counter(); zk.prove_raw(rq); v.append((time.perf_counter() - t0) * 1e3)
    v.sort()
    print("RES", tag, "median %.2f min %.2f" % (v[len(v) // 2], v[0]), flush=True)
lat("fresh")
body = ("[" + ",".join(zk.synth.request_json(alg, zk.synth.make_instance(alg, rng)).decode() for _ in range(2048)) + "]").encode()
zk.prove_batch_raw(body)
lat("after a 2048-request ProveBatch")
