"""one synchronous ProveBatch(n requests) call at a time, 8 calls; under rocprofv3 --kernel-trace the end of the trace shows one call kernel
by kernel (scripts/debug/timeline_call.py <trace> 0.05 <window ms>):  python3 scripts/debug/bn_trace.py <alg> <n>"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zk = importlib.import_module("zk-symmetric-crypto_amd")
alg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
zk.lib.ZkpAllowUnpinnedKeys(1)
r1cs = zk.synth.r1cs_bytes(alg)
pk, _ = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs)
rng = zk.synth.SplitMix64(5)
if "--tables" in sys.argv:      # a server that has served a large batch: comb tables exist, batches >= 16 take them
    big = json.dumps([json.loads(zk.synth.request_json(alg, zk.synth.make_instance(alg, rng))) for _ in range(128)]).encode()
    zk.prove_batch_raw(big)
body = json.dumps([json.loads(zk.synth.request_json(alg, zk.synth.make_instance(alg, rng))) for _ in range(n)]).encode()
lat = []
for _ in range(8):
    t0 = time.perf_counter(); zk.prove_batch_raw(body); lat.append((time.perf_counter() - t0) * 1e3)
print("ProveBatch(%d) ms:" % n, " ".join("%.2f" % x for x in lat))
