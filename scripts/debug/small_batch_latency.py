"""wall time of one ProveBatch(JSON array of n requests) call for small n (median of 9), ChaCha20 or AES-256:
python3 scripts/debug/small_batch_latency.py <alg> [n ...]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zk = importlib.import_module("zk-symmetric-crypto_amd")
alg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
args = [a for a in sys.argv[2:] if a != "--tables"]
ns = [int(x) for x in args] or [1, 2, 4, 8, 16, 32, 64]
with_tables = "--tables" in sys.argv      # a server that has already served a large batch: the comb tables exist, batches >= 16 take them
zk.lib.ZkpAllowUnpinnedKeys(1)
r1cs = zk.synth.r1cs_bytes(alg)
pk, _ = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs)
rng = zk.synth.SplitMix64(5)
reqs = [json.loads(zk.synth.request_json(alg, zk.synth.make_instance(alg, rng))) for _ in range(max(ns))]
if with_tables:
    big = [json.loads(zk.synth.request_json(alg, zk.synth.make_instance(alg, rng))) for _ in range(128)]
    t0 = time.perf_counter(); zk.prove_batch_raw(json.dumps(big).encode())
    print("alg %d: first batch of 128 (table build) %.2f s" % (alg, time.perf_counter() - t0))
for n in ns:
    body = json.dumps(reqs[:n]).encode()
    lat = []
    for _ in range(10):
        t0 = time.perf_counter(); out = zk.prove_batch_raw(body); lat.append((time.perf_counter() - t0) * 1e3)
    res = json.loads(out)
    assert len(res) == n and all("proof" in r for r in res)
    lat = sorted(lat[1:])
    print("alg %d n %3d: %.2f ms per call (min %.2f) = %.0f proofs/s" % (alg, n, lat[len(lat) // 2], lat[0], n / lat[len(lat) // 2] * 1e3))
