"""three circuits in one process, a 1024-batch ProveBatch on each in turn (twice): memory budget behaviour"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zk = importlib.import_module("zk-symmetric-crypto_amd")
H = zk.synth
zk.lib.ZkpAllowUnpinnedKeys(1)
vks = {}
for alg in (0, 2, 1):
    r1cs = H.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, H.toxic_from_seed(1), raw=True)
    assert zk.init_algorithm(alg, pk, r1cs)
    vks[alg] = vk
for rnd in range(2):
    for alg in (0, 2, 1):
        rng = H.SplitMix64(77 + alg)
        reqs = [json.loads(H.request_json(alg, H.make_instance(alg, rng))) for _ in range(2048)]
        t0 = time.perf_counter()
        out = json.loads(zk.prove_batch_raw(json.dumps(reqs).encode()))
        dt = time.perf_counter() - t0
        ok = sum(1 for x in out if isinstance(x, dict) and "proof" in x)
        bad = next((x for x in out if not (isinstance(x, dict) and "proof" in x)), None)
        print("round %d alg %d: %d/%d ok in %.2f s (%.0f proofs/s) first failure: %r" % (rnd, alg, ok, len(out), dt, len(out) / dt, bad), flush=True)
