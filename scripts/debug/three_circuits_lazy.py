"""the JS operator's pattern: every circuit is initialised at its first use, here after the circuits before it have filled the device"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zk = importlib.import_module("zk-symmetric-crypto_amd")
H = zk.synth
zk.lib.ZkpAllowUnpinnedKeys(1)
for rnd in range(2):
    for alg in (0, 2, 1):
        if rnd == 0:
            r1cs = H.r1cs_bytes(alg)
            pk, vk = zk.setup(r1cs, H.toxic_from_seed(1), raw=True)
            t0 = time.perf_counter()
            ok = zk.init_algorithm(alg, pk, r1cs)
            print("init alg %d: %s in %.2f s" % (alg, ok, time.perf_counter() - t0), flush=True)
        rng = H.SplitMix64(77 + alg)
        reqs = [json.loads(H.request_json(alg, H.make_instance(alg, rng))) for _ in range(2048)]
        t0 = time.perf_counter()
        out = json.loads(zk.prove_batch_raw(json.dumps(reqs).encode()))
        dt = time.perf_counter() - t0
        good = sum(1 for x in out if isinstance(x, dict) and "proof" in x)
        bad = next((x for x in out if not (isinstance(x, dict) and "proof" in x)), None)
        print("round %d alg %d: %d/%d ok in %.2f s, first failure: %r" % (rnd, alg, good, len(out), dt, bad), flush=True)
