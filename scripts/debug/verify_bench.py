"""wall / device time of VerifyBatch over n proofs of one circuit (GPU box)"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
zk.lib.ZkpAllowUnpinnedKeys(1)
H = zk.synth
alg, n = int(sys.argv[1]), int(sys.argv[2])
r1cs = H.r1cs_bytes(alg)
pk, vk = zk.setup(r1cs, H.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)
rng = H.SplitMix64(77)
insts = [H.make_instance(alg, rng) for _ in range(n)]
reqs = [json.loads(H.request_json(alg, i)) for i in insts]
res = json.loads(zk.prove_batch_raw(json.dumps(reqs).encode()))
vreqs = [{"cipher": reqs[i]["cipher"], "proof": res[i]["proof"], "publicSignals": {"ciphertext": res[i]["ciphertext"], "blocks": [{"nonce": b["nonce"], "counter": b["counter"]} for b in reqs[i]["blocks"]], "input": reqs[i]["input"]}} for i in range(n)]
body = json.dumps(vreqs).encode()
zk.verify_batch_raw(body)
for _ in range(3):
    t0 = time.perf_counter(); out, ms = zk.verify_batch_raw(body, timed=True); dt = time.perf_counter() - t0
    print("alg %d n %d: wall %.1f ms, device %.1f ms, all true %s" % (alg, n, dt * 1e3, ms, all(out)))
t0 = time.perf_counter(); ok = zk.verify_raw(json.dumps(vreqs[0]).encode()); print("single Verify: %.2f ms %s" % ((time.perf_counter() - t0) * 1e3, ok))
