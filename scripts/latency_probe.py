import importlib, json, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
zk = importlib.import_module("zk-symmetric-crypto_amd")
zk.lib.ZkpAllowUnpinnedKeys(1)
for alg in (0,2):
    r1cs = zk.synth.r1cs_bytes(alg)
    pk,_ = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
    assert zk.init_algorithm(alg, pk, r1cs)
    rng = zk.synth.SplitMix64(5)
    inst = zk.synth.make_instance(alg, rng)
    rq = zk.synth.request_json(alg, inst)
    for B in (1, 8, 64):
        reqs = json.dumps([json.loads(rq)]*B).encode()
        zk.prove_batch_raw(reqs)
        t0=time.perf_counter(); n=5
        for _ in range(n): zk.prove_batch_raw(reqs)
        dt=(time.perf_counter()-t0)/n
        print("RES alg",alg,"B",B,"ms per call %.2f"%(dt*1e3), "proofs/s %.0f"%(B/dt))
    # phases at B=1
    wit,_ = zk.request_witness(rq)[1:]
    rb = zk.ResidentBatch(alg, np.array([wit], np.uint32), [inst["r"]], [inst["s"]], [inst["randomize"]])
    rb.step(); print("RES phases B=1", rb.step())
