"""two host threads, one streamed ProveBatch call each, different ciphers, at the same time (twice); every answer checked, a sample verified"""
import base64, importlib, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
zk = importlib.import_module("zk-symmetric-crypto_amd")
import helpers as H
from oracle import ref
zk.lib.ZkpAllowUnpinnedKeys(1)
algs = (0, 2)
for alg in algs:
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, ref.toxic_from_seed(1), raw=True)
    assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)
work = {}
for alg in algs:
    rng = H.SplitMix64(999 + alg)
    insts = [H.make_instance(alg, rng) for _ in range(3000)]
    work[alg] = (insts, ("[" + ",".join(H.request_json(alg, i).decode() for i in insts) + "]").encode())
for alg in algs: zk.prove_batch_raw(work[alg][1][:1] + work[alg][1][1:])      # warm: tables, buffers
res = {}
def run(alg):
    t0 = time.perf_counter()
    res[alg] = (json.loads(zk.prove_batch_raw(work[alg][1])), time.perf_counter() - t0)
for rnd in range(2):
    th = [threading.Thread(target=run, args=(a,)) for a in algs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    line = {"round": rnd, "wall_s": round(wall, 3)}
    for alg in algs:
        insts = work[alg][0]; out, dt = res[alg]
        good = 0
        for k in range(len(insts)):
            g = isinstance(out[k], dict) and "proof" in out[k]
            if g and (k % 300 == 0 or k == len(insts) - 1):
                o = zk.request_witness(H.request_json(alg, insts[k]))[2]
                g = base64.b64decode(out[k]["ciphertext"]) == o and zk.verify(H.CIPHER[alg], base64.b64decode(out[k]["proof"]), o, list(zip(insts[k]["nonces"], insts[k]["counters"])), insts[k]["data"])
            good += bool(g)
        line["alg%d" % alg] = {"ok": good, "of": len(insts), "s": round(dt, 3)}
    print(json.dumps(line), flush=True)
