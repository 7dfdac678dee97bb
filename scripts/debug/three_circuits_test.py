
import base64, importlib, json, os, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.path.join('/root/repo', "tests"))
zk = importlib.import_module("zk-symmetric-crypto_amd")
import helpers as H
from oracle import ref
zk.lib.ZkpAllowUnpinnedKeys(1)
def comb(alg):
    info = np.zeros(8, np.int64)
    zk.lib.zkp_quotient_msm_info(alg, info.ctypes.data)
    return int(info[0])
order = (0, 2, 1)                       # ChaCha20, AES-256, then AES-128 arrives on a full device
for alg in order:
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, ref.toxic_from_seed(1), raw=True)
    assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)
out = {"ok": {}, "comb_before": {}, "comb_after": {}}
for rnd in range(2):
    for alg in order:
        rng = H.SplitMix64(4242 + alg)
        insts = [H.make_instance(alg, rng) for _ in range(2048)]
        res = json.loads(zk.prove_batch_raw(("[" + ",".join(H.request_json(alg, i).decode() for i in insts) + "]").encode()))
        good = 0
        for k in range(len(insts)):
            g = isinstance(res[k], dict) and "proof" in res[k]
            if g and (k % 256 == 0 or k == len(insts) - 1):
                o = zk.request_witness(H.request_json(alg, insts[k]))[2]
                g = base64.b64decode(res[k]["ciphertext"]) == o and zk.verify(H.CIPHER[alg], base64.b64decode(res[k]["proof"]), o, list(zip(insts[k]["nonces"], insts[k]["counters"])), insts[k]["data"])
            good += bool(g)
        out["ok"]["%d.%d" % (rnd, alg)] = good
        if rnd == 0 and alg == 2: out["comb_before"] = {str(a): comb(a) for a in (0, 2)}
out["comb_after"] = {str(a): comb(a) for a in order}
print(json.dumps(out))
