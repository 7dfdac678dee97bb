import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
zk = importlib.import_module("zk-symmetric-crypto_amd")
from oracle import ref as oracle
oracle.lib(); oracle.set_threads(8)
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
def rand_fr(rng, n):
    return np.frombuffer(b"".join((int.from_bytes(rng.bytes(32), "little") % R).to_bytes(32, "little") for _ in range(n)), np.uint8).reshape(n, 32).copy()
n, batch, c, special = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(n * 31 + c)
pts = oracle.g1_fixed_base(rand_fr(rng, n))
sc = np.stack([rand_fr(rng, n) for _ in range(batch)])
if special & 1: pts[5] = 0
if special & 2: pts[9] = pts[8]; sc[0, 9] = sc[0, 8]
if special & 4:
    pts[9] = pts[8]
    sc[1, 9] = np.frombuffer(((R - int.from_bytes(sc[1, 8].tobytes(), "little")) % R).to_bytes(32, "little"), np.uint8)
exp = np.stack([oracle.msm_g1(pts, sc[b]) for b in range(batch)])
got = np.zeros((batch, 64), np.uint8)
rc = zk.testlib.zkp_msm_g1_dense(n, batch, pts.ctypes.data, sc.ctypes.data, got.ctypes.data, c)
print(sys.argv[1:], "tree=%s piece=%s" % (os.environ.get("ZKPROVE_MSM_TREE"), os.environ.get("ZKPROVE_MSM_PIECE")), "rc", rc, "match per instance:", [bool(np.array_equal(got[b], exp[b])) for b in range(batch)], flush=True)
