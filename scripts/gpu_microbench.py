"""Raw device throughput of the BN254 building blocks (run on the GPU box)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
for kind, name, per_iter in ((0, "Fp mul", 2), (1, "G1 madd", 1), (2, "G2 madd", 1)):
    for threads in (256 * 256 * 4, 256 * 256 * 8, 256 * 256 * 16):
        iters = 2000 if kind == 0 else (400 if kind == 1 else 150)
        ms = zk.lib.zkp_microbench(kind, threads, iters)
        ops = threads * iters * per_iter
        print("%-8s threads=%8d iters=%5d  %8.3f ms  %8.2f Gop/s" % (name, threads, iters, ms, ops / ms / 1e6), flush=True)
