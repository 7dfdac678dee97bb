import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
for kind, name in ((11, "8x32 weak"), (12, "9-limb inline"), (13, "9-limb call")):
    best = 0
    for threads in (256 * 256 * 2, 256 * 256 * 4, 256 * 256 * 8):
        iters = 100
        ms = zk.testlib.zkp_microbench(kind, threads, iters)
        rate = threads * iters * 12 / ms * 1e3
        best = max(best, rate)
    print("%-16s %.2f G butterflies/s" % (name, best / 1e9), flush=True)
