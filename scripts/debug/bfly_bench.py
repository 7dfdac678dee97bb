import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
for kind, name, per in ((11, "radix-8 8x32 weak", 12), (12, "radix-8 9-limb inline", 12), (13, "radix-8 9-limb call", 12), (15, "radix-4 8x32 weak", 4), (14, "radix-4 9-limb inline", 4),
                        (16, "radix-4 9-limb 2 wg/CU", 4)):
    best = 0
    for threads in (256 * 256 * 2, 256 * 256 * 4, 256 * 256 * 8):
        iters = 100
        ms = zk.testlib.zkp_microbench(kind, threads, iters)
        rate = threads * iters * per / ms * 1e3
        best = max(best, rate)
    print("%-24s %.2f G butterflies/s" % (name, best / 1e9), flush=True)
