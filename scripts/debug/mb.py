import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for name in sys.argv[1:]:
    lib = C.CDLL(os.path.join(ROOT, "zk-symmetric-crypto_amd", name))
    lib.zkp_microbench.restype = C.c_double
    lib.zkp_microbench.argtypes = [C.c_int, C.c_long, C.c_int]
    for kind, nm, per, iters in ((0,"fp_mul",2,2000),(1,"g1_madd",1,400),(5,"mad",64,2000),(6,"mad+addc",64,2000)):
        best = 0
        for th in (256*256*4, 256*256*8, 256*256*12):
            ms = lib.zkp_microbench(kind, th, iters)
            best = max(best, th*iters*per/ms*1e3)
        print(name, nm, "%.3f G/s" % (best/1e9), flush=True)
