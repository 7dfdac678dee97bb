import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for name in sys.argv[1:]:
    lib = C.CDLL(os.path.join(ROOT, "zk-symmetric-crypto_amd", name))
    lib.zkp_bench_to_coset.restype = C.c_double
    lib.zkp_bench_to_coset.argtypes = [C.c_int, C.c_long, C.c_int]
    for logn in (16, 17):
        print(name, "logn", logn, "to_coset ms:", round(lib.zkp_bench_to_coset(logn, 1024, 5), 3), flush=True)
