import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zk = importlib.import_module("zk-symmetric-crypto_amd")
for args in [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]:
    print(args, "->", zk.testlib.zkp_debug_tree(*args), flush=True)
