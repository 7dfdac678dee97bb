"""Null-device feed benchmark of the streamed ProveBatch's host stages (VERDICT r3 task 7: one feed has to keep eight GPUs busy,
8 x 11.5 k ChaCha20 requests/s, with headroom).  No GPU is touched: the JSON array is cut into elements, every 1 024-chunk is decoded
(single-pass scanner, DOM reader for anything unusual), ciphered and packed into its 320-byte witness records, gets its blinding
scalars, and is answered (proof bytes -> base64 -> JSON) exactly as cabi.cpp::prove_batch_json's first and third stage do
(zkp_host_stage_bench in the test library).  Prints one JSON line per cipher; usage: python scripts/host_feed_bench.py [requests] [iters]"""
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
H = zk.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for alg in (0, 2):
    rng = H.SplitMix64(0xFEED + alg)
    reqs = [json.loads(H.request_json(alg, H.make_instance(alg, rng))) for _ in range(n)]
    body = json.dumps(reqs, separators=(",", ":")).encode()               # JSON.stringify's spacing
    out = np.zeros(4)
    b = np.frombuffer(body, np.uint8).copy()
    assert zk.testlib.zkp_host_stage_bench(b.ctypes.data, len(body), iters, out.ctypes.data) == 0
    print(json.dumps({"cipher": H.CIPHER[alg], "requests": n, "iters": iters, "host_threads": min(16, os.cpu_count() or 1), "bytes_per_request": len(body) // n,
                      "decode_stage_requests_per_s": round(out[0]), "serialise_stage_requests_per_s": round(out[1]), "scanner_took": int(out[2]), "records": int(out[3]),
                      "device_record_bytes": 320, "what": "zkp_host_stage_bench: cut + decode + cipher + pack + blinding scalars per 1024-chunk; proof bytes -> base64 -> JSON"}), flush=True)
