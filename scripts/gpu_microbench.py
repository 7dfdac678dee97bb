"""Raw device throughput of the BN254 building blocks (run on the GPU box).  Prints a table and, with an argument, writes the
best rates as JSON (profiles/<tag>_microbench.json; bench.py reads the file profiles/current.json names)."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
best = {}
for kind, name, per_iter, iters in ((0, "fp_mul", 2, 2000), (1, "g1_mixed_add", 1, 400), (2, "g2_mixed_add", 1, 150), (3, "fp_inv_fermat", 1, 4), (4, "fp_inv_gcd", 1, 16),
                                    (5, "v_mad_u64_u32", 64, 2000), (6, "mad_addc_pair", 64, 2000)):
    for threads in (256 * 256 * 4, 256 * 256 * 8, 256 * 256 * 16):
        ms = zk.lib.zkp_microbench(kind, threads, iters)
        ops = threads * iters * per_iter
        rate = ops / ms * 1e3
        best[name] = max(best.get(name, 0.0), rate)
        print("%-14s threads=%8d iters=%5d  %8.3f ms  %9.3f Gop/s" % (name, threads, iters, ms, rate / 1e9), flush=True)
out = {"fp_mul_per_s": best["fp_mul"], "g1_mixed_add_per_s": best["g1_mixed_add"], "g2_mixed_add_per_s": best["g2_mixed_add"],
       "fp_inv_fermat_per_s": best["fp_inv_fermat"], "fp_inv_gcd_per_s": best["fp_inv_gcd"],
       "v_mad_u64_u32_per_s": best["v_mad_u64_u32"], "mad_addc_pair_per_s": best["mad_addc_pair"],
       # ISA-derived ceilings: a Montgomery product is 128 carry pairs (+ 8 v_mul_lo, ~50 moves / selects); a mixed addition 10 products
       "fp_mul_isa_ceiling_per_s": best["mad_addc_pair"] / 128.0, "g1_mixed_add_isa_ceiling_per_s": best["mad_addc_pair"] / 1280.0,
       "inv_gcd_cost_in_fp_mul": best["fp_mul"] / best["fp_inv_gcd"], "inv_fermat_cost_in_fp_mul": best["fp_mul"] / best["fp_inv_fermat"],
       "note": "zkp_microbench (csrc/test_api.hip): back-to-back dependent operations per thread, best of 4/8/16 wavefronts per SIMD worth of threads"}
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
