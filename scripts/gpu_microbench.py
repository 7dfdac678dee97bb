"""Raw device throughput of the BN254 building blocks (run on the GPU box).  Prints a table and, with an argument, writes the
best rates as JSON (profiles/<tag>_microbench.json; bench.py reads the file profiles/current.json names)."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-symmetric-crypto_amd")
best = {}
for kind, name, per_iter, iters in ((0, "fp_mul", 2, 2000), (1, "g1_mixed_add", 1, 400), (2, "g2_mixed_add", 1, 150), (3, "fp_inv_fermat", 1, 4), (4, "fp_inv_gcd", 1, 16),
                                    (5, "v_mad_u64_u32", 64, 2000), (6, "mad_addc_pair", 64, 2000), (7, "fp9_mul", 2, 2000), (8, "g1_mixed_add_9limb", 1, 400),
                                    (9, "g2_mixed_add_9limb", 1, 150), (10, "v_mad_i64_i32", 64, 2000)):
    for threads in (256 * 256 * 4, 256 * 256 * 8, 256 * 256 * 16):
        ms = zk.testlib.zkp_microbench(kind, threads, iters)
        ops = threads * iters * per_iter
        rate = ops / ms * 1e3
        best[name] = max(best.get(name, 0.0), rate)
        print("%-14s threads=%8d iters=%5d  %8.3f ms  %9.3f Gop/s" % (name, threads, iters, ms, rate / 1e9), flush=True)
out = {"fp_mul_per_s": best["fp9_mul"], "g1_mixed_add_per_s": best["g1_mixed_add_9limb"], "g2_mixed_add_per_s": best["g2_mixed_add_9limb"],
       "fp_mul_words_per_s": best["fp_mul"], "g1_mixed_add_words_per_s": best["g1_mixed_add"], "g2_mixed_add_words_per_s": best["g2_mixed_add"],
       "v_mad_i64_i32_per_s": best["v_mad_i64_i32"],
       "fp_inv_fermat_per_s": best["fp_inv_fermat"], "fp_inv_gcd_per_s": best["fp_inv_gcd"],
       "v_mad_u64_u32_per_s": best["v_mad_u64_u32"], "mad_addc_pair_per_s": best["mad_addc_pair"],
       # ISA-derived ceilings of the production (9-limb, field9.cuh) path: a product is 162 v_mad_i64_i32 (+ 43 shifts / masks / v_mul_lo),
       # a G1 mixed addition 1 467 of them (6 products, 2 squarings, one fused pair); of the 8 x 32 path: 128 carry pairs / 1 280
       "fp_mul_isa_ceiling_per_s": best["v_mad_i64_i32"] / 162.0, "g1_mixed_add_isa_ceiling_per_s": best["v_mad_i64_i32"] / 1467.0,
       "fp_mul_words_isa_ceiling_per_s": best["mad_addc_pair"] / 128.0, "g1_mixed_add_words_isa_ceiling_per_s": best["mad_addc_pair"] / 1280.0,
       "inv_gcd_cost_in_fp_mul": best["fp_mul"] / best["fp_inv_gcd"], "inv_fermat_cost_in_fp_mul": best["fp_mul"] / best["fp_inv_fermat"],      # in 8 x 32 products
       "note": "zkp_microbench (csrc/test_api.hip): back-to-back dependent operations per thread, best of 4/8/16 wavefronts per SIMD worth of threads"}
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
