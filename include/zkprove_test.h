/* libzkprove_test.so -- kernel-level entry points for the parity tests and micro-benchmarks.
 *
 * NOT part of the drop-in: libzkprove.so exports include/zkprove.h and nothing else.  This library is linked from the same object
 * files plus csrc/test_api.hip and exports both headers; tests/test_gpu_kernels.py, scripts/gpu_microbench.py and bench.py's
 * same-box ceiling leg load it.  (libzkprove_exp.so is the same with -DZK_EXPERIMENTS: the measured-and-lost alternatives.)
 * Inputs / outputs are host buffers of canonical little-endian 32-byte field elements in [instance][index] order unless stated.
 */
#ifndef ZKPROVE_TEST_H
#define ZKPROVE_TEST_H
#include "zkprove.h"
#ifdef __cplusplus
extern "C" {
#endif

/* data: [batch][2^logn][32] canonical LE; mode 0 fwd (DIF kernels) 1 inv (DIF) 2 fwd (DIT) 3 inv (DIT) */
int zkp_ntt(int logn, long batch, uint8_t *data, int mode);
#ifdef ZK_EXPERIMENTS   /* only in libzkprove_exp.so (make experiments): gnark's unfolded 7-transform computeH, which the product does not run */
/* computeH: a,b,c [batch][ncons][32] -> h [batch][2^logn][32] (natural coefficient order) */
int zkp_compute_h(int logn, long batch, long ncons, const uint8_t *a, const uint8_t *b, const uint8_t *c, uint8_t *h_out);
#endif
/* points n*64 (G1: x|y) / n*128 (G2: x.a0|x.a1|y.a0|y.a1) canonical LE; scalars [batch][n][32]; c = window bits */
int zkp_msm_g1(long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c);
int zkp_msm_g2(long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c);
/* same as zkp_msm_g1 through the dense (quotient-MSM) sort path: no 0/+-1 fast path, two-level sort when c >= 9 */
int zkp_msm_g1_dense(long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c);
/* wire MSM (0 / +-1 fast path + bucket pipeline) with a wire comb attached: cls[n] = width class of each base's scalars (0: 0 / +-1,
 * 1: < 2^8, 2: < 2^32, 3: any); group 1 = G1, 2 = G2; values wider than their class must still give the right sum (bucket hand-over) */
int zkp_msm_wire_comb(int group, long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c, int k, const uint8_t *cls);
/* same sum through the fixed-base bit-plane comb of the quotient MSM of large batches (csrc/msm_comb.hip): clumps of k bases, table
 * of their signed subset sums; mode & 0x0f == 1 passes the scalars as products of two matrices (the prover's call shape); mode & 0x10: the
 * split walk / fold of mid-size batches (comb_msm_split); mode >> 8 = n1:
 * the first n1 clumps hold k + 1 bases (mixed table, the shape the memory budget of the quotient tables produces).
 * returns 0, or 1 when the bases are degenerate for the comb (point at infinity, coinciding subset sum) and nothing was computed */
int zkp_msm_comb(long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int k, int mode);
/* timing of the comb path at production shape (2^logn synthetic bases, pseudo-random scalars): out_ms[5] = table build, patterns,
 * walk, fold, table GiB */
int zkp_bench_comb(int logn, long batch, int k, int iters, double *out_ms);
/* Key folding done at InitAlgorithm (csrc/keyfold.hip): pk.G1.Z (2^logn - 1 points, 64 B canonical LE x|y each) ->
 * ze[i] = 1/(N D) sum_k w^-ik g^-k Z_k (2^logn points) and zc[j] = -1/(N D) sum_k w^-jk Z_k (ncons points), D = g^N - 1;
 * replaces the last four transforms of gnark's computeH (backend/groth16/bn254/prove.go) per proof. */
int zkp_fold_quotient_bases(int logn, long ncons, const uint8_t *z, uint8_t *ze, uint8_t *zc);
/* second half of the folding: K'_w = K_w + sum_{terms of wire w} coeffs[coeff] * zc[cons] (the R1CS output matrix applied to zc).
 * k: the points of the wires rows_k (ascending); terms: (wire, cons, coeff index) u32 triples.  Returns the number of folded rows
 * written to out_rows / out_points (union of rows_k and the term wires, ascending), -1 on error. */
long zkp_fold_output_matrix(long nzc, const uint8_t *zc, long nk, const uint8_t *k, const uint32_t *rows_k, long nterms, const uint32_t *terms,
                            long ncoeff, const uint8_t *coeffs, uint32_t *out_rows, uint8_t *out_points);
/* device time (ms) of the prover's transform stage -- inverse + coset-forward NTT of two [2^logn][batch] matrices -- mean of iters runs */
double zkp_bench_to_coset(int logn, long batch, int iters);
/* field 0 = Fp, 1 = Fr; in / out: n canonical LE 32-byte elements; out[i] = in[i]^-1 (0 -> 0) by the device's binary-GCD inversion */
int zkp_field_inv(int field, long n, const uint8_t *in, uint8_t *out);
/* kind 0: Fp mul (2/iter) 1: G1 mixed add 2: G2 mixed add 3: Fp Fermat inversion 4: Fp binary-GCD inversion; returns milliseconds */
double zkp_microbench(int kind, long threads, int iters);


/* ---- pins against published / reference-made data (tests/test_pins.py, tests/test_gpu_pins.py) */
/* prod e(P_i, Q_i) == 1 through the verifier's host pairing (device = 0) or with the Miller loops in the batch verifier's device kernel
 * (device = 1).  Points in gnark-crypto's uncompressed encoding (G1: X | Y, G2: X.A1 | X.A0 | Y.A1 | Y.A0, 32-byte big-endian).
 * 1 / 0, -1: undecodable point */
int zkp_pairing_product_check(const uint8_t *g1_raw, const uint8_t *g2_raw, long n, int device);
/* a verifying key (gnark VerifyingKey.WriteTo bytes, e.g. the reference's verifier/impl/generated/vk.*) through the verifier's own key
 * reader: bit 0 = e(beta1, G2) == e(G1, beta2), bit 1 = e(delta1, G2) == e(G1, delta2); -1: the key does not parse */
int zkp_vk_pairing_identities(const uint8_t *vk, long len, int device);
/* RFC 9380 expand_message_xmd(SHA-256): device = 0 the host function behind the verifier's BSB22 challenge, device = 1 the device function
 * behind the prover's k_bsb22_challenge */
int zkp_expand_message_xmd(const uint8_t *msg, long mlen, const uint8_t *dst, long dlen, uint8_t *out, long out_len, int device);
/* the prover's challenge kernel on n commitments (x | y canonical little-endian 32-byte coordinates) -> n canonical LE Fr elements */
int zkp_bsb22_challenge(const uint8_t *com, long n, uint8_t *out);
/* the verifier's host path for the same: commitments as X | Y 32-byte big-endian (G1Affine.Marshal) */
int zkp_bsb22_challenge_host(const uint8_t *com_be64, long n, uint8_t *out);
/* ---- parity-test views of the resident batch (zkp_bench_upload / zkp_bench_step of THIS library; the drop-in does not export them) */
/* after a step: {sum, max, instances with any} of the entries the last main-stream wire MSM handed to the bucket pipeline */
int zkp_debug_leftovers(int algorithmID, long *out3);
/* the solver program of an initialised algorithm: {levels, chain runs, levels inside chain runs, wavefronts of the longest chain run}
 * (csrc/solver.hip, k_solve_chains: runs of narrow levels walked by one wavefront per independent component) */
int zkp_debug_solver_plan(int algorithmID, long *out4);
/* after a step: the solved wire vectors [n][n_wires][32] and the quotient-stage products f_i = A(g w^i) * B(g w^i) [n][N][32] (canonical LE,
 * natural order; the scalars of the folded quotient MSM, see csrc/keyfold.hip) of the n listed instances of the batch (instances == NULL:
 * all of them, n ignored); either output may be NULL */
int zkp_debug_fetch(int algorithmID, const uint32_t *instances, long n, uint8_t *wires, uint8_t *h);
/* ---- request layer */
/* the two request decoders (single-pass scanner for the canonical shape, DOM reader with Go's encoding/json semantics) on one input:
 * 1 = the scanner took it and agrees with the DOM reader field by field, 0 = the scanner declined, -1 = they disagree */
int zkp_request_decode_both(const uint8_t *json, long len);
/* host stages of a streamed ProveBatch without a device (decode + cipher + pack per 1 024-chunk; response serialisation):
 * out4 = {decode stage requests/s, serialise stage requests/s, requests the scanner took, requests that produced a record} */
int zkp_host_stage_bench(const uint8_t *json_array, long len, int iters, double *out4);
#ifdef ZK_EXPERIMENTS
int zkp_debug_tree(int n, int c, int piece, int dbg);
#endif

#ifdef __cplusplus
}
#endif
#endif
