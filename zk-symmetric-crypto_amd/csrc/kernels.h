// Internal C++ interface between the HIP kernel translation units and the prover host code.
#pragma once
#include <memory>
#include <vector>
#include "common.h"
#include "curve.cuh"

// The fold kernels are chains of dependent group operations on a handful of wavefronts (latency, not throughput); raising their issue
// priority (-DZK_FOLD_PRIO=3; the tail kernels and k_fold_parts_tree take it too) was measured twice: nothing for either circuit at 1 024 proofs
// (same box, three runs each, profiles/r05zk_*), nothing at 1..128 proofs per call (profiles/r05zn_*_fold_and_tail_priority_ab.txt) -- off
#ifndef ZK_FOLD_PRIO
#define ZK_FOLD_PRIO 0
#endif
#if ZK_FOLD_PRIO > 0
#define ZK_FOLD_SETPRIO() __builtin_amdgcn_s_setprio(ZK_FOLD_PRIO)
#else
#define ZK_FOLD_SETPRIO() ((void)0)
#endif

namespace zk {

// ------------------------------------------------------------------------------------------ NTT
struct NttDomain {
    int logn = 0;
    size_t n = 0;
    DevBuf<Fr> consts, tw_fwd, tw_inv, coset_pre, coset_post;
    bool f9 = false;                                           // butterflies on the 9 x 29-bit field (ntt.hip, Ntt9): opt-in experiment, ZKPROVE_NTT_F9=1
    bool hi_prio = false;                                      // the tile pass kernels raise their wavefronts' issue priority (ntt.hip, ZK_NTT_PRIO): set for the programs whose transforms are the window's critical chain
    DevBuf<Fr> tw_fwd9, tw_inv9, coset_pre9, coset_post9;      // the same multiplier tables in the 2^261-Montgomery form
    const Fr *table9(const Fr *t) const;
    void init(int logn, hipStream_t s);
    // data layout [n][B]; see ntt.hip
    void dif(Fr *data, size_t B, bool inverse, const Fr *postmul, hipStream_t s) const;
    void dit(Fr *data, size_t B, bool inverse, const Fr *premul, hipStream_t s) const;
#ifdef ZK_EXPERIMENTS
    void compute_h(Fr *a, Fr *b, Fr *c, size_t B, hipStream_t s) const;
#endif
    // a, b (per-constraint values) -> evaluations of their interpolants on the coset g<w>, natural order
    void to_coset(Fr *a, Fr *b, size_t B, hipStream_t s) const;
    void scale_ninv(Fr *a, size_t B, hipStream_t s) const;
};
void fr_to_mont(Fr *a, size_t total, hipStream_t s);
void fr_from_mont(Fr *a, size_t total, hipStream_t s);

// ------------------------------------------------------------------------------- comb MSM (msm_comb.hip)
// Fixed-base bit-plane comb: per clump of k consecutive bases the table of all signed subset sums (2^(k-1) affine entries,
// 2^261-Montgomery words, of the HALVED bases), see msm_comb.hip.  Used for the quotient MSM of large batches (T = 254 planes, scalars
// = products formed on the fly) and, per width class, for the wire positions of the AES circuits that are not 0 / +-1 (MsmBases::wcomb).
struct CombShape {
    size_t n = 0, nclumps = 0, E = 0;      // bases, clumps, entries per clump of size k
    int k = 0, T = 254;                    // clump size, bit planes
    // Mixed clump sizes (quotient tables; the memory budget is filled to the last GiB, Algorithm::comb_prepare): the FIRST n1 clumps hold
    // k + 1 bases and 2E entries each, the rest k bases and E entries.  Clump c starts at base c*k + min(c, n1) and at table entry
    // (c + min(c, n1)) * E.  n1 == 0: one size throughout (the wire combs).
    size_t n1 = 0;
    size_t clump_base(size_t c) const { return c * (size_t)k + (c < n1 ? c : n1); }
    size_t clump_entry(size_t c) const { return (c + (c < n1 ? c : n1)) * E; }
    size_t entries() const { return (nclumps + n1) * E; }
    DevBuf<uint32_t> rows;                 // wire comb: row of base j in the [row][B] scalar matrix
    DevBuf<uint32_t> base_index;           // wire comb: index of base j in the owning MsmBases (for the bucket-pipeline hand-over)
    bool ok = false;
    bool hi_prio = false;                  // the pattern kernel raises its wavefronts' issue priority (as the transforms in front of it: NttDomain::hi_prio)
    bool has_inf = false;                  // the table holds entries at infinity (a base at infinity, equal / opposite bases in a clump): the walk skips them
};
template <class F>
struct CombBasesT : CombShape {
    DevBuf<Affine<F>> table;               // [nclumps][E] (clump_entry(c): the first n1 clumps take 2E)
    DevBuf<Affine<F>> corr;                // (2^T - 1) * sum P_i / 2: the constant of the signed-bit form
};
using CombBases = CombBasesT<Fp>;
struct CombScratch {
    DevBuf<uint32_t> pat[2];               // [clump][T][B] table index | sign << 31; two copies: the next batch's is written while this one is walked
    DevBuf<uint8_t> acc, r1;               // range accumulators [Q][T][B] and first-stage fold results [T/8][B] (XYZZ9 records)
    void reserve(const CombBases &cb, size_t B, int nslots);
};
size_t comb_table_bytes(size_t n, int k);
// mixed sizes: n1 clumps of k + 1 bases, the rest of k (n1 * (k + 1) <= n): clumps and table bytes
size_t comb_mixed_clumps(size_t n, int k, size_t n1);
size_t comb_mixed_bytes(size_t n, int k, size_t n1);
// the largest n1 whose mixed table fits `bytes` (0 when not even the pure-k table does: the caller checks comb_table_bytes first)
size_t comb_mixed_fit(size_t n, int k, size_t bytes);
size_t comb_pattern_words(const CombShape &cb, size_t B);
size_t comb_acc_bytes(const CombShape &cb, size_t B, size_t x9_bytes);
size_t comb_r1_bytes(const CombShape &cb, size_t B, size_t x9_bytes);
// (bases at infinity and clumps whose subset sums cancel or coincide are tabulated by the complete addition rules; has_inf then makes
// the walk skip the entries at infinity)
bool comb_build(CombBases &cb, const G1Affine *d_points, size_t n, int k, hipStream_t s, size_t n1 = 0);
bool comb_build_planes(CombBasesT<Fp> &cb, const G1Affine *d_points, size_t n, int k, int T, hipStream_t s);
bool comb_build_planes(CombBasesT<Fp2> &cb, const G2Affine *d_points, size_t n, int k, int T, hipStream_t s);
// scalars x[row][B] (y == nullptr) or the products x * y, Montgomery form -> pat
void comb_patterns(const CombBases &cb, const Fr *x, const Fr *y, size_t B, uint32_t *pat, hipStream_t s);
// wire scalars through cb.rows; values wider than cb.T planes go to the compact list (gen_*) of the bucket pipeline instead
void comb_patterns_rows(const CombShape &cb, const Fr *scalars, size_t B, uint32_t *pat, uint32_t *gen_count, uint32_t *gen_idx, Fr *gen_val, size_t gen_stride, hipStream_t s);
// out[b] = sum_i scalar_i(b) * P_i
void comb_msm(const CombBases &cb, const uint32_t *pat, size_t B, CombScratch &sc, G1XYZZ *out, hipStream_t s, hipEvent_t *seg_events = nullptr);
// the same for a mid-size batch (one call's latency): two walk launches, the upper planes' fold on `side` beside the lower planes' walk; ev2: two events
void comb_msm_split(const CombBases &cb, const uint32_t *pat, size_t B, CombScratch &sc, G1XYZZ *out, hipStream_t s, hipStream_t side, hipEvent_t *ev2, hipEvent_t *seg_events = nullptr);
// its two halves: the walk leaves range accumulators in sc, the fold (latency bound) turns them into out[b]; the fold may run on another stream
void comb_walk(const CombBases &cb, const uint32_t *pat, size_t B, CombScratch &sc, hipStream_t s, hipEvent_t *seg_events = nullptr);
void comb_fold(const CombBases &cb, size_t B, CombScratch &sc, G1XYZZ *out, hipStream_t s);
// the same into out[b * out_stride] (a partial-sum slot of msm_batch)
void comb_msm_strided(const CombBasesT<Fp> &cb, const uint32_t *pat, size_t B, CombScratch &sc, G1XYZZ *out, size_t out_stride, hipStream_t s);
void comb_msm_strided(const CombBasesT<Fp2> &cb, const uint32_t *pat, size_t B, CombScratch &sc, G2XYZZ *out, size_t out_stride, hipStream_t s);

// ------------------------------------------------------------------------------------------ MSM
// One fixed base set (a slice of the proving key) prepared for batched bucket MSMs.
//   table[j * W + w] = 2^(c*w) * P_j   (affine, Montgomery)  -- precomputed window multiples so that
//   all windows of a scalar share ONE bucket set (single bucket reduction, no Horner doublings).
template <class F>
struct MsmBases {
    size_t n = 0;          // number of base points
    int c = 0;             // window bits (signed digits, 2^(c-1) buckets)
    int W = 0;             // windows
    DevBuf<Affine<F>> table;
    DevBuf<Affine<F>> tern;        // conditional-sum path: signed subset sums of every group of 6 bases (msm.hip, TERN_*)
    DevBuf<uint32_t> scalar_row;   // row index of the scalar of point j in the [row][B] scalar matrix
    bool f9 = false;               // table / tern coordinates are in the 2^261-Montgomery form of field9.cuh (walked by the *9 kernels of msm.hip)
    bool dense = false;            // quotient MSM: all scalars wide, read from an instance-major copy [B][dense_stride]
    size_t dense_stride = 0;
    // Wire comb (AES circuits; msm_comb.hip): the positions whose values are not 0 / +-1 in general -- found by a calibration solve,
    // Algorithm::wire_comb_prepare -- by width class (8 / 32 / 254 bit planes), each with comb tables of its own.  Batches that fill
    // whole wavefronts take these positions through the comb instead of sort -> buckets -> reduction; wide_mask tells the
    // conditional-sum kernel to leave them alone.
    DevBuf<Affine<F>> pts;         // the base points themselves (kept for the table build)
    std::vector<std::unique_ptr<CombBasesT<F>>> wcomb;
    DevBuf<uint8_t> wide_mask;     // [n]
};
using MsmBasesG1 = MsmBases<Fp>;
using MsmBasesG2 = MsmBases<Fp2>;

// points: n affine points in Montgomery form on the device (infinity = (0,0) allowed).
// h_scalar_rows == nullptr selects the dense mode (set mb.dense_stride afterwards).
void msm_bases_build(MsmBasesG1 &mb, const G1Affine *d_points, size_t n, const uint32_t *h_scalar_rows, int c, hipStream_t s);
void msm_bases_build(MsmBasesG2 &mb, const G2Affine *d_points, size_t n, const uint32_t *h_scalar_rows, int c, hipStream_t s);

// Scratch shared by all MSMs of a batch (sized for the largest).
struct MsmScratch {
    DevBuf<uint32_t> entries;        // [B][max_entries]
    DevBuf<uint32_t> bucket_start;   // [B][nb + 1]
    DevBuf<G1XYZZ> part1;            // partial sums G1
    DevBuf<G2XYZZ> part2;            // partial sums G2
    DevBuf<G1XYZZ> sums1;            // [B][nb] bucket sums G1
    DevBuf<G2XYZZ> sums2;            // [B][nb] bucket sums G2
    size_t B = 0;
    CombScratch wire_comb;           // patterns / accumulators of the wire combs of the MSMs that share this scratch (one at a time)
    void reserve_wire_comb(size_t pat_words, size_t acc_bytes, size_t r1_bytes) { wire_comb.pat[0].ensure(pat_words); wire_comb.acc.ensure(acc_bytes); wire_comb.r1.ensure(r1_bytes); }
    void reserve(size_t B, size_t max_entries_per_inst, size_t max_nb, size_t max_part);
    void reserve_sums(size_t B, size_t nb_g1, size_t nb_g2);
    DevBuf<uint32_t> gen_count, gen_idx;   // compact per-instance lists of the scalars that are neither 0 nor +-1
    DevBuf<Fr> gen_val;
    void reserve_compact(size_t B, size_t max_sparse_n);
    DevBuf<uint32_t> entries2, coarse_base;   // dense two-level sort: partitioned entries, per-instance coarse bin offsets
    DevBuf<G1XYZZ> vout1;                     // results of the virtual instances of a split small-batch dense MSM
    DevBuf<G2XYZZ> vout2;                     // the same for the G2 wire MSM of a one-proof call (msm_sparse_split)
    void reserve_dense(size_t B, size_t max_dense_entries, size_t nb = 0);
    DevBuf<uint32_t> entries_z, bucket_start_z;   // sorted list of the dense MSM when it is kept apart from the wire MSMs' (nb != 0 above)
    DevBuf<uint32_t> entries_zb, bucket_start_zb; // second copy (reserve_dense_second): the NEXT batch's list is sorted while this one is walked
    bool presorted = false;                       // msm_dense_presort has filled slot `zslot` for the next dense msm_batch call
    int zslot = 0;
    uint32_t *ez(int slot) { return slot ? entries_zb.p : entries_z.p; }
    uint32_t *bz(int slot) { return slot ? bucket_start_zb.p : bucket_start_z.p; }
    void reserve_dense_second(size_t B, size_t max_dense_entries, size_t nb);
    // affine reduction tree (msm_tree.cuh): ping-pong layouts, final bucket points, per-wavefront prefix-product rows
    DevBuf<G1Affine> treeA1, treeB1, fin1;
    DevBuf<uint8_t> tree_scratch;
    void reserve_tree(size_t B, size_t max_entries, size_t nb);
};

// out[b] (+)= sum_j scalars[row_j][b] * P_j for b in [0,B).  scalars: [rows][B] Fr Montgomery.
// If accumulate is false out[b] is overwritten.
// seg_events (optional): two events recorded around the bucket-accumulation kernel (k_msm_accumulate) for the roofline line
void msm_batch(const MsmBasesG1 &mb, const Fr *scalars, size_t B, MsmScratch &sc, G1XYZZ *out, hipStream_t s, hipEvent_t *seg_events = nullptr);
void msm_batch(const MsmBasesG2 &mb, const Fr *scalars, size_t B, MsmScratch &sc, G2XYZZ *out, hipStream_t s);
// sort stage of a dense msm_batch on its own (stream s), so that it can run beside other MSMs that share the scratch
// would msm_dense_presort sort this batch into slot 0 (two-level sort applicable, list buffers reserved)?
bool msm_dense_presort_applicable(const MsmBasesG1 &mb, size_t B, const MsmScratch &sc);
bool msm_dense_presort(const MsmBasesG1 &mb, const Fr *scalars, size_t B, MsmScratch &sc, hipStream_t s, int slot = 0);

// out[b * stride + i] = x[i * B + b] * y[i * B + b]
void fr_transpose_mul(const Fr *x, const Fr *y, Fr *out, size_t rows, size_t B, size_t stride, hipStream_t s);


// BSB22 challenges H2F(commitment.Marshal(), "bsb22-commitment") of n canonical affine commitments, Montgomery form (prover.hip)
void bsb22_challenge_device(const G1Affine *com_canonical, Fr *challenge_mont, size_t n, hipStream_t s);

// ------------------------------------------------------------------------------- key folding (keyfold.hip)
// Unscaled inverse DFT over G1: out[i] = sum_k w^{-ik} * data[k] for i < n_out (natural order; data is destroyed)
void g1_point_idft(G1XYZZ *data, const NttDomain &dom, size_t n_out, G1Affine *out, hipStream_t s);
// pk.G1.Z (N-1 points, Montgomery affine) -> evaluation-form bases Ze (N points) and the constraint-value bases Zc
void fold_quotient_bases(const G1Affine *d_z, const NttDomain &dom, size_t n_constraints, G1Affine *d_zeval, G1Affine *d_zc, hipStream_t s);
// one non-zero of the R1CS output matrix: constraint `cons` has the term coeffs[coeff] * wire
struct OutputTerm { uint32_t wire, cons, coeff; };
// K'_w = K_w + sum over the terms of wire w of coeffs[coeff] * Zc[cons].  d_k: pk.G1.K points of the wires rows_k (ascending).
// out_rows (ascending) = rows_k united with the wires of the terms; out_points[i] = K'_{out_rows[i]}.  d_coeffs: Montgomery.
void fold_output_matrix(const G1Affine *d_zc, const G1Affine *d_k, const std::vector<uint32_t> &rows_k, std::vector<OutputTerm> terms,
                        const Fr *d_coeffs, DevBuf<G1Affine> &out_points, std::vector<uint32_t> &out_rows, hipStream_t s);

// wire comb: cls[j] (host, per base of mb: 0 = 0 / +-1 only, 1 = < 2^8, 2 = < 2^32, 3 = any) -> tables per non-empty class, clumps of k
// bases; false when a table cannot be built (degenerate bases): mb is left without a wire comb
bool msm_attach_wire_comb(MsmBasesG1 &mb, const uint8_t *cls, int k, hipStream_t s);
bool msm_attach_wire_comb(MsmBasesG2 &mb, const uint8_t *cls, int k, hipStream_t s);
// scratch the wire combs of mb need at batch size B: pattern words, accumulator bytes, first-stage fold bytes (0 when mb has none)
void msm_wire_comb_needs(const MsmBasesG1 &mb, size_t B, size_t &pat_words, size_t &acc_bytes, size_t &r1_bytes);
void msm_wire_comb_needs(const MsmBasesG2 &mb, size_t B, size_t &pat_words, size_t &acc_bytes, size_t &r1_bytes);

// number of XYZZ partial slots per instance that msm_batch needs in MsmScratch::part*
size_t msm_part_stride(size_t n, int c);

// affine import helpers: canonical big-endian coordinate bytes -> Montgomery affine on device
void g1_from_be_bytes(G1Affine *d_out, const uint8_t *d_in /* n*64 */, size_t n, hipStream_t s);
void g2_from_be_bytes(G2Affine *d_out, const uint8_t *d_in /* n*128: X.A1|X.A0|Y.A1|Y.A0 */, size_t n, hipStream_t s);
// compressed forms (gnark-crypto marshal): returns via d_ok[i] = 0 on invalid point
void g1_decompress(G1Affine *d_out, const uint8_t *d_in /* n*32 */, size_t n, uint32_t *d_bad, hipStream_t s);
void g2_decompress(G2Affine *d_out, const uint8_t *d_in /* n*64 */, size_t n, uint32_t *d_bad, hipStream_t s);
// key-slice validation (gnark-crypto decoder semantics): reduced coordinates (d_raw = uncompressed source bytes or nullptr),
// on-curve, G2 subgroup membership; every failing point increments *d_bad
void g1_validate(const G1Affine *d_pts, const uint8_t *d_raw, size_t n, uint32_t *d_bad, hipStream_t s);
void g2_validate(const G2Affine *d_pts, const uint8_t *d_raw, size_t n, uint32_t *d_bad, hipStream_t s);
// XYZZ -> canonical (non-Montgomery) affine, little-endian limbs; infinity -> zeros
void g1_to_affine_canonical(const G1XYZZ *d_in, G1Affine *d_out, size_t n, hipStream_t s);
void g2_to_affine_canonical(const G2XYZZ *d_in, G2Affine *d_out, size_t n, hipStream_t s);

}  // namespace zk
