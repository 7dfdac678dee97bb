// Batched multi-scalar multiplication on BN254 G1 / G2 for gfx950.
//
// Replaces gnark-crypto v0.19.2 ecc/bn254 G1Jac.MultiExp / G2Jac.MultiExp (signed-digit bucket
// method; `processChunkG1BatchAffine`, `msmReduceChunkG1Affine`, `partitionScalars` in the
// reference's default.pgo -- the 68.6 % + 16.2 % of prove time), called by
// groth16.Prove from gnark/libraries/prover/impl/provers.go:179,269.
//
// MI355X design (one launch sequence serves a whole batch of B proofs that share the bases):
//   * fixed bases => window multiples 2^(c*w)*P_j are precomputed once per proving key
//     (table[j*W + w], affine).  All W windows of a scalar then feed ONE bucket set, so there is a
//     single bucket reduction per MSM and no Horner doubling chain.
//   * scalars arrive straight from the solver / NTT in the [row][B] Montgomery layout.
//     0, 1 and -1 (all ChaCha wire scalars, 94 % of the AES ones) never enter the bucket pipeline:
//     k_msm_ones does a conditional sum with lanes = instances (coalesced scalar rows, wave-uniform
//     base point).
//   * k_msm_sort: one workgroup per instance builds the bucket histogram in LDS (up to 128 KiB of
//     the 160 KiB), scans it and scatters (table index | sign) entries into bucket order.
//   * k_msm_accumulate: the sorted entry list of an instance is cut into equal pieces (boundaries
//     rounded to bucket boundaries); one thread per piece, perfectly balanced lanes, bucket sums to HBM.
//   * k_msm_bucket_reduce: one thread per 64 buckets, running-sum reduction (sum_k (k+1) B_k).
//   * k_sum_groups folds the per-chunk / per-segment partial sums.
// Results are group elements; any correct schedule yields the same affine point, which is what the
// parity tests compare bit-for-bit with the oracle.
#include <algorithm>
#include "kernels.h"
#include "curve9.cuh"
#include "g2_subgroup.cuh"
// Build flag ZK_EXPERIMENTS (make experiments -> libzkprove_exp.so): the measured-and-lost alternatives stay buildable and parity-tested
// without being part of the product library -- the batch-affine reduction tree (msm_tree.cuh, ZKPROVE_MSM_TREE=1) and the bucket / conditional
// -sum kernels on the 8 x 32 field (ZKPROVE_MSM_F9=0, ZKPROVE_MSM_F9_G2=0).  The product walks the 9 x 29-bit field only.
#ifdef ZK_EXPERIMENTS
#include "msm_tree.cuh"
#endif

namespace zk {

// buckets per bucket-reduction thread: 64 when that still gives >= 32k threads (any MSM at batch 1024), else 16 -- a small batch
// (one Prove() call) is pure latency of dependent additions, and 4x more threads cut it 4x
// -- and 8 for the batches of single Prove() calls (at most TINY_BATCH proofs), whose every chain of dependent additions is call latency
static constexpr int SEG_LOG = 6, SEG_LOG_SMALL = 4, SEG_LOG_TINY = 3;
static constexpr size_t TINY_BATCH = 8;
static constexpr uint32_t SPARSE_SPLIT = 16;          // virtual instances per proof of a one-proof wire MSM (msm_sparse_split)
static constexpr size_t SPARSE_SPLIT_MAX_B = 4;
static constexpr size_t FOLD_TREE_MAX_B = 128;      // batches up to this size fold their partial sums in one tree launch (k_fold_parts_tree)
static inline bool tiny_batch(size_t B) {
    static const bool on = [] { const char *e = ZK_AB_ENV("ZKPROVE_MSM_TINY"); return !(e && e[0] == '0'); }();
    return on && B <= TINY_BATCH;
}
static inline uint32_t seg_len(uint32_t nb, size_t B) {
    static const size_t min_threads = [] { const char *e = ZK_AB_ENV("ZKPROVE_SEG_THREADS"); return e ? (size_t)atol(e) : (size_t)32768; }();
    uint32_t L = nb < (1u << SEG_LOG) ? nb : (1u << SEG_LOG);
    if ((size_t)(nb / L) * B < min_threads && nb > (1u << SEG_LOG_SMALL)) L = 1u << SEG_LOG_SMALL;
    if (tiny_batch(B) && nb > (1u << SEG_LOG_TINY)) L = 1u << SEG_LOG_TINY;
    return L;
}
static constexpr size_t FOLD_GROUP = 8;      // partial sums are folded 8 at a time, level by level
static size_t part_stride_for(size_t n, int c, size_t B);
// Conditional-sum path (scalars 0 / +1 / -1): bases are taken TERN_G at a time and every signed subset sum of a group is
// precomputed, so a group costs one mixed addition whatever its +-1 pattern is.  Index of a pattern: sum (t_i + 1) 3^i,
// t_i in {-1, 0, +1}; the patterns above the centre (all zero) are stored, the ones below are their negatives.
#ifndef ZK_TERN_G
#define ZK_TERN_G 8                                        // 6: 364 entries (23 KB) per group, 8: 3 280 (205 KB), 9: 9 841 (615 KB)
#endif
static constexpr int TERN_G = ZK_TERN_G;
static constexpr int pow3(int e) { return e == 0 ? 1 : 3 * pow3(e - 1); }
static constexpr int TERN_CENTER = (pow3(TERN_G) - 1) / 2; // the all-zero pattern
static constexpr int TERN_ENTRIES = TERN_CENTER;           // patterns CENTER + 1 .. 3^G - 1
#ifndef ZK_ONES_BASES
#define ZK_ONES_BASES 128                                  // bases per conditional-sum thread (<= 128: the four marker words); A/B build: -DZK_ONES_BASES=64
#endif
static constexpr int ONES_GROUPS = ZK_ONES_BASES / TERN_G; // groups per conditional-sum thread
static_assert(pow3(TERN_G) <= 65536, "group patterns are parked as 16-bit words");
static constexpr int ONES_CHUNK = TERN_G * ONES_GROUPS;    // base points per conditional-sum thread
static constexpr int SORT_THREADS = 1024;
static constexpr int WCOMB_CLASSES = 3;                     // wire-comb width classes: 8, 32, 254 bit planes
// batches from this size take the wire combs (where the scratch they run on holds room for them): the walk's lanes are (plane, proof)
// pairs, so 16 proofs fill wavefronts with four planes each; below, the one-proof paths (sparse split, tree folds) are faster
static constexpr size_t WCOMB_MIN_B = 16;

#ifdef ZK_EXPERIMENTS
static bool tree_enabled();
// G1 bucket accumulation on the 9 x 29-bit field (field9.cuh): on unless ZKPROVE_MSM_F9=0 (A/B runs) or the affine-tree experiment,
// which reads the table in the 2^256 form, is selected
static bool f9_enabled() {
    static const bool on = [] { const char *e = getenv("ZKPROVE_MSM_F9"); return !(e && e[0] == '0'); }();
    return on && !tree_enabled();
}
// the same for G2 (the tower Fp29), with a switch of its own for A/B runs: ZKPROVE_MSM_F9_G2=0
static bool f9_g2_enabled() {
    static const bool on = [] { const char *e = getenv("ZKPROVE_MSM_F9_G2"); return !(e && e[0] == '0'); }();
    static const bool all = [] { const char *e = getenv("ZKPROVE_MSM_F9"); return !(e && e[0] == '0'); }();
    return on && all;
}
#else
static constexpr bool tree_enabled() { return false; }
static constexpr bool f9_enabled() { return true; }
static constexpr bool f9_g2_enabled() { return true; }
#endif

// ------------------------------------------------------------------------------ table precompute
template <class F>
__global__ __launch_bounds__(256) void k_window_multiples(const Affine<F> *__restrict__ pts, size_t n, int c, int W, XYZZ<F> *__restrict__ out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    XYZZ<F> cur = XYZZ<F>::from_affine(pts[j]);
    for (int w = 0; w < W; w++) {
        out[j * W + w] = cur;
        if (w + 1 < W)
            for (int k = 0; k < c; k++) cur = cur.dbl();
    }
}
template <class F>
__global__ __launch_bounds__(256) void k_xyzz_to_affine(const XYZZ<F> *__restrict__ in, Affine<F> *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = in[i].to_affine();
}

// table *= 2^5 per coordinate: 2^256-Montgomery words -> 2^261-Montgomery words (infinity (0, 0) stays)
__global__ __launch_bounds__(256) void k_table_to_f9(Fp *__restrict__ coords, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp k;
#pragma unroll
    for (int j = 0; j < 8; j++) k.l[j] = Fp9Params::SCALE32[j];
    coords[i] = coords[i] * k;
}

// tern[g * TERN_ENTRIES + e] = sum_i t_i * P_{6g+i} for the pattern e + TERN_CENTER + 1
template <class F>
__global__ __launch_bounds__(256) void k_ternary_table(const Affine<F> *__restrict__ pts, size_t n, size_t ngroups, Affine<F> *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ngroups * TERN_ENTRIES) return;
    const size_t g = t / TERN_ENTRIES;
    uint32_t pat = (uint32_t)(t % TERN_ENTRIES) + TERN_CENTER + 1;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = 0; i < TERN_G; i++, pat /= 3) {
        const uint32_t d = pat % 3;
        const size_t j = g * TERN_G + i;
        if (d == 1 || j >= n) continue;
        Affine<F> p = pts[j];
        if (d == 0) p.y = p.y.neg();
        acc.madd(p);
    }
    out[t] = acc.to_affine();
}

template <class F>
static void bases_build(MsmBases<F> &mb, const Affine<F> *d_points, size_t n, const uint32_t *h_rows, int c, hipStream_t s) {
    mb.n = n; mb.c = c; mb.W = (254 + c - 1) / c;
    mb.dense = (h_rows == nullptr);
    mb.scalar_row.alloc(n ? n : 1);
    if (n && h_rows) ZK_HIP(hipMemcpyAsync(mb.scalar_row.p, h_rows, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (n && h_rows) {          // wire MSM: the points are kept for a later wire-comb table build (msm_attach_wire_comb)
        mb.pts.alloc(n);
        ZK_HIP(hipMemcpyAsync(mb.pts.p, d_points, n * sizeof(Affine<F>), hipMemcpyDeviceToDevice, s));
    }
    size_t tot = n * mb.W;
    mb.table.alloc(tot ? tot : 1);
    const size_t ngroups = mb.dense ? 0 : (n + TERN_G - 1) / TERN_G;
    mb.tern.alloc(ngroups ? ngroups * TERN_ENTRIES : 1);
    if (!n) return;
    if (ngroups) hipLaunchKernelGGL(k_ternary_table<F>, dim3(ceil_div(ngroups * TERN_ENTRIES, 256)), dim3(256), 0, s, d_points, n, ngroups, mb.tern.p);
    // chunked to bound the XYZZ temp
    const size_t CH = 1 << 16;
    DevBuf<XYZZ<F>> tmp(CH * mb.W);
    for (size_t off = 0; off < n; off += CH) {
        size_t m = n - off < CH ? n - off : CH;
        hipLaunchKernelGGL(k_window_multiples<F>, dim3(ceil_div(m, 256)), dim3(256), 0, s, d_points + off, m, c, mb.W, tmp.p);
        hipLaunchKernelGGL(k_xyzz_to_affine<F>, dim3(ceil_div(m * mb.W, 256)), dim3(256), 0, s, tmp.p, mb.table.p + off * mb.W, m * mb.W);
    }
    mb.f9 = sizeof(F) == 32 ? f9_enabled() : f9_g2_enabled();
    if (mb.f9) {
        const size_t per = sizeof(Affine<F>) / sizeof(Fp);           // Fp coordinates per point: 2 (G1) / 4 (G2)
        hipLaunchKernelGGL(k_table_to_f9, dim3(ceil_div(tot * per, 256)), dim3(256), 0, s, (Fp *)mb.table.p, tot * per);
        const size_t tn = ngroups * TERN_ENTRIES * per;
        if (tn) hipLaunchKernelGGL(k_table_to_f9, dim3(ceil_div(tn, 256)), dim3(256), 0, s, (Fp *)mb.tern.p, tn);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(s));
}
void msm_bases_build(MsmBasesG1 &mb, const G1Affine *p, size_t n, const uint32_t *rows, int c, hipStream_t s) { bases_build<Fp>(mb, p, n, rows, c, s); }
void msm_bases_build(MsmBasesG2 &mb, const G2Affine *p, size_t n, const uint32_t *rows, int c, hipStream_t s) { bases_build<Fp2>(mb, p, n, rows, c, s); }

// A small batch splits every instance of the dense MSM into up to DENSE_SPLIT_MAX "virtual instances" (a slice of the scalars
// each, own buckets) so that its sort / accumulate / reduce chain has workgroups to spread over; the per-instance arrays are
// therefore sized for at least VIRT_MIN instances.
static constexpr size_t VIRT_MIN = 64, DENSE_SPLIT_MAX = 16;
void MsmScratch::reserve_dense(size_t B_, size_t max_dense_entries, size_t nb) {
    entries2.ensure(B_ * max_dense_entries + 1);
    if (nb) {                                   // own sorted list of the dense MSM (see msm_dense_presort)
        entries_z.ensure(B_ * max_dense_entries + 1);
        bucket_start_z.ensure((B_ > VIRT_MIN ? B_ : VIRT_MIN) * (nb + 1));
    }
    coarse_base.ensure((B_ > VIRT_MIN ? B_ : VIRT_MIN) * 257);
    vout1.ensure(VIRT_MIN + 1);
}
void MsmScratch::reserve_dense_second(size_t B_, size_t max_dense_entries, size_t nb) {
    entries_zb.ensure(B_ * max_dense_entries + 1);
    bucket_start_zb.ensure((B_ > VIRT_MIN ? B_ : VIRT_MIN) * (nb + 1));
}
// lanes per instance of the accumulation kernels: ~2048 entries per lane at the worst case (all scalars wide); 2048 measured best
// at B = 1024 (among 512..4096); small batches (a single Prove() call) cut the list finer so that the launch still has ~256k threads
static uint32_t accumulate_lanes(size_t max_entries, size_t B) {
    size_t piece = 2048;
    // (a lane takes whole buckets: below the bucket size a finer cut changes nothing -- the wire MSMs' buckets hold ~28 entries, the
    // quotient MSM's of a one-proof call two: 64 dependent additions per lane there were 0.5 ms of the call's critical chain)
    const size_t floor_piece = tiny_batch(1) && B <= 32 ? 16 : 64;
    while (piece > floor_piece && (max_entries * B) / piece < 262144) piece >>= 1;
    // parity tests force the production cut (many buckets and several inversion chunks per thread) at batch sizes the CPU oracle can follow
    static const size_t forced = [] { const char *e = ZK_AB_ENV("ZKPROVE_MSM_PIECE"); return e ? (size_t)atol(e) : (size_t)0; }();
    if (forced >= 16) piece = forced;
    const uint32_t nlanes = (uint32_t)((max_entries + piece - 1) / piece);
    return nlanes < 1 ? 1 : nlanes;
}
void MsmScratch::reserve_tree(size_t B_, size_t max_entries, size_t nb) {
#ifdef ZK_EXPERIMENTS
    if (!tree_enabled()) return;                 // tens of GB at batch 1024: only for the experiment
    const size_t Bm = B_ > VIRT_MIN ? B_ : VIRT_MIN;
    // the finest cut any batch size uses decides the wavefront count and the per-piece slack of the buffers
    size_t waves = 0, needA = 0, needB = 0;
    for (size_t b = 1;; b <<= 1) {
        const size_t bb = b < Bm ? b : Bm;
        const size_t nl = accumulate_lanes(max_entries, bb);
        waves = std::max(waves, (size_t)ceil_div(nl * bb, 64));
        needA = std::max(needA, bb * tree_capacity(max_entries, nb, nl, 1));
        needB = std::max(needB, bb * tree_capacity(max_entries, nb, nl, 2));
        if (bb == Bm) break;
    }
    treeA1.ensure(needA);
    treeB1.ensure(needB);
    fin1.ensure(Bm * nb);
    tree_scratch.ensure((waves + 1) * TREE_CH * 64 * tree_slot_bytes<Fp>());
#else
    (void)B_; (void)max_entries; (void)nb;
#endif
}
void MsmScratch::reserve_compact(size_t B_, size_t max_sparse_n) {
    gen_count.ensure(B_ + 1);
    gen_idx.ensure(B_ * max_sparse_n + 1);
    gen_val.ensure(B_ * max_sparse_n + 1);
}
void MsmScratch::reserve_sums(size_t B_, size_t nb_g1, size_t nb_g2) {
    sums1.ensure(((B_ > VIRT_MIN ? B_ : VIRT_MIN) * nb_g1 + 1) * sizeof(G1X9) / sizeof(G1XYZZ) + 1);      // G1 bucket records are G1X9 (144 bytes) on the 9-limb path
    sums2.ensure((std::max<size_t>(B_, SPARSE_SPLIT * std::min(B_, SPARSE_SPLIT_MAX_B)) * nb_g2 + 1) * sizeof(G2X9) / sizeof(G2XYZZ) + 1);      // (msm_sparse_split: 16 virtual instances per proof, for every batch up to B_)
}
void MsmScratch::reserve(size_t B_, size_t max_entries, size_t max_nb, size_t max_part) {
    B = B_;
    entries.ensure(B * max_entries + 1 + 4096);      // (+ the rounding of msm_sparse_split's shares: up to 15 scalars' entries per proof)
    const size_t Bm = B > VIRT_MIN ? B : VIRT_MIN;
    bucket_start.ensure(Bm * (max_nb + 1));
    part1.ensure(Bm * max_part + 1);
    // Reservations only grow, and a scratch reserved for B serves every batch up to B: rows for the largest need of ANY of them -- B
    // itself, twice the instances of a tiny batch (part_stride_for: the finer bucket reduction of batches <= TINY_BATCH needs up to
    // twice the partial slots per instance), SPARSE_SPLIT virtual instances per proof of a split batch (msm_sparse_split).  (Sized by
    // the reserving batch alone, a first call of 9..15 proofs left a later batch of 8 without room: "msm scratch too small".)
    part2.ensure(std::max({B, 2 * std::min(B, TINY_BATCH), SPARSE_SPLIT * std::min(B, SPARSE_SPLIT_MAX_B)}) * max_part + 1);
    vout1.ensure(VIRT_MIN + 1);                                          // (msm_sparse_split)
    vout2.ensure(VIRT_MIN + 1);
}

// --------------------------------------------------------------------------------- digit helpers
// scalar classes: 0 -> skip, +1 / -1 -> conditional-sum path (k_msm_ones), else bucket path.
// (-1 = r-1 is as common as it is wide: every "wide" wire of the ChaCha circuit and 2/3 of the AES ones.)
__device__ __forceinline__ Fr fr_minus_one() { return Fr::one().neg(); }
__device__ __forceinline__ int classify(const Fr &s) {
    if (s.is_zero()) return 0;
    if (s == Fr::one() || s == fr_minus_one()) return 1;
    return 2;
}
// signed digit w of canonical scalar (window c); carry in/out
__device__ __forceinline__ int signed_digit(const uint32_t *l, int w, int c, int &carry) {
    const int bit = w * c;
    const int li = bit >> 5, sh = bit & 31;
    uint64_t v = l[li];
    if (li + 1 < 8) v |= (uint64_t)l[li + 1] << 32;
    int raw = (int)((v >> sh) & ((1u << c) - 1)) + carry;
    if (raw > (1 << (c - 1))) { carry = 1; return raw - (1 << c); }
    carry = 0;
    return raw;
}

// One workgroup per instance.  dyn LDS: nb counters + SORT_THREADS scan slots.
// DENSE  : every scalar is wide (quotient MSM); scalars are read from the instance-major copy
//          dense[b * dense_stride + j] (Montgomery), fully coalesced.
// !DENSE : only the scalars that k_msm_ones found to be neither 0 nor +-1 are processed: compact per-instance
//          list gen_idx / gen_val (canonical) of length gen_count[b].
template <bool DENSE>
__global__ __launch_bounds__(SORT_THREADS) void k_msm_sort(const Fr *__restrict__ dense, size_t dense_stride, const uint32_t *__restrict__ gen_count,
                                                           const uint32_t *__restrict__ gen_idx, const Fr *__restrict__ gen_val, size_t n, int c,
                                                           int W, uint32_t *__restrict__ bucket_start, uint32_t *__restrict__ entries,
                                                           size_t max_entries, const uint32_t *__restrict__ skip_if_zero, uint32_t split = 1) {
    extern __shared__ uint32_t lds[];
    const uint32_t nb = 1u << (c - 1);
    uint32_t *hist = lds;
    uint32_t *scan = lds + nb;
    // split > 1 (msm_sparse_split): workgroup blockIdx.x sorts the share blockIdx.x % split of instance blockIdx.x / split as an
    // instance of its own (bucket_start / entries rows by blockIdx.x)
    const size_t bb = blockIdx.x, b = bb / split;
    const uint32_t vshare = (uint32_t)(bb % split);
    const int tid = threadIdx.x;
    // wire MSMs (9-limb kernels): *skip_if_zero = gen_count[B] is the batch total of the compact lists; 0 (the normal case once the wire
    // combs take the wide positions, and always for the ChaCha20 circuit) means no instance has anything for the bucket pipeline: its
    // kernels -- this one, k_msm_accumulate9 and k_msm_bucket_reduce9 -- return at once
    if (skip_if_zero && *skip_if_zero == 0) return;
    const size_t cnt_all = DENSE ? n : (size_t)gen_count[b];
    const size_t k_lo = cnt_all * vshare / split, cnt = cnt_all * (vshare + 1) / split;      // this workgroup's entries of the list: [k_lo, cnt)
    uint32_t *bs = bucket_start + bb * (nb + 1);
    if (cnt == k_lo) {     // nothing for the bucket pipeline: publish an empty histogram and leave
        for (uint32_t i = tid; i <= nb; i += SORT_THREADS) bs[i] = 0;
        return;
    }
    for (uint32_t i = tid; i < nb; i += SORT_THREADS) hist[i] = 0;
    __syncthreads();
    auto load = [&](size_t k, uint32_t &j) -> Fr {
        if (DENSE) { j = (uint32_t)k; return dense[b * dense_stride + k].from_mont(); }
        j = gen_idx[b * n + k];
        return gen_val[b * n + k];
    };
    for (size_t k = k_lo + tid; k < cnt; k += SORT_THREADS) {
        uint32_t j;
        Fr cs = load(k, j);
        int carry = 0;
        for (int w = 0; w < W; w++) {
            int d = signed_digit(cs.l, w, c, carry);
            if (d) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
        }
    }
    __syncthreads();
    const uint32_t per = (nb + SORT_THREADS - 1) / SORT_THREADS;
    const uint32_t lo = tid * per;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < lo + per && i < nb; i++) sum += hist[i];
    scan[tid] = sum;
    __syncthreads();
    for (int off = 1; off < SORT_THREADS; off <<= 1) {
        uint32_t v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    uint32_t run = scan[tid] - sum;   // exclusive prefix of this thread's chunk
    for (uint32_t i = lo; i < lo + per && i < nb; i++) {
        uint32_t cntb = hist[i];
        hist[i] = run;                // becomes the scatter cursor
        bs[i] = run;
        run += cntb;
    }
    if (tid == SORT_THREADS - 1) bs[nb] = scan[tid];
    __syncthreads();
    uint32_t *ent = entries + bb * max_entries;
    for (size_t k = k_lo + tid; k < cnt; k += SORT_THREADS) {
        uint32_t j;
        Fr cs = load(k, j);
        int carry = 0;
        for (int w = 0; w < W; w++) {
            int d = signed_digit(cs.l, w, c, carry);
            if (!d) continue;
            uint32_t neg = d < 0;
            uint32_t pos = atomicAdd(&hist[(neg ? -d : d) - 1], 1u);
            ent[pos] = (uint32_t)((size_t)j * W + w) | (neg << 31);
        }
    }
}

// ---- dense two-level sort (quotient MSM) -------------------------------------------------------
// A single-level scatter writes 4-byte entries to random places of a 4 MiB region per instance: on HBM that is
// one 64-byte transaction per entry (26 ms per batch).  Two levels keep every global write in >= 128-byte runs:
//   k_msm_partition: one workgroup per instance; tiles of 512 scalars x W digits are counting-sorted in LDS by
//                    COARSE bin (top bits of the bucket id) and appended to the bin's region as contiguous runs;
//   k_msm_binsort  : one workgroup per (coarse bin, instance): counting sort of the bin (~4096 entries) by the
//                    7 low bucket bits into the final array; all scattered writes stay inside a 16 KiB window (L2).
// Intermediate entry: idx (21 bits) | fine bucket (7 bits) << 21 | sign << 31.
// Exclusive prefix sum of up to 256 LDS counters by ONE wavefront (lanes 0..63 of the workgroup, four counters per lane):
// out[k] = sum_{i<k} in[i]; returns the total in every lane.  Replaces single-thread scans that cost ~70 cycles of LDS latency
// per counter, once per tile / bin.
__device__ __forceinline__ uint32_t wave_exclusive_scan_256(const uint32_t *in, uint32_t *out, uint32_t n, int lane) {
    uint32_t v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t k = 4u * lane + q; v[q] = k < n ? in[k] : 0u; s += v[q]; }
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    uint32_t run = incl - s;
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t k = 4u * lane + q; if (k < n) out[k] = run; run += v[q]; }
    return __shfl(incl, 63, 64);
}

static constexpr int FINE_BITS = 7;
static constexpr int PART_THREADS = 512;
static constexpr int MAX_COARSE = 256;

__global__ __launch_bounds__(PART_THREADS) void k_msm_partition(const Fr *__restrict__ dense, size_t dense_stride, size_t n, int c, int W,
                                                                uint32_t *__restrict__ coarse_base, uint32_t *__restrict__ part_entries,
                                                                size_t max_entries, uint32_t split) {
    // workgroup = virtual instance blockIdx.x = instance * split + slice; it sorts scalars [slice * n, slice * n + n) of the instance
    // (n = scalars per slice) and emits GLOBAL table indices
    extern __shared__ uint32_t lds[];
    const int cb = c - 1 - FINE_BITS;                 // coarse bits
    const uint32_t nbin = 1u << cb;
    const int nwave = PART_THREADS / 64;
    uint32_t *wave_hist = lds;                        // [nwave][nbin]  (pass 0)
    uint32_t *bin_base = wave_hist + nwave * MAX_COARSE;   // [nbin + 1]
    uint32_t *bin_cursor = bin_base + MAX_COARSE + 1; // [nbin]
    uint32_t *tile_cnt = bin_cursor + MAX_COARSE;     // [nbin]
    uint32_t *tile_off = tile_cnt + MAX_COARSE;       // [nbin]
    uint32_t *tile_ent = tile_off + MAX_COARSE;       // [PART_THREADS * W]
    uint32_t *tile_dst = tile_ent + PART_THREADS * W;     // tiles sized by the actual window count: 73 KiB at W = 16 -> two workgroups per CU
    const size_t b = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6;
    const size_t jbase = (size_t)(blockIdx.x % split) * n;
    const Fr *sc = dense + (b / split) * dense_stride + jbase;
    // ---- pass 0: coarse histogram (privatised per wavefront)
    for (uint32_t i = tid; i < (uint32_t)nwave * nbin; i += PART_THREADS) wave_hist[(i / nbin) * MAX_COARSE + (i % nbin)] = 0;
    if (tid < (int)nbin) { bin_cursor[tid] = 0; tile_cnt[tid] = 0; }
    __syncthreads();
    for (size_t j = tid; j < n; j += PART_THREADS) {
        const Fr cs = sc[j].from_mont();
        int carry = 0;
        for (int w = 0; w < W; w++) {
            const int d = signed_digit(cs.l, w, c, carry);
            if (d) atomicAdd(&wave_hist[wave * MAX_COARSE + ((uint32_t)((d < 0 ? -d : d) - 1) >> FINE_BITS)], 1u);
        }
    }
    __syncthreads();
    if (tid < (int)nbin) {                      // fold the per-wavefront histograms into tile_off (scratch here)
        uint32_t t = 0;
        for (int wv = 0; wv < nwave; wv++) t += wave_hist[wv * MAX_COARSE + tid];
        tile_off[tid] = t;
    }
    __syncthreads();
    if (tid < 64) {
        const uint32_t total = wave_exclusive_scan_256(tile_off, bin_base, nbin, tid);
        if (tid == 0) bin_base[nbin] = total;
    }
    __syncthreads();
    for (uint32_t k = tid; k <= nbin; k += PART_THREADS) coarse_base[b * (MAX_COARSE + 1) + k] = bin_base[k];
    // ---- pass A: tiles of PART_THREADS scalars
    uint32_t *out = part_entries + b * max_entries;
    const size_t ntiles = (n + PART_THREADS - 1) / PART_THREADS;
    for (size_t tile = 0; tile < ntiles; tile++) {
        const size_t j = tile * PART_THREADS + tid;
        // statically indexed per-window slots (a runtime-indexed array would live in scratch); 0xffffffff = no entry
        uint32_t ent[24];
        uint32_t binrank[24];
#pragma unroll
        for (int w = 0; w < 24; w++) binrank[w] = 0xffffffffu;
        if (j < n) {
            const Fr cs = sc[j].from_mont();
            int carry = 0;
#pragma unroll
            for (int w = 0; w < 24; w++) {
                if (w < W) {
                    const int d = signed_digit(cs.l, w, c, carry);
                    if (d) {
                        const uint32_t neg = d < 0, bk = (uint32_t)(neg ? -d : d) - 1;
                        const uint32_t bin = bk >> FINE_BITS;
                        const uint32_t r = atomicAdd(&tile_cnt[bin], 1u);
                        ent[w] = (uint32_t)((jbase + j) * W + w) | ((bk & ((1u << FINE_BITS) - 1)) << 21) | (neg << 31);
                        binrank[w] = (bin << 16) | r;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 64) wave_exclusive_scan_256(tile_cnt, tile_off, nbin, tid);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 24; w++) {
            if (binrank[w] != 0xffffffffu) {
                const uint32_t bin = binrank[w] >> 16, r = binrank[w] & 0xffff;
                const uint32_t pos = tile_off[bin] + r;
                tile_ent[pos] = ent[w];
                tile_dst[pos] = bin_base[bin] + bin_cursor[bin] + r;
            }
        }
        __syncthreads();
        uint32_t total = tile_off[nbin - 1] + tile_cnt[nbin - 1];
        for (uint32_t i = tid; i < total; i += PART_THREADS) out[tile_dst[i]] = tile_ent[i];
        __syncthreads();
        if (tid < (int)nbin) { bin_cursor[tid] += tile_cnt[tid]; tile_cnt[tid] = 0; }
        __syncthreads();
    }
}

// grid (nbin, B), 256 threads.  Bins that fit BIN_CAP entries (the normal case: ~4096 +- 64) are sorted entirely in LDS and
// written back as one linear run; larger bins (adversarial scalar sets) fall back to scattering inside the bin's window.
// 9 472 = the 8 192 entries per bin of the AES circuits' quotient MSM (2^17 scalars x 16 windows / 256 bins) + 14 sigma; 74 KiB of
// LDS per workgroup, two workgroups per CU
static constexpr uint32_t BIN_CAP = 9472;
__global__ __launch_bounds__(256) void k_msm_binsort(const uint32_t *__restrict__ coarse_base, const uint32_t *__restrict__ part_entries,
                                                     uint32_t *__restrict__ entries, size_t max_entries, int c,
                                                     uint32_t *__restrict__ bucket_start) {
    __shared__ uint32_t hist[1 << FINE_BITS];
    __shared__ uint32_t cursor[1 << FINE_BITS];
    __shared__ uint32_t buf_in[BIN_CAP], buf_out[BIN_CAP];
    const uint32_t nb = 1u << (c - 1);
    const uint32_t bin = blockIdx.x;
    const size_t b = blockIdx.y;
    const int tid = threadIdx.x;
    const uint32_t lo = coarse_base[b * (MAX_COARSE + 1) + bin], hi = coarse_base[b * (MAX_COARSE + 1) + bin + 1];
    const uint32_t cnt = hi - lo;
    const bool fits = cnt <= BIN_CAP;
    const uint32_t *in = part_entries + b * max_entries;
    uint32_t *out = entries + b * max_entries;
    uint32_t *bs = bucket_start + b * (nb + 1) + ((size_t)bin << FINE_BITS);
    const uint32_t fmask = (1u << FINE_BITS) - 1;
    if (tid < (1 << FINE_BITS)) hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += 256) {
        const uint32_t v = in[lo + i];
        if (fits) buf_in[i] = v;
        atomicAdd(&hist[(v >> 21) & fmask], 1u);
    }
    __syncthreads();
    if (tid < 64) {
        const uint32_t total = wave_exclusive_scan_256(hist, cursor, 1u << FINE_BITS, tid);
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t k = 4u * tid + q; if (k < (1u << FINE_BITS)) bs[k] = lo + cursor[k]; }    // own writes
        if (tid == 0 && bin + 1 == gridDim.x) bs[1 << FINE_BITS] = lo + total;          // bucket_start[nb] = total
    }
    __syncthreads();
    if (fits) {
        for (uint32_t i = tid; i < cnt; i += 256) {
            const uint32_t v = buf_in[i];
            buf_out[atomicAdd(&cursor[(v >> 21) & fmask], 1u)] = v & 0x801fffffu;
        }
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += 256) out[lo + i] = buf_out[i];
    } else {
        for (uint32_t i = tid; i < cnt; i += 256) {
            const uint32_t v = in[lo + i];
            out[lo + atomicAdd(&cursor[(v >> 21) & fmask], 1u)] = v & 0x801fffffu;
        }
    }
}

// Bucket accumulation, balanced: instance b's sorted entry list is cut into `nlanes` pieces of (almost)
// equal length whose boundaries are rounded up to the next bucket boundary; one thread walks one piece
// linearly.  A bucket change only costs a predicated 128-byte store, so all lanes of a wavefront run the
// same number of mixed additions (+-1 bucket) -- the earlier "one thread per 64 buckets" form lost ~30 %
// to the spread of Poisson(32) bucket sizes across the 64 lanes.
// sums[b * nb + k] receives the sum of bucket k when it is non-empty (empty buckets are never written;
// the reduce kernel reads emptiness from bucket_start).
// DENSE only names the instantiation (quotient MSM vs. wire MSMs) so that profiler summaries keep the dominant launch apart.
// both groups accumulate in weak form (madd_w); G1 with the multiplier expanded in place, G2 through the shared out-of-line copy
// Accumulator policies of the walk below.  AccWords: XYZZ<F> in weak form on the 8 x 32 field (G2; G1 when the 9-limb path is off).
// Acc9: the 9 x 29-bit field (field9.cuh / curve9.cuh; Fp9 for G1, the tower Fp29 for G2) -- the table then holds the 2^261-Montgomery
// form (bases_build), the bucket sums are XYZZ9 records read by k_msm_bucket_reduce9, which hands the canonical 2^256 form on.
// 9-limb counterparts of the coordinate fields

#ifdef ZK_EXPERIMENTS
template <class F>
struct AccWords {
    using Out = XYZZ<F>;
    XYZZ<F> a;
    __device__ __forceinline__ void reset() { a = XYZZ<F>::inf(); }
    __device__ __forceinline__ void add(Affine<F> p, bool negate) {
        if (negate) p.y = p.y.neg();
        if constexpr (sizeof(F) == 32) a.template madd_w<true>(p);     // multiplier expanded in place
        else a.template madd_w<false>(p);                              // G2: shared out-of-line weak multiplier
    }
    __device__ __forceinline__ XYZZ<F> result() const { return a.canon(); }
};
#endif  // ZK_EXPERIMENTS
template <class F>
struct Acc9 {
    using F9 = typename Nine<F>::F9;
    using X9 = typename Nine<F>::X9;
    using Out = X9;                // raw strict limbs (144 / 288 bytes): leaving a bucket costs no arithmetic -- a wavefront leaves some bucket
                                   // in almost every iteration, so whatever a store costs is paid by all 64 lanes nearly every time
    X9 a;
    __device__ __forceinline__ void reset() { a = X9::inf(); }
    __device__ __forceinline__ void add(const Affine<F> &p, bool negate) {
        if (p.is_inf()) return;
        a.madd(F9::from_words(p.x), F9::cneg(F9::from_words(p.y), negate ? -1 : 0));
    }
    __device__ __forceinline__ X9 result() const { return a; }
};
template <class F, bool DENSE, class ACC>
__device__ __forceinline__ void accumulate_body(const Affine<F> *__restrict__ table, const uint32_t *__restrict__ bucket_start,
                                                const uint32_t *__restrict__ entries, size_t max_entries, int c, size_t B,
                                                uint32_t nlanes, typename ACC::Out *__restrict__ sums, const uint32_t *__restrict__ skip_if_zero = nullptr) {
    const uint32_t nb = 1u << (c - 1);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (skip_if_zero && *skip_if_zero == 0) return;                 // empty compact lists (k_msm_sort did not run either)
    if (t >= (size_t)nlanes * B) return;
    const size_t b = t / nlanes;
    const uint32_t lane = (uint32_t)(t % nlanes);
    const uint32_t *bs = bucket_start + b * (nb + 1);
    const uint32_t *ent = entries + b * max_entries;
    typename ACC::Out *out = sums + b * nb;
    const uint32_t total = bs[nb];
    // first bucket k with bs[k] >= target
    auto first_bucket = [&](uint32_t target) -> uint32_t {
        uint32_t lo = 0, hi = nb;          // answer in [0, nb]
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (bs[mid] >= target) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    const uint32_t k0 = first_bucket((uint32_t)(((uint64_t)total * lane + nlanes - 1) / nlanes));
    const uint32_t k1 = lane + 1 == nlanes ? nb : first_bucket((uint32_t)(((uint64_t)total * (lane + 1) + nlanes - 1) / nlanes));
    if (k0 >= k1) return;
    uint32_t e = bs[k0];
    const uint32_t e_end = bs[k1];
    if (e >= e_end) return;
    // bucket boundaries are prefetched one ahead: leaving a bucket then costs no dependent memory round trip
    uint32_t cur = k0, cur_start = e, next_b = bs[k0 + 1];
    uint32_t next_b2 = bs[(k0 + 2 <= nb) ? k0 + 2 : nb];
    ACC acc;
    acc.reset();
    uint32_t v = ent[e];
    Affine<F> p = table[v & 0x7fffffffu];
    while (true) {
        while (e >= next_b) {              // leave bucket `cur` (also skips empty buckets)
            if (next_b > cur_start) { out[cur] = acc.result(); acc.reset(); }   // every non-empty bucket is written, even if its sum is infinity
            cur++;
            cur_start = next_b;
            next_b = next_b2;
            next_b2 = bs[(cur + 2 <= nb) ? cur + 2 : nb];
        }
        // only the next entry WORD is fetched ahead; the base point itself is loaded just in time (requesting it before the
        // addition costs 16 live registers = one wavefront per SIMD less and measured nothing: 72.8 vs 73.1 ms)
        const uint32_t e_next = e + 1;
        uint32_t v_next = 0;
        if (e_next < e_end) v_next = ent[e_next];
        acc.add(p, (v >> 31) != 0);
        if (e_next >= e_end) break;
        e = e_next; v = v_next; p = table[v & 0x7fffffffu];
    }
    out[cur] = acc.result();
}
// Fp2 kernels: 2 wavefronts per SIMD (the unconstrained allocation takes 256 VGPRs plus a few AGPRs = 1 wavefront per SIMD, with
// nothing to hide a 128-byte gather behind)
#define ZK_FP2_WAVES(F) (sizeof(F) == 64 ? 2 : 1)
#ifdef ZK_EXPERIMENTS
template <class F, bool DENSE>
__global__ __launch_bounds__(64, ZK_FP2_WAVES(F)) void k_msm_accumulate(const Affine<F> *__restrict__ table, const uint32_t *__restrict__ bucket_start,
                                                       const uint32_t *__restrict__ entries, size_t max_entries, int c, size_t B,
                                                       uint32_t nlanes, XYZZ<F> *__restrict__ sums) {
    accumulate_body<F, DENSE, AccWords<F>>(table, bucket_start, entries, max_entries, c, B, nlanes, sums);
}
// G1: 3 wavefronts per SIMD (168 VGPRs, 64 B of scratch) hide the multiplier's dependent chains better than 2 at 190 VGPRs
// (measured 111 vs 115 ms for the quotient MSM; 4 wavefronts at 128 VGPRs spill too much: 120 ms).  G2 needs all 256.
#ifndef ZK_ACC_WAVES
#define ZK_ACC_WAVES 3
#endif
#define ZK_ACC_G1(DENSE)                                                                                                                  \
    template <>                                                                                                                           \
    __global__ __launch_bounds__(64, ZK_ACC_WAVES) void k_msm_accumulate<Fp, DENSE>(const Affine<Fp> *__restrict__ table, const uint32_t *__restrict__ bucket_start, \
                                                                         const uint32_t *__restrict__ entries, size_t max_entries, int c, size_t B,       \
                                                                         uint32_t nlanes, XYZZ<Fp> *__restrict__ sums) {                                  \
        accumulate_body<Fp, DENSE, AccWords<Fp>>(table, bucket_start, entries, max_entries, c, B, nlanes, sums);                         \
    }
ZK_ACC_G1(true)
ZK_ACC_G1(false)
#undef ZK_ACC_G1
#endif  // ZK_EXPERIMENTS
// the 9-limb walks (the default; ZKPROVE_MSM_F9=0 keeps the 8 x 32 ones above for comparison)
#ifndef ZK_ACC9_WAVES
#define ZK_ACC9_WAVES 4
#endif
#define ZK_ACC9_BOUNDS(F) (sizeof(F) == 64 ? 2 : ZK_ACC9_WAVES)
template <class F, bool DENSE>
__global__ __launch_bounds__(64, ZK_ACC9_BOUNDS(F)) void k_msm_accumulate9(const Affine<F> *__restrict__ table, const uint32_t *__restrict__ bucket_start,
                                                                          const uint32_t *__restrict__ entries, size_t max_entries, int c, size_t B,
                                                                          uint32_t nlanes, typename Nine<F>::X9 *__restrict__ sums, const uint32_t *__restrict__ skip_if_zero) {
    accumulate_body<F, DENSE, Acc9<F>>(table, bucket_start, entries, max_entries, c, B, nlanes, sums, skip_if_zero);
}

#ifdef ZK_EXPERIMENTS
// thread (b, seg): running-sum reduction of buckets [seg*L, seg*L + L) of instance b
template <class F>
__global__ __launch_bounds__(64) void k_msm_bucket_reduce(const XYZZ<F> *__restrict__ sums, const uint32_t *__restrict__ bucket_start, int c, size_t B,
                                                          XYZZ<F> *__restrict__ part, size_t part_stride, size_t part_off, uint32_t L) {
    const uint32_t nb = 1u << (c - 1);
    const uint32_t nseg = nb / L;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nseg * B) return;
    const size_t b = t / nseg;
    const uint32_t seg = (uint32_t)(t % nseg);
    const uint32_t *bs = bucket_start + b * (nb + 1);
    const XYZZ<F> *in = sums + b * nb;
    XYZZ<F> run = XYZZ<F>::inf(), tot = XYZZ<F>::inf();
    const uint32_t s0 = seg * L;
    uint32_t hi = bs[s0 + L];
    for (int k = (int)L - 1; k >= 0; k--) {
        const uint32_t lo = bs[s0 + k];
        if (hi > lo) run.add_w(in[s0 + k]);     // both running sums in weak form (curve.cuh add_w)
        tot.add_w(run);
        hi = lo;
    }
    run = run.canon(); tot = tot.canon();
    // contribution = tot + s0 * run   (bucket index i carries weight i+1)
    if (s0 != 0 && !run.is_inf()) {
        XYZZ<F> m = XYZZ<F>::inf();
        for (int bit = 31 - __clz(s0); bit >= 0; bit--) {
            m = m.dbl();
            if ((s0 >> bit) & 1) m.add(run);
        }
        tot.add(m);
    }
    part[b * part_stride + part_off + seg] = tot;
}

#endif  // ZK_EXPERIMENTS
// the same running-sum reduction over the G1X9 bucket records of k_msm_accumulate9; the segment total leaves in the canonical
// 2^256 form, the rare s0 * run correction is done there as above
template <class F>
__global__ __launch_bounds__(64, ZK_FP2_WAVES(F)) void k_msm_bucket_reduce9(const typename Nine<F>::X9 *__restrict__ sums, const uint32_t *__restrict__ bucket_start,
                                                                           int c, size_t B, XYZZ<F> *__restrict__ part, size_t part_stride, size_t part_off,
                                                                           uint32_t L, const uint32_t *__restrict__ skip_if_zero) {
    using X9 = typename Nine<F>::X9;
    const uint32_t nb = 1u << (c - 1);
    const uint32_t nseg = nb / L;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nseg * B) return;
    const size_t b = t / nseg;
    const uint32_t seg = (uint32_t)(t % nseg);
    if (skip_if_zero && *skip_if_zero == 0) { part[b * part_stride + part_off + seg] = XYZZ<F>::inf(); return; }      // empty lists: the segment totals are the identity
    const uint32_t *bs = bucket_start + b * (nb + 1);
    const X9 *in = sums + b * nb;
    X9 run9 = X9::inf(), tot9 = X9::inf();
    const uint32_t s0 = seg * L;
    uint32_t hi = bs[s0 + L];
    for (int k = (int)L - 1; k >= 0; k--) {
        const uint32_t lo = bs[s0 + k];
        if (hi > lo) run9.add(in[s0 + k]);
        tot9.add(run9);
        hi = lo;
    }
    // contribution = tot + s0 * run   (bucket index i carries weight i+1), double-and-add on the same field
    if (s0 != 0 && !run9.is_inf()) {
        X9 m = X9::inf();
#pragma unroll 1
        for (int bit = 31 - __clz(s0); bit >= 0; bit--) {
            m = X9::dbl_of(m);
            if ((s0 >> bit) & 1) m.add(run9);
        }
        tot9.add(m);
    }
    part[b * part_stride + part_off + seg] = tot9.to_xyzz();
}

// thread (chunk, b): lanes = consecutive instances.  Scalars +-1 are summed here; scalars that are neither 0 nor
// +-1 are appended (index + canonical value) to the instance's compact list for the bucket pipeline.
// Two phases per thread: (1) classify the chunk's 120 scalars, one group of 6 (coalesced row reads) at a time, into the
// group's pattern index, (2) one mixed addition per non-empty group with the group's precomputed signed subset sum --
// 20 additions per wavefront and chunk where adding the +-1 bases one by one took ~77 (49 % of the ChaCha wire scalars
// are +-1, and a wavefront runs the maximum over its lanes).
#ifdef ZK_EXPERIMENTS
template <class F>
__global__ __launch_bounds__(256, ZK_FP2_WAVES(F)) void k_msm_ones(const Affine<F> *__restrict__ tern, const Fr *__restrict__ scalars, size_t B,
                                                  const uint32_t *__restrict__ rows, size_t n, XYZZ<F> *__restrict__ part,
                                                  size_t part_stride, size_t part_off, uint32_t *__restrict__ gen_count,
                                                  uint32_t *__restrict__ gen_idx, Fr *__restrict__ gen_val, const uint8_t *__restrict__ wide_mask) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nchunks = (n + ONES_CHUNK - 1) / ONES_CHUNK;
    if (t >= nchunks * B) return;
    const size_t b = t % B, chunk = t / B;
    const size_t j0 = chunk * ONES_CHUNK;
    const size_t j1 = j0 + ONES_CHUNK < n ? j0 + ONES_CHUNK : n;
    const Fr one = Fr::one(), mone = fr_minus_one();
    uint32_t pat[ONES_GROUPS];                 // pattern index per group (fully unrolled: stays in registers)
    uint32_t gen[4] = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < ONES_GROUPS; g++) {
        const size_t jb = j0 + (size_t)g * TERN_G;
        uint32_t idx = TERN_CENTER;
        if (jb < j1) {
            // the six scalar rows of a group are loaded together (unconditional loads, tail clamped)
            Fr sv[TERN_G];
#pragma unroll
            for (int q = 0; q < TERN_G; q++) {
                const size_t j = jb + q < j1 ? jb + q : j1 - 1;
                sv[q] = scalars[(size_t)rows[j] * B + b];
            }
            uint32_t p3 = 1;
#pragma unroll
            for (int q = 0; q < TERN_G; q++, p3 *= 3) {
                if (jb + q >= j1) break;
                const Fr &s = sv[q];
                if (s.is_zero() || (wide_mask && wide_mask[jb + q])) continue;      // a wire-comb position: not this kernel's
                if (s == one) idx += p3;
                else if (s == mone) idx -= p3;
                else { const uint32_t k = (uint32_t)(g * TERN_G + q); gen[k >> 5] |= 1u << (k & 31); }
            }
        }
        pat[g] = idx;
    }
    // wide scalars (6 % of the AES wires) are only MARKED above: appending them inside the scan would make every wavefront
    // pay an atomic + a Montgomery conversion in almost all of its iterations; here each lane walks just its own few
#pragma unroll
    for (int wd = 0; wd < 4; wd++) {
        uint32_t m = gen[wd];
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            const size_t j = j0 + wd * 32 + bit;
            const uint32_t q = atomicAdd(&gen_count[b], 1u);
            atomicAdd(&gen_count[B], 1u);                      // batch total: the bucket-pipeline kernels leave at once when it stays 0
            gen_idx[b * n + q] = (uint32_t)j;
            gen_val[b * n + q] = scalars[(size_t)rows[j] * B + b].from_mont();
        }
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    const Affine<F> *tg = tern + chunk * ONES_GROUPS * (size_t)TERN_ENTRIES;
    // group g's addend: entry |pat - centre| - 1, negated when pat < centre; the next group's is loaded under the addition
    auto fetch = [&](int g, uint32_t pt) -> Affine<F> {
        const uint32_t e = pt > TERN_CENTER ? pt - TERN_CENTER - 1 : TERN_CENTER - 1 - pt;
        return tg[(size_t)g * TERN_ENTRIES + e];
    };
    Affine<F> p = Affine<F>::inf();
    if (pat[0] != TERN_CENTER) p = fetch(0, pat[0]);
#pragma unroll
    for (int g = 0; g < ONES_GROUPS; g++) {
        Affine<F> pn = Affine<F>::inf();
        if (g + 1 < ONES_GROUPS && pat[g + 1] != TERN_CENTER) pn = fetch(g + 1, pat[g + 1]);
        if (pat[g] != TERN_CENTER) {
            if (pat[g] < TERN_CENTER) p.y = p.y.neg();
            acc.template madd_w<false>(p);      // weak form, shared out-of-line multiplier
        }
        p = pn;
    }
    acc = acc.canon();
    part[b * part_stride + part_off + chunk] = acc;
}

#endif  // ZK_EXPERIMENTS
// k_msm_ones on the 9-limb field (field9.cuh): same classification, the group patterns are parked in LDS so that the
// additions are ONE rolled loop around one expanded mixed addition (twenty unrolled copies would be 300 KB of code); tern holds the
// 2^261-Montgomery form (bases_build), the chunk total leaves in the canonical 2^256 form
template <class F>
__global__ __launch_bounds__(256, ZK_FP2_WAVES(F)) void k_msm_ones9(const Affine<F> *__restrict__ tern, const Fr *__restrict__ scalars, size_t B,
                                                                    const uint32_t *__restrict__ rows, size_t n, XYZZ<F> *__restrict__ part,
                                                                    size_t part_stride, size_t part_off, uint32_t *__restrict__ gen_count,
                                                                    uint32_t *__restrict__ gen_idx, Fr *__restrict__ gen_val, const uint8_t *__restrict__ wide_mask) {
    using F9 = typename Nine<F>::F9;
    using X9 = typename Nine<F>::X9;
    __shared__ uint16_t spat[ONES_GROUPS][256];
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nchunks = (n + ONES_CHUNK - 1) / ONES_CHUNK;
    if (t >= nchunks * B) return;
    const size_t b = t % B, chunk = t / B;
    const size_t j0 = chunk * ONES_CHUNK;
    const size_t j1 = j0 + ONES_CHUNK < n ? j0 + ONES_CHUNK : n;
    const Fr one = Fr::one(), mone = fr_minus_one();
    uint32_t gen[4] = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < ONES_GROUPS; g++) {
        const size_t jb = j0 + (size_t)g * TERN_G;
        uint32_t idx = TERN_CENTER;
        if (jb < j1) {
            Fr sv[TERN_G];
#pragma unroll
            for (int q = 0; q < TERN_G; q++) {
                const size_t j = jb + q < j1 ? jb + q : j1 - 1;
                sv[q] = scalars[(size_t)rows[j] * B + b];
            }
            uint32_t p3 = 1;
#pragma unroll
            for (int q = 0; q < TERN_G; q++, p3 *= 3) {
                if (jb + q >= j1) break;
                const Fr &sc = sv[q];
                if (sc.is_zero() || (wide_mask && wide_mask[jb + q])) continue;     // a wire-comb position: not this kernel's
                if (sc == one) idx += p3;
                else if (sc == mone) idx -= p3;
                else { const uint32_t k = (uint32_t)(g * TERN_G + q); gen[k >> 5] |= 1u << (k & 31); }
            }
        }
        spat[g][threadIdx.x] = (uint16_t)idx;          // own column only: no barrier needed
    }
#pragma unroll
    for (int wd = 0; wd < 4; wd++) {
        uint32_t m = gen[wd];
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            const size_t j = j0 + wd * 32 + bit;
            const uint32_t q = atomicAdd(&gen_count[b], 1u);
            atomicAdd(&gen_count[B], 1u);                      // batch total: the bucket-pipeline kernels leave at once when it stays 0
            gen_idx[b * n + q] = (uint32_t)j;
            gen_val[b * n + q] = scalars[(size_t)rows[j] * B + b].from_mont();
        }
    }
    X9 acc = X9::inf();
    const Affine<F> *tg = tern + chunk * ONES_GROUPS * (size_t)TERN_ENTRIES;
    auto fetch = [&](int g, uint32_t pt) -> Affine<F> {
        const uint32_t e = pt > TERN_CENTER ? pt - TERN_CENTER - 1 : TERN_CENTER - 1 - pt;
        return tg[(size_t)g * TERN_ENTRIES + e];
    };
    uint32_t pt = spat[0][threadIdx.x];
    Affine<F> p = Affine<F>::inf();
    if (pt != TERN_CENTER) p = fetch(0, pt);
#pragma unroll 1
    for (int g = 0; g < ONES_GROUPS; g++) {
        const uint32_t ptn = g + 1 < ONES_GROUPS ? spat[g + 1][threadIdx.x] : (uint32_t)TERN_CENTER;
        Affine<F> pn = Affine<F>::inf();
        if (ptn != TERN_CENTER) pn = fetch(g + 1, ptn);          // the next group's addend is loaded under the addition
        if (pt != TERN_CENTER && !p.is_inf()) acc.madd(F9::from_words(p.x), F9::cneg(F9::from_words(p.y), pt < TERN_CENTER ? -1 : 0));
        p = pn;
        pt = ptn;
    }
    part[b * part_stride + part_off + chunk] = acc.to_xyzz();
}

// instance-major copy of the element-wise product of two [rows][B] matrices (tiled through LDS):
// out[b * stride + i] = x[i * B + b] * y[i * B + b]   (quotient stage: f = A*B on the coset)
__global__ __launch_bounds__(256) void k_transpose_mul_fr(const Fr *__restrict__ x, const Fr *__restrict__ y, Fr *__restrict__ out, size_t rows, size_t B, size_t stride) {
    __shared__ uint4 tile[2][32][33];
    const size_t i0 = (size_t)blockIdx.x * 32, b0 = (size_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    uint4 *dst = (uint4 *)out;
    for (int r = ty; r < 32; r += 8) {
        const size_t i = i0 + r, b = b0 + tx;
        if (i < rows && b < B) {
            const Fr p = x[i * B + b] * y[i * B + b];
            tile[0][r][tx] = make_uint4(p.l[0], p.l[1], p.l[2], p.l[3]);
            tile[1][r][tx] = make_uint4(p.l[4], p.l[5], p.l[6], p.l[7]);
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const size_t b = b0 + r, i = i0 + tx;
        if (i < rows && b < B) { dst[(b * stride + i) * 2] = tile[0][tx][r]; dst[(b * stride + i) * 2 + 1] = tile[1][tx][r]; }
    }
}
void fr_transpose_mul(const Fr *x, const Fr *y, Fr *out, size_t rows, size_t B, size_t stride, hipStream_t s) {
    hipLaunchKernelGGL(k_transpose_mul_fr, dim3(ceil_div(rows, 32), ceil_div(B, 32)), dim3(256), 0, s, x, y, out, rows, B, stride);
}

// out[b*out_stride + g] = sum_{i < group} in[b*in_stride + g*group + i]   (i bounded by count)
template <class F>
__global__ __launch_bounds__(64, ZK_FP2_WAVES(F)) void k_sum_groups(const XYZZ<F> *__restrict__ in, size_t in_stride, size_t count, size_t group,
                                                   XYZZ<F> *__restrict__ out, size_t out_stride, size_t B) {
    const size_t ngroups = (count + group - 1) / group;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ngroups * B) return;
    const size_t b = t / ngroups, g = t % ngroups;
    XYZZ<F> acc = XYZZ<F>::inf();
    const size_t i0 = g * group, i1 = i0 + group < count ? i0 + group : count;
    for (size_t i = i0; i < i1; i++) acc.add(in[b * in_stride + i]);
    out[b * out_stride + g] = acc;
}

#ifdef ZK_EXPERIMENTS
// Dense (quotient) MSM: ZKPROVE_MSM_TREE=1 selects the affine reduction tree (msm_tree.cuh, gnark's batch-affine idea) instead of the
// extended-Jacobian walk.  It is bit-exact (tests) but measured SLOWER on MI355X (126 vs 110 ms per 1024-batch, profiles/r02_tree_*):
// Montgomery's trick needs two passes over every pair, which triples the random 64-byte traffic (4.4 G gathers + 100 GB of
// prefix rows per launch against 1.07 G gathers), and that, not the multiplier, then bounds the kernel.  Kept as an experiment.
static bool tree_enabled() {
    static const bool on = [] { const char *e = getenv("ZKPROVE_MSM_TREE"); return e && e[0] == '1'; }();
    return on;
}
// accumulate + bucket reduction of one sorted dense MSM through the tree kernels (G1)
static void launch_tree_g1(const MsmBasesG1 &mb, MsmScratch &sc, const uint32_t *bs_p, const uint32_t *ent_p, size_t max_entries, int c, size_t Bv, uint32_t nlanes,
                           G1XYZZ *part, size_t stride, size_t part_off, uint32_t L, hipStream_t s, hipEvent_t *seg_events) {
    const uint32_t nb = 1u << (c - 1);
    const size_t nseg = nb / L;
    const size_t capA = tree_capacity(max_entries, nb, nlanes, 1), capB = tree_capacity(max_entries, nb, nlanes, 2);
    static const uint32_t dbg = [] { const char *e = getenv("ZKPROVE_TREE_DEBUG"); return e ? (uint32_t)atoi(e) : 0u; }();
    const unsigned grid = (dbg & 8) ? (unsigned)(nlanes * Bv) : ceil_div((size_t)nlanes * Bv, 64);
    if (dbg & 8) sc.tree_scratch.ensure((size_t)grid * TREE_CH * 64 * tree_slot_bytes<Fp>());
    if (sc.treeA1.n < Bv * capA || sc.treeB1.n < Bv * capB || sc.fin1.n < Bv * (size_t)nb ||
        sc.tree_scratch.n < (size_t)grid * TREE_CH * 64 * tree_slot_bytes<Fp>())
        throw ZkError("msm tree scratch too small");
    if (dbg & 4) { ZK_HIP(hipMemsetAsync(sc.treeA1.p, 0, Bv * capA * sizeof(G1Affine), s)); ZK_HIP(hipMemsetAsync(sc.treeB1.p, 0, Bv * capB * sizeof(G1Affine), s)); }
    if (seg_events) ZK_HIP(hipEventRecord(seg_events[0], s));
    hipLaunchKernelGGL((k_msm_tree<Fp, true>), dim3(grid), dim3(64), 0, s, mb.table.p, bs_p, ent_p, max_entries, c, Bv, nlanes,
                       sc.treeA1.p, capA, sc.treeB1.p, capB, sc.fin1.p, sc.tree_scratch.p, dbg);
    if (seg_events) ZK_HIP(hipEventRecord(seg_events[1], s));
    hipLaunchKernelGGL(k_msm_bucket_reduce_tree<Fp>, dim3(ceil_div(nseg * Bv, 64)), dim3(64), 0, s, sc.fin1.p, mb.table.p, ent_p, max_entries,
                       bs_p, c, Bv, part, stride, part_off, L);
}

#endif  // ZK_EXPERIMENTS
// accumulation + bucket reduction of one sorted MSM: buckets -> nseg segment totals in part[part_off ..]
template <class F, bool DENSE>
static void launch_accumulate_reduce(const MsmBases<F> &mb, const uint32_t *bs_p, const uint32_t *ent_p, size_t max_entries, int c, size_t B, uint32_t nlanes,
                                     XYZZ<F> *sums, size_t sums_cap, XYZZ<F> *part, size_t stride, size_t part_off, uint32_t L, hipStream_t s,
                                     hipEvent_t *seg_events, const uint32_t *skip_if_zero = nullptr) {
    const uint32_t nb = 1u << (c - 1);
    const size_t nseg = nb / L;
    const dim3 grid(ceil_div((size_t)nlanes * B, 64)), rgrid(ceil_div(nseg * B, 64));
    if (mb.f9) {
        using X9 = typename Nine<F>::X9;
        if (sums_cap * sizeof(XYZZ<F>) < B * (size_t)nb * sizeof(X9)) throw ZkError("msm bucket-sum scratch too small");
        X9 *sums9 = reinterpret_cast<X9 *>(sums);
        if (seg_events) ZK_HIP(hipEventRecord(seg_events[0], s));
        hipLaunchKernelGGL((k_msm_accumulate9<F, DENSE>), grid, dim3(64), 0, s, mb.table.p, bs_p, ent_p, max_entries, c, B, nlanes, sums9, skip_if_zero);
        if (seg_events) ZK_HIP(hipEventRecord(seg_events[1], s));
        hipLaunchKernelGGL(k_msm_bucket_reduce9<F>, rgrid, dim3(64), 0, s, sums9, bs_p, c, B, part, stride, part_off, L, skip_if_zero);
        return;
    }
#ifdef ZK_EXPERIMENTS
    if (sums_cap < B * (size_t)nb) throw ZkError("msm bucket-sum scratch too small");
    if (seg_events) ZK_HIP(hipEventRecord(seg_events[0], s));
    hipLaunchKernelGGL((k_msm_accumulate<F, DENSE>), grid, dim3(64), 0, s, mb.table.p, bs_p, ent_p, max_entries, c, B, nlanes, sums);
    if (seg_events) ZK_HIP(hipEventRecord(seg_events[1], s));
    hipLaunchKernelGGL(k_msm_bucket_reduce<F>, rgrid, dim3(64), 0, s, sums, bs_p, c, B, part, stride, part_off, L);
#else
    throw ZkError("msm: base set without the 9-limb tables");
#endif
}

template <class F>
struct PartSel;
template <>
struct PartSel<Fp> { static G1XYZZ *get(MsmScratch &s) { return s.part1.p; } static size_t cap(MsmScratch &s) { return s.part1.n; } };
template <>
struct PartSel<Fp2> { static G2XYZZ *get(MsmScratch &s) { return s.part2.p; } static size_t cap(MsmScratch &s) { return s.part2.n; } };

template <class F>
struct SumsSel;
template <>
struct SumsSel<Fp> { static G1XYZZ *get(MsmScratch &s) { return s.sums1.p; } static size_t cap(MsmScratch &s) { return s.sums1.n; } };
template <>
struct SumsSel<Fp2> { static G2XYZZ *get(MsmScratch &s) { return s.sums2.p; } static size_t cap(MsmScratch &s) { return s.sums2.n; } };

template <class F>
struct VoutSel;
template <>
struct VoutSel<Fp> { static G1XYZZ *get(MsmScratch &s) { return s.vout1.p; } static size_t cap(MsmScratch &s) { return s.vout1.n; } };
template <>
struct VoutSel<Fp2> { static G2XYZZ *get(MsmScratch &s) { return s.vout2.p; } static size_t cap(MsmScratch &s) { return s.vout2.n; } };

// the dynamic-LDS limit of a kernel is a per-device attribute
static void partition_attr() {
    static PerDeviceOnce once;
    once.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_msm_partition, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)); });
}

// The same sum for batches of at most FOLD_TREE_MAX_B proofs as ONE launch: a workgroup per instance, thread t first adds the partial sums
// t, t + 256, ..., then eight rounds through LDS halve the 256 running sums -- 3 + 8 dependent additions for the ~860 partial sums of
// the AES G2 MSM where the four levels of k_sum_groups were 4 launches of 7 (1.0 ms of a one-proof call).  Same group element.
template <class F>
__global__ __launch_bounds__(256) void k_fold_parts_tree(const XYZZ<F> *__restrict__ part, size_t stride, size_t npart, XYZZ<F> *__restrict__ out) {
    extern __shared__ uint4 fold_lds_raw[];
    ZK_FOLD_SETPRIO();
    XYZZ<F> *lds = reinterpret_cast<XYZZ<F> *>(fold_lds_raw);
    const size_t b = blockIdx.x;
    const int t = threadIdx.x;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (size_t i = t; i < npart; i += 256) acc.add(part[b * stride + i]);
    for (int step = 128; step >= 1; step >>= 1) {
        if (t >= step && t < 2 * step) lds[t - step] = acc;
        __syncthreads();
        if (t < step) acc.add(lds[t]);
        __syncthreads();
    }
    if (t == 0) out[b] = acc;
}
// Dense two-level path of a small batch, every instance split into `split` virtual instances (G1 only).
template <class F>
static void msm_dense_split(const MsmBases<F> &mb, const Fr *scalars, size_t B, MsmScratch &sc, XYZZ<F> *out, hipStream_t s, hipEvent_t *seg_events,
                            uint32_t split) {
    if constexpr (sizeof(F) == 32) {
        const int c = mb.c, W = mb.W;
        const uint32_t nb = 1u << (c - 1);
        const size_t Bv = B * split, nv = mb.n / split, max_entries = nv * W;
        const uint32_t L = seg_len(nb, Bv);
        const size_t nseg = nb / L, npart = nseg;
        const size_t stride = part_stride_for(mb.n, c, Bv);
        XYZZ<F> *part = PartSel<F>::get(sc), *sums = SumsSel<F>::get(sc);
        if (sc.entries.n < Bv * max_entries + 1 || sc.entries2.n < Bv * max_entries + 1 || sc.bucket_start.n < Bv * (nb + 1) ||
            PartSel<F>::cap(sc) < Bv * stride + 1 || SumsSel<F>::cap(sc) < Bv * (size_t)nb || sc.coarse_base.n < Bv * (MAX_COARSE + 1) || sc.vout1.n < Bv)
            throw ZkError("msm scratch too small for the split dense path");
        const size_t lds2 = (size_t)(PART_THREADS / 64 * MAX_COARSE + (MAX_COARSE + 1) + 3 * MAX_COARSE + 2 * PART_THREADS * (size_t)W) * sizeof(uint32_t);
        partition_attr();
        hipLaunchKernelGGL(k_msm_partition, dim3((unsigned)Bv), dim3(PART_THREADS), lds2, s, scalars, mb.dense_stride, nv, c, W, sc.coarse_base.p,
                           sc.entries2.p, max_entries, split);
        hipLaunchKernelGGL(k_msm_binsort, dim3(1u << (c - 1 - FINE_BITS), (unsigned)Bv), dim3(256), 0, s, sc.coarse_base.p, sc.entries2.p, sc.entries.p,
                           max_entries, c, sc.bucket_start.p);
        const uint32_t nlanes = accumulate_lanes(max_entries, Bv);
#ifdef ZK_EXPERIMENTS
        if (tree_enabled() && sc.treeA1.n) {
            launch_tree_g1(mb, sc, sc.bucket_start.p, sc.entries.p, max_entries, c, Bv, nlanes, part, stride, (size_t)0, L, s, seg_events);
        } else
#endif
        {
            launch_accumulate_reduce<F, true>(mb, sc.bucket_start.p, sc.entries.p, max_entries, c, Bv, nlanes, sums, SumsSel<F>::cap(sc), part, stride, (size_t)0, L, s,
                                              seg_events);
        }
        if (Bv <= FOLD_TREE_MAX_B && tiny_batch(1) && npart > FOLD_GROUP) {
            // (a one-proof call: the 2 048 segment totals of each of its 16 virtual instances in one tree launch instead of four levels)
            static PerDeviceOnce attr;
            attr.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_fold_parts_tree<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (int)sizeof(XYZZ<F>))); });
            hipLaunchKernelGGL(k_fold_parts_tree<F>, dim3((unsigned)Bv), dim3(256), 128 * sizeof(XYZZ<F>), s, part, stride, npart, (XYZZ<F> *)sc.vout1.p);
        } else {
            size_t off = 0, cnt = npart;
            do {
                const size_t groups = (cnt + FOLD_GROUP - 1) / FOLD_GROUP;
                const bool last = groups == 1;
                hipLaunchKernelGGL(k_sum_groups<F>, dim3(ceil_div(groups * Bv, 64)), dim3(64), 0, s, part + off, stride, cnt, FOLD_GROUP,
                                   last ? (XYZZ<F> *)sc.vout1.p : part + off + cnt, last ? (size_t)1 : stride, Bv);
                off += cnt;
                cnt = groups;
            } while (cnt > 1);
        }
        if (Bv <= FOLD_TREE_MAX_B && tiny_batch(1) && split > 2)        // the virtual instances' totals: a tree again (15 dependent additions -> 4)
            hipLaunchKernelGGL(k_fold_parts_tree<F>, dim3((unsigned)B), dim3(256), 128 * sizeof(XYZZ<F>), s, (const XYZZ<F> *)sc.vout1.p, (size_t)split, (size_t)split, out);
        else
            hipLaunchKernelGGL(k_sum_groups<F>, dim3(ceil_div(B, 64)), dim3(64), 0, s, (const XYZZ<F> *)sc.vout1.p, (size_t)split, (size_t)split, (size_t)split, out,
                               (size_t)1, B);
        ZK_HIP(hipGetLastError());
    } else {
        (void)mb; (void)scalars; (void)B; (void)sc; (void)out; (void)s; (void)seg_events; (void)split;
        throw ZkError("split dense path is G1 only");
    }
}

// dense MSM, two-level sort applicable?  0 = no (single-level k_msm_sort<true>), 1 = yes, > 1 = yes with every instance of a small
// batch split into that many virtual instances
template <class F>
static uint32_t dense_split_factor(const MsmBases<F> &mb, size_t B, const MsmScratch &sc) {
    const int c = mb.c, W = mb.W;
    const size_t n = mb.n, max_entries = n * W;
    if (!(c - 1 > FINE_BITS && c - 1 - FINE_BITS <= 8 && W <= 24 && n * (size_t)W <= (1u << 21) && sc.entries2.n >= B * max_entries + 1)) return 0;
    uint32_t split = 1;
    if (sizeof(F) == 32 && sc.vout1.n > VIRT_MIN)
        while (split < DENSE_SPLIT_MAX && B * split * 2 <= VIRT_MIN && (n % (split * 2)) == 0 && n / (split * 2) >= 4096) split *= 2;
    return split;
}
template <class F>
static void dense_two_level_sort(const MsmBases<F> &mb, const Fr *scalars, size_t B, MsmScratch &sc, uint32_t *ent_p, uint32_t *bs_p, hipStream_t s) {
    const int c = mb.c, W = mb.W;
    const size_t n = mb.n, max_entries = n * W;
    const size_t lds2 = (size_t)(PART_THREADS / 64 * MAX_COARSE + (MAX_COARSE + 1) + 3 * MAX_COARSE + 2 * PART_THREADS * (size_t)W) * sizeof(uint32_t);
    partition_attr();
    sc.coarse_base.ensure(B * (MAX_COARSE + 1));
    hipLaunchKernelGGL(k_msm_partition, dim3((unsigned)B), dim3(PART_THREADS), lds2, s, scalars, mb.dense_stride, n, c, W, sc.coarse_base.p,
                       sc.entries2.p, max_entries, 1u);
    hipLaunchKernelGGL(k_msm_binsort, dim3(1u << (c - 1 - FINE_BITS), (unsigned)B), dim3(256), 0, s, sc.coarse_base.p, sc.entries2.p, ent_p,
                       max_entries, c, bs_p);
}
// Sort of the dense (quotient) MSM alone, on stream s, into the dense list buffers; the msm_batch call that follows (on a stream that
// has waited for s) then starts at the accumulation.  Returns false (and does nothing) when the batch takes another sort path.
bool msm_dense_presort_applicable(const MsmBasesG1 &mb, size_t B, const MsmScratch &sc) {
    const size_t max_entries = mb.n * (size_t)mb.W;
    const uint32_t nb = 1u << (mb.c - 1);
    return mb.dense && mb.n != 0 && dense_split_factor(mb, B, sc) == 1 && sc.entries_z.n >= B * max_entries + 1 && sc.bucket_start_z.n >= B * (size_t)(nb + 1);
}
bool msm_dense_presort(const MsmBasesG1 &mb, const Fr *scalars, size_t B, MsmScratch &sc, hipStream_t s, int slot) {
    const size_t max_entries = mb.n * (size_t)mb.W;
    const uint32_t nb = 1u << (mb.c - 1);
    if (!mb.dense || mb.n == 0 || dense_split_factor(mb, B, sc) != 1) return false;
    if (sc.entries_z.n < B * max_entries + 1 || sc.bucket_start_z.n < B * (size_t)(nb + 1)) return false;
    if (slot && (sc.entries_zb.n < B * max_entries + 1 || sc.bucket_start_zb.n < B * (size_t)(nb + 1))) return false;
    dense_two_level_sort(mb, scalars, B, sc, sc.ez(slot), sc.bz(slot), s);
    ZK_HIP(hipGetLastError());
    sc.presorted = true;          // for the msm_batch call that follows; a caller that sorts batches ahead sets presorted / zslot itself
    sc.zslot = slot;
    return true;
}

// fold the npart partial sums of every instance, FOLD_GROUP at a time; the last level writes out[b]
template <class F>
static void msm_fold_parts(XYZZ<F> *part, size_t stride, size_t npart, XYZZ<F> *out, size_t B, hipStream_t s) {
    if (B <= FOLD_TREE_MAX_B && tiny_batch(1) && npart > FOLD_GROUP) {
        static PerDeviceOnce attr;
        attr.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_fold_parts_tree<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (int)sizeof(XYZZ<F>))); });
        hipLaunchKernelGGL(k_fold_parts_tree<F>, dim3((unsigned)B), dim3(256), 128 * sizeof(XYZZ<F>), s, part, stride, npart, out);
        ZK_HIP(hipGetLastError());
        return;
    }
    size_t off = 0, cnt = npart;
    do {
        const size_t groups = (cnt + FOLD_GROUP - 1) / FOLD_GROUP;
        const bool last = groups == 1;
        hipLaunchKernelGGL(k_sum_groups<F>, dim3(ceil_div(groups * B, 64)), dim3(64), 0, s, part + off, stride, cnt, FOLD_GROUP,
                           last ? out : part + off + cnt, last ? (size_t)1 : stride, B);
        off += cnt;
        cnt = groups;
    } while (cnt > 1);
    ZK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------- wire comb
template <class F>
static void wire_comb_needs(const MsmBases<F> &mb, size_t B, size_t &pat_words, size_t &acc_bytes, size_t &r1_bytes) {
    pat_words = acc_bytes = r1_bytes = 0;
    for (auto &cb : mb.wcomb) {
        pat_words = std::max(pat_words, comb_pattern_words(*cb, B));
        acc_bytes = std::max(acc_bytes, comb_acc_bytes(*cb, B, sizeof(typename Nine<F>::X9)));
        r1_bytes = std::max(r1_bytes, comb_r1_bytes(*cb, B, sizeof(typename Nine<F>::X9)));
    }
}
void msm_wire_comb_needs(const MsmBasesG1 &mb, size_t B, size_t &p, size_t &a, size_t &r) { wire_comb_needs<Fp>(mb, B, p, a, r); }
void msm_wire_comb_needs(const MsmBasesG2 &mb, size_t B, size_t &p, size_t &a, size_t &r) { wire_comb_needs<Fp2>(mb, B, p, a, r); }

template <class F>
__global__ __launch_bounds__(256) void k_gather_points(const Affine<F> *__restrict__ pts, const uint32_t *__restrict__ idx, size_t n, Affine<F> *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pts[idx[i]];
}
template <class F>
static bool attach_wire_comb(MsmBases<F> &mb, const uint8_t *cls, int k, hipStream_t s) {
    static const int PLANES[WCOMB_CLASSES] = {8, 32, 254};
    mb.wcomb.clear();
    mb.wide_mask.release();
    if (!mb.n || mb.dense || !mb.f9 || !mb.pts.n) return true;
    std::vector<uint32_t> rows(mb.n);
    ZK_HIP(hipMemcpy(rows.data(), mb.scalar_row.p, mb.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint8_t> mask(mb.n, 0);
    std::vector<std::unique_ptr<CombBasesT<F>>> built;
    for (int c = 0; c < WCOMB_CLASSES; c++) {
        std::vector<uint32_t> idx, crow;
        for (size_t j = 0; j < mb.n; j++)
            if (cls[j] == c + 1) { idx.push_back((uint32_t)j); crow.push_back(rows[j]); mask[j] = 1; }
        if (idx.empty()) continue;
        std::unique_ptr<CombBasesT<F>> cb(new CombBasesT<F>());
        DevBuf<uint32_t> didx(idx.size());
        DevBuf<Affine<F>> sel(idx.size());
        didx.upload(idx.data(), idx.size(), s);
        hipLaunchKernelGGL(k_gather_points<F>, dim3(ceil_div(idx.size(), 256)), dim3(256), 0, s, mb.pts.p, didx.p, idx.size(), sel.p);
        const int kk = (int)std::min<size_t>((size_t)k, std::max<size_t>(2, idx.size()));
        if (!comb_build_planes(*cb, sel.p, idx.size(), kk, PLANES[c], s)) return false;
        cb->rows.upload(crow.data(), crow.size(), s);
        cb->base_index.upload(idx.data(), idx.size(), s);
        ZK_HIP(hipStreamSynchronize(s));
        built.push_back(std::move(cb));
    }
    if (built.empty()) return true;
    mb.wide_mask.upload(mask.data(), mask.size(), s);
    ZK_HIP(hipStreamSynchronize(s));
    mb.wcomb = std::move(built);
    return true;
}
bool msm_attach_wire_comb(MsmBasesG1 &mb, const uint8_t *cls, int k, hipStream_t s) { return attach_wire_comb<Fp>(mb, cls, k, s); }
bool msm_attach_wire_comb(MsmBasesG2 &mb, const uint8_t *cls, int k, hipStream_t s) { return attach_wire_comb<Fp2>(mb, cls, k, s); }

// Wire MSM of a one-proof call (G1): the scalars that are neither 0 nor +-1 -- the byte- and word-valued wires of the AES circuits -- crowd
// into a few buckets (24 425 committed bytes into 255 of 4 096: ~96 dependent additions for the lane that owns such a bucket, 0.9-1.9 ms
// of k_msm_accumulate9 in every chain of the call).  As for the quotient MSM of small batches (msm_dense_split) every instance becomes
// SPARSE_SPLIT virtual instances: a sixteenth of the compact list each to its own sort / accumulate / reduce rows, the totals joined by
// the tree launch.  The +-1 sums stay with the real instance (virtual instance 0's first slots).  Same group element.
template <class F>
static bool msm_sparse_split(const MsmBases<F> &mb, const Fr *scalars, size_t B, MsmScratch &sc, XYZZ<F> *out, hipStream_t s) {
    {
        const size_t n = mb.n, Bv = B * SPARSE_SPLIT;
        const int c = mb.c, W = mb.W;
        const uint32_t nb = 1u << (c - 1);
        const size_t max_entries_v = ((n + SPARSE_SPLIT - 1) / SPARSE_SPLIT) * (size_t)W;      // a share holds at most ceil(n / split) scalars
        const uint32_t L = seg_len(nb, Bv);
        const size_t nseg = nb / L, nchunks = (n + ONES_CHUNK - 1) / ONES_CHUNK, npart = nchunks + nseg;
        const size_t stride = part_stride_for(n, c, Bv);
        using X9 = typename Nine<F>::X9;
        if (npart > stride || sc.entries.n < Bv * max_entries_v + 1 || sc.bucket_start.n < Bv * (size_t)(nb + 1) || PartSel<F>::cap(sc) < Bv * stride + 1 ||
            SumsSel<F>::cap(sc) * sizeof(XYZZ<F>) < Bv * (size_t)nb * sizeof(X9) || VoutSel<F>::cap(sc) < Bv || sc.gen_count.n < B + 1 || sc.gen_idx.n < B * n ||
            sc.gen_val.n < B * n)
            return false;                                    // scratch of another shape: the caller takes the ordinary path
        XYZZ<F> *part = PartSel<F>::get(sc);
        ZK_HIP(hipMemsetAsync(part, 0, Bv * stride * sizeof(XYZZ<F>), s));              // the +-1 slots of the virtual instances 1.. : the identity
        ZK_HIP(hipMemsetAsync(sc.gen_count.p, 0, (B + 1) * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_msm_ones9<F>, dim3(ceil_div(nchunks * B, 256)), dim3(256), 0, s, mb.tern.p, scalars, B, mb.scalar_row.p, n, part, stride * SPARSE_SPLIT,
                           (size_t)0, sc.gen_count.p, sc.gen_idx.p, sc.gen_val.p, (const uint8_t *)nullptr);
        const size_t lds = (size_t)(nb + SORT_THREADS) * sizeof(uint32_t);
        const uint32_t *skip = sc.gen_count.p + B;
        hipLaunchKernelGGL(k_msm_sort<false>, dim3((unsigned)Bv), dim3(SORT_THREADS), lds, s, (const Fr *)nullptr, (size_t)0, sc.gen_count.p, sc.gen_idx.p, sc.gen_val.p, n, c,
                           W, sc.bucket_start.p, sc.entries.p, max_entries_v, skip, SPARSE_SPLIT);
        launch_accumulate_reduce<F, false>(mb, sc.bucket_start.p, sc.entries.p, max_entries_v, c, Bv, accumulate_lanes(max_entries_v, Bv), SumsSel<F>::get(sc),
                                           SumsSel<F>::cap(sc), part, stride, nchunks, L, s, nullptr, skip);
        msm_fold_parts<F>(part, stride, npart, VoutSel<F>::get(sc), Bv, s);
        hipLaunchKernelGGL(k_fold_parts_tree<F>, dim3((unsigned)B), dim3(256), 128 * sizeof(XYZZ<F>), s, (const XYZZ<F> *)VoutSel<F>::get(sc), (size_t)SPARSE_SPLIT,
                           (size_t)SPARSE_SPLIT, out);
        ZK_HIP(hipGetLastError());
        return true;
    }
}

template <class F>
static void msm_batch_impl(const MsmBases<F> &mb, const Fr *scalars, size_t B, MsmScratch &sc, XYZZ<F> *out, hipStream_t s, hipEvent_t *seg_events) {
    const size_t n = mb.n;
    const int c = mb.c, W = mb.W;
    const uint32_t nb = 1u << (c - 1);
    const uint32_t L = seg_len(nb, B);
    const size_t nseg = nb / L;
    const size_t nchunks = mb.dense ? 0 : (n + ONES_CHUNK - 1) / ONES_CHUNK;      // the dense MSM has no conditional-sum partials
    const size_t max_entries = n * W;
    // wire combs: batches that fill whole wavefronts, scratch reserved (Algorithm::wire_comb_prepare), 9-limb tables
    bool wcomb = !mb.dense && !mb.wcomb.empty() && mb.f9 && B >= WCOMB_MIN_B;
    if (wcomb) {
        size_t pw = 0, ab = 0, rb = 0;
        wire_comb_needs(mb, B, pw, ab, rb);
        wcomb = sc.wire_comb.pat[0].n >= pw && sc.wire_comb.acc.n >= ab && sc.wire_comb.r1.n >= rb;
    }
    const size_t nw = wcomb ? mb.wcomb.size() : 0;
    const size_t npart = nchunks + nseg + nw;
    const size_t stride = part_stride_for(n, c, B);      // level-0 partials followed by the fold levels
    if (sc.entries.n < B * max_entries + 1 || sc.bucket_start.n < B * (nb + 1) || PartSel<F>::cap(sc) < B * stride + 1)
        throw ZkError("msm scratch too small");
    XYZZ<F> *part = PartSel<F>::get(sc);
    if (n == 0) {
        ZK_HIP(hipMemsetAsync(out, 0, B * sizeof(XYZZ<F>), s));
        return;
    }
    size_t lds = (size_t)(nb + SORT_THREADS) * sizeof(uint32_t);
    static PerDeviceOnce attr_set;
    attr_set.run([] {
        ZK_HIP(hipFuncSetAttribute((const void *)k_msm_sort<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        ZK_HIP(hipFuncSetAttribute((const void *)k_msm_sort<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    });
    if (!mb.dense && mb.f9 && !wcomb && B <= SPARSE_SPLIT_MAX_B && tiny_batch(1) && msm_sparse_split(mb, scalars, B, sc, out, s)) return;
    // the sorted list of the dense MSM lives in buffers of its own when they exist (entries_z): its sort may then run on another
    // stream beside the wire MSMs (msm_dense_presort), which use entries / bucket_start
    uint32_t *ent_p = sc.entries.p, *bs_p = sc.bucket_start.p;
    if (mb.dense && sc.entries_z.n >= B * max_entries + 1 && sc.bucket_start_z.n >= B * (size_t)(nb + 1)) {
        const int slot = sc.presorted ? sc.zslot : 0;
        ent_p = sc.ez(slot); bs_p = sc.bz(slot);
    }
    if (mb.dense) {
        // scalars = instance-major [B][dense_stride] copy; no +-1 fast path (every quotient coefficient is wide)
        const uint32_t split = dense_split_factor(mb, B, sc);
        if (sc.presorted) {
            if (ent_p != sc.ez(sc.zslot) || split != 1) throw ZkError("msm: presorted dense list without its buffers");
            sc.presorted = false;                        // sorted on another stream; the caller has joined it into s
        } else if (split >= 1) {
            // small batch: split every instance into `split` virtual instances (see reserve_dense); everything downstream just sees
            // B * split instances of n / split scalars, and the split results are added at the end
            if (split > 1) {
                msm_dense_split(mb, scalars, B, sc, out, s, seg_events, split);
                return;
            }
            dense_two_level_sort(mb, scalars, B, sc, ent_p, bs_p, s);
        } else {
            hipLaunchKernelGGL(k_msm_sort<true>, dim3((unsigned)B), dim3(SORT_THREADS), lds, s, scalars, mb.dense_stride, (const uint32_t *)nullptr,
                               (const uint32_t *)nullptr, (const Fr *)nullptr, n, c, W, bs_p, ent_p, max_entries, (const uint32_t *)nullptr);
        }
    } else {
        if (sc.gen_count.n < B + 1 || sc.gen_idx.n < B * n || sc.gen_val.n < B * n) throw ZkError("msm compact-list scratch too small");
        ZK_HIP(hipMemsetAsync(sc.gen_count.p, 0, (B + 1) * sizeof(uint32_t), s));      // [B] = the batch total
        if (mb.f9)
            hipLaunchKernelGGL(k_msm_ones9<F>, dim3(ceil_div(nchunks * B, 256)), dim3(256), 0, s, mb.tern.p, scalars, B, mb.scalar_row.p, n, part, stride,
                               (size_t)0, sc.gen_count.p, sc.gen_idx.p, sc.gen_val.p, wcomb ? mb.wide_mask.p : (const uint8_t *)nullptr);
#ifdef ZK_EXPERIMENTS
        else
            hipLaunchKernelGGL(k_msm_ones<F>, dim3(ceil_div(nchunks * B, 256)), dim3(256), 0, s, mb.tern.p, scalars, B, mb.scalar_row.p, n, part,
                               stride, (size_t)0, sc.gen_count.p, sc.gen_idx.p, sc.gen_val.p, (const uint8_t *)nullptr);
#else
        else throw ZkError("msm: base set without the 9-limb tables");
#endif
        // wire-comb positions: patterns (values too wide for their class join the compact list) -> walk -> fold into a partial-sum slot
        for (size_t ci = 0; ci < nw; ci++) {
            const CombBasesT<F> &cb = *mb.wcomb[ci];
            comb_patterns_rows(cb, scalars, B, sc.wire_comb.pat[0].p, sc.gen_count.p, sc.gen_idx.p, sc.gen_val.p, n, s);
            comb_msm_strided(cb, sc.wire_comb.pat[0].p, B, sc.wire_comb, part + nchunks + nseg + ci, stride, s);
        }
        hipLaunchKernelGGL(k_msm_sort<false>, dim3((unsigned)B), dim3(SORT_THREADS), lds, s, (const Fr *)nullptr, (size_t)0, sc.gen_count.p, sc.gen_idx.p,
                           sc.gen_val.p, n, c, W, sc.bucket_start.p, sc.entries.p, max_entries, mb.f9 ? sc.gen_count.p + B : (const uint32_t *)nullptr);
    }
    const uint32_t nlanes = accumulate_lanes(max_entries, B);
#ifdef ZK_EXPERIMENTS
    if constexpr (sizeof(F) == 32) {
        if (mb.dense && tree_enabled() && sc.treeA1.n) {
            launch_tree_g1(mb, sc, bs_p, ent_p, max_entries, c, B, nlanes, part, stride, nchunks, L, s, seg_events);
            msm_fold_parts<F>(part, stride, npart, out, B, s);
            return;
        }
    }
#endif
    XYZZ<F> *sums = SumsSel<F>::get(sc);
    if (mb.dense) launch_accumulate_reduce<F, true>(mb, bs_p, ent_p, max_entries, c, B, nlanes, sums, SumsSel<F>::cap(sc), part, stride, nchunks, L, s, seg_events);
    else launch_accumulate_reduce<F, false>(mb, sc.bucket_start.p, sc.entries.p, max_entries, c, B, nlanes, sums, SumsSel<F>::cap(sc), part, stride, nchunks, L, s,
                                            seg_events, mb.f9 ? sc.gen_count.p + B : (const uint32_t *)nullptr);
    msm_fold_parts<F>(part, stride, npart, out, B, s);
}
void msm_batch(const MsmBasesG1 &mb, const Fr *sc, size_t B, MsmScratch &scr, G1XYZZ *out, hipStream_t s, hipEvent_t *ev) { msm_batch_impl<Fp>(mb, sc, B, scr, out, s, ev); }
void msm_batch(const MsmBasesG2 &mb, const Fr *sc, size_t B, MsmScratch &scr, G2XYZZ *out, hipStream_t s) { msm_batch_impl<Fp2>(mb, sc, B, scr, out, s, nullptr); }

// the stride a batch of B uses: one-proof batches cut the bucket reduction finer (seg_len) and need about twice the room per instance,
// which the reservations give them (part1 is sized for at least VIRT_MIN instances, part2 for twice the instances up to TINY_BATCH)
static size_t part_stride_for(size_t n, int c, size_t B) {
    if (!tiny_batch(B)) return msm_part_stride(n, c);
    const uint32_t nb = 1u << (c - 1);
    const uint32_t L = seg_len(nb, B);
    size_t npart = (n + ONES_CHUNK - 1) / ONES_CHUNK + nb / L + WCOMB_CLASSES, total = 0;
    for (size_t cnt = npart; cnt > 1; cnt = (cnt + FOLD_GROUP - 1) / FOLD_GROUP) total += cnt;
    return total + 2;
}
size_t msm_part_stride(size_t n, int c) {
    const uint32_t nb = 1u << (c - 1);
    const uint32_t L = nb > (1u << SEG_LOG_SMALL) ? (1u << SEG_LOG_SMALL) : nb;     // the small-batch segment length gives the most partials
    size_t npart = (n + ONES_CHUNK - 1) / ONES_CHUNK + nb / L + WCOMB_CLASSES, total = 0;      // + one slot per wire-comb class
    for (size_t cnt = npart; cnt > 1; cnt = (cnt + FOLD_GROUP - 1) / FOLD_GROUP) total += cnt;
    return total + 2;
}

// ------------------------------------------------------------------------ point import / export
static constexpr uint32_t P_PLUS1_DIV4[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
static constexpr uint32_t HALF_P[8] = {0x6c3e7ea3u, 0x9e10460bu, 0xb438e546u, 0xcbc0b548u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};

__device__ __forceinline__ Fp fp_from_be(const uint8_t *p, bool mask_flags) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t *q = p + 28 - 4 * i;
        r.l[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
    if (mask_flags) r.l[7] &= 0x3fffffffu;
    return r;
}
__device__ __forceinline__ bool fp_lex_largest(const Fp &y_mont) {
    Fp c = y_mont.from_mont();
    for (int i = 7; i >= 0; i--) {
        if (c.l[i] > HALF_P[i]) return true;
        if (c.l[i] < HALF_P[i]) return false;
    }
    return false;
}
__device__ __forceinline__ bool fp_sqrt(const Fp &a, Fp &r) {
    r = a.pow(P_PLUS1_DIV4, 254);
    return r.sqr() == a;
}
__device__ bool fp2_sqrt(const Fp2 &a, Fp2 &r) {
    if (a.is_zero()) { r = Fp2::zero(); return true; }
    if (a.a1.is_zero()) {
        Fp s;
        if (fp_sqrt(a.a0, s)) { r = Fp2{s, Fp::zero()}; return true; }
        if (fp_sqrt(a.a0.neg(), s)) { r = Fp2{Fp::zero(), s}; return true; }
        return false;
    }
    Fp n;
    if (!fp_sqrt(a.a0.sqr() + a.a1.sqr(), n)) return false;
    const Fp inv2 = Fp::from_u32(2).inv();
    for (int k = 0; k < 2; k++) {
        Fp t = (k == 0 ? a.a0 + n : a.a0 - n) * inv2;
        Fp x0;
        if (!fp_sqrt(t, x0) || x0.is_zero()) continue;
        Fp x1 = a.a1 * x0.dbl().inv();
        Fp2 cand{x0, x1};
        if (cand.sqr() == a) { r = cand; return true; }
    }
    return false;
}

__global__ void k_g1_from_be(G1Affine *out, const uint8_t *in, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = G1Affine{fp_from_be(in + 64 * i, false).to_mont(), fp_from_be(in + 64 * i + 32, false).to_mont()};
}
__global__ void k_g2_from_be(G2Affine *out, const uint8_t *in, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = in + 128 * i;
    G2Affine r;
    r.x.a1 = fp_from_be(p, false).to_mont(); r.x.a0 = fp_from_be(p + 32, false).to_mont();
    r.y.a1 = fp_from_be(p + 64, false).to_mont(); r.y.a0 = fp_from_be(p + 96, false).to_mont();
    out[i] = r;
}
// gnark-crypto marshal flags: 0b10 smallest y, 0b11 largest y, 0b01 infinity
__global__ void k_g1_decompress(G1Affine *out, const uint8_t *in, size_t n, uint32_t *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = in + 32 * i;
    const uint32_t flag = p[0] >> 6;
    if (flag == 1) { out[i] = G1Affine::inf(); return; }
    if (flag == 0) { atomicAdd(bad, 1u); out[i] = G1Affine::inf(); return; }
    const Fp xc = fp_from_be(p, true);
    if (Fp::geq_mod(xc.l)) { atomicAdd(bad, 1u); out[i] = G1Affine::inf(); return; }       // non-reduced coordinate
    Fp x = xc.to_mont();
    Fp y;
    if (!fp_sqrt(x.sqr() * x + Fp::from_u32(3), y)) { atomicAdd(bad, 1u); out[i] = G1Affine::inf(); return; }
    if (fp_lex_largest(y) != (flag == 3)) y = y.neg();
    out[i] = G1Affine{x, y};
}
__global__ void k_g2_decompress(G2Affine *out, const uint8_t *in, size_t n, uint32_t *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = in + 64 * i;
    const uint32_t flag = p[0] >> 6;
    if (flag == 1) { out[i] = G2Affine::inf(); return; }
    if (flag == 0) { atomicAdd(bad, 1u); out[i] = G2Affine::inf(); return; }
    const Fp x0c = fp_from_be(p + 32, false), x1c = fp_from_be(p, true);
    if (Fp::geq_mod(x0c.l) || Fp::geq_mod(x1c.l)) { atomicAdd(bad, 1u); out[i] = G2Affine::inf(); return; }
    Fp2 x{x0c.to_mont(), x1c.to_mont()};
    // twist coefficient b' = 3/(9+u)
    const Fp2 xi{Fp::from_u32(9), Fp::one()};
    const Fp2 bt = Fp2{Fp::from_u32(3), Fp::zero()} * xi.inv();
    Fp2 y;
    if (!fp2_sqrt(x.sqr() * x + bt, y)) { atomicAdd(bad, 1u); out[i] = G2Affine::inf(); return; }
    const bool largest = y.a1.is_zero() ? fp_lex_largest(y.a0) : fp_lex_largest(y.a1);
    if (largest != (flag == 3)) y = y.neg();
    out[i] = G2Affine{x, y};
}
__global__ void k_g1_to_affine_canon(const G1XYZZ *in, G1Affine *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a = in[i].to_affine();
    out[i] = G1Affine{a.x.from_mont(), a.y.from_mont()};
}
__global__ void k_g2_to_affine_canon(const G2XYZZ *in, G2Affine *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G2Affine a = in[i].to_affine();
    out[i] = G2Affine{a.x.from_mont(), a.y.from_mont()};
}
// Point validation of a decoded key slice, as gnark-crypto's decoder does in ProvingKey.ReadFrom (impl/library.go:104-108):
// coordinates reduced, point on the curve, and for G2 (cofactor != 1) membership of the order-r subgroup ([r]P = O).
// d_raw (optional): the big-endian source coordinates of an uncompressed slice, to reject non-reduced encodings.
__device__ __forceinline__ bool be_reduced(const uint8_t *p) {
    const Fp v = fp_from_be(p, false);
    return !Fp::geq_mod(v.l);
}
__global__ void k_g1_validate(const G1Affine *pts, const uint8_t *raw, size_t n, uint32_t *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = pts[i];
    bool ok = true;
    if (raw) ok = be_reduced(raw + 64 * i) && be_reduced(raw + 64 * i + 32);
    if (ok && !p.is_inf()) ok = p.y.sqr() == p.x.sqr() * p.x + Fp::from_u32(3);
    if (!ok) atomicAdd(bad, 1u);
}
__global__ __launch_bounds__(64) void k_g2_validate(const G2Affine *pts, const uint8_t *raw, size_t n, uint32_t *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G2Affine p = pts[i];
    bool ok = true;
    if (raw)
        for (int k = 0; k < 4; k++) ok = ok && be_reduced(raw + 128 * i + 32 * k);
    if (ok && !p.is_inf()) {
        const Fp2 xi{Fp::from_u32(9), Fp::one()};
        const Fp2 bt = Fp2{Fp::from_u32(3), Fp::zero()} * xi.inv();
        ok = p.y.sqr() == p.x.sqr() * p.x + bt;
        if (ok) ok = g2_in_subgroup(p);          // [r]P = O by gnark-crypto's own test (g2_subgroup.cuh)
    }
    if (!ok) atomicAdd(bad, 1u);
}
void g1_validate(const G1Affine *d_pts, const uint8_t *d_raw, size_t n, uint32_t *d_bad, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_g1_validate, dim3(ceil_div(n, 128)), dim3(128), 0, s, d_pts, d_raw, n, d_bad);
}
void g2_validate(const G2Affine *d_pts, const uint8_t *d_raw, size_t n, uint32_t *d_bad, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_g2_validate, dim3(ceil_div(n, 64)), dim3(64), 0, s, d_pts, d_raw, n, d_bad);
}
void g1_from_be_bytes(G1Affine *o, const uint8_t *in, size_t n, hipStream_t s) { if (n) hipLaunchKernelGGL(k_g1_from_be, dim3(ceil_div(n, 256)), dim3(256), 0, s, o, in, n); }
void g2_from_be_bytes(G2Affine *o, const uint8_t *in, size_t n, hipStream_t s) { if (n) hipLaunchKernelGGL(k_g2_from_be, dim3(ceil_div(n, 256)), dim3(256), 0, s, o, in, n); }
void g1_decompress(G1Affine *o, const uint8_t *in, size_t n, uint32_t *bad, hipStream_t s) { if (n) hipLaunchKernelGGL(k_g1_decompress, dim3(ceil_div(n, 128)), dim3(128), 0, s, o, in, n, bad); }
void g2_decompress(G2Affine *o, const uint8_t *in, size_t n, uint32_t *bad, hipStream_t s) { if (n) hipLaunchKernelGGL(k_g2_decompress, dim3(ceil_div(n, 128)), dim3(128), 0, s, o, in, n, bad); }
void g1_to_affine_canonical(const G1XYZZ *in, G1Affine *o, size_t n, hipStream_t s) { if (n) hipLaunchKernelGGL(k_g1_to_affine_canon, dim3(ceil_div(n, 128)), dim3(128), 0, s, in, o, n); }
void g2_to_affine_canonical(const G2XYZZ *in, G2Affine *o, size_t n, hipStream_t s) { if (n) hipLaunchKernelGGL(k_g2_to_affine_canon, dim3(ceil_div(n, 128)), dim3(128), 0, s, in, o, n); }

}  // namespace zk
