// Batched radix-2^K number-theoretic transforms over Fr for gfx950, and the Groth16 quotient
// pipeline ("computeH").
//
// Replaces gnark v0.14 backend/groth16/bn254 computeH + gnark-crypto v0.19.2 fr/fft Domain.FFT /
// FFTInverse (DIF / DIT kernels, coset shift = 5) -- the "13.1 %" slice of the reference prove
// profile, reached from gnark/libraries/prover/impl/provers.go:179,269.
//
// Data layout in HBM: element i of instance b lives at data[i * B + b] (32-byte Fr, Montgomery).
// Consecutive lanes of a wavefront are consecutive *instances*, so every global access is a
// contiguous 64 x 32 B = 2 KiB segment and every twiddle is wave-uniform.  A thread keeps 2^K
// elements in VGPRs and performs K butterfly stages per HBM round trip.
//
// All butterflies work in WEAK form (field.cuh: values in [0, 2r), no conditional subtraction in the multiplier): the passes
// read and write weak values, and whatever consumes a transform multiplies first (products A*B, the 1/n scaling, from_mont),
// which returns to canonical form.
// Like gnark we never run a bit-reversal pass: inverse transforms are DIF (natural in, bit-reversed
// out), forward transforms are DIT (bit-reversed in, natural out).  The prover only runs the pair
// inverse + coset-forward on a and b (to_coset); the rest of computeH lives in the folded key (keyfold.hip).
// compute_h (all 7 transforms) is kept as the kernel-level entry point zkp_compute_h of the experiments build (-DZK_EXPERIMENTS) for parity tests.
#include "kernels.h"
#include "field9.cuh"

namespace zk {

static constexpr uint32_t R_MINUS_1[8] = {0xf0000000u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};

__device__ __forceinline__ Fr fr_pow_u64(Fr base, uint64_t e) {
    Fr acc = Fr::one();
    while (e) {
        if (e & 1) acc = acc * base;
        base = base.sqr();
        e >>= 1;
    }
    return acc;
}
__device__ __forceinline__ uint32_t bitrev(uint32_t v, int bits) { return __brev(v) >> (32 - bits); }

// consts: [0]=w [1]=w^-1 [2]=g^-1 [3]=1/n [4]=1/(g^n-1) [5]=g
__global__ void k_ntt_consts(Fr *consts, int logn) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = R_MINUS_1[i];
    for (int s = 0; s < logn; s++) {
        for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0u);
    }
    Fr g = Fr::from_u32(5);
    Fr w = g.pow(e, 256);
    consts[0] = w;
    consts[1] = w.inv();
    consts[2] = g.inv();
    Fr n = Fr::from_u32(1u << logn);
    consts[3] = n.inv();
    Fr gn = g;
    for (int s = 0; s < logn; s++) gn = gn.sqr();
    consts[4] = (gn - Fr::one()).inv();
    consts[5] = g;
}
__global__ void k_ntt_tables(const Fr *consts, int logn, Fr *tw_fwd, Fr *tw_inv, Fr *coset_pre, Fr *coset_post) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)1 << logn;
    if (i >= n) return;
    if (i < n / 2 || n == 1) {
        tw_fwd[i] = fr_pow_u64(consts[0], i);
        tw_inv[i] = fr_pow_u64(consts[1], i);
    }
    uint32_t r = logn ? bitrev((uint32_t)i, logn) : 0;
    coset_pre[i] = fr_pow_u64(consts[5], r) * consts[3];
    coset_post[i] = fr_pow_u64(consts[2], r) * consts[3];
}

// out = 2^5 * in: the 2^261-Montgomery form of a multiplier table (canonical words)
__global__ void k_ntt_table9(const Fr *__restrict__ in, Fr *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr k;
#pragma unroll
    for (int j = 0; j < 8; j++) k.l[j] = Fr9Params::SCALE32[j];
    out[i] = in[i] * k;
}

void NttDomain::init(int logn_, hipStream_t s) {
    logn = logn_;
    n = (size_t)1 << logn;
    consts.alloc(8);
    tw_fwd.alloc(n / 2 + 1);
    tw_inv.alloc(n / 2 + 1);
    coset_pre.alloc(n);
    coset_post.alloc(n);
    hipLaunchKernelGGL(k_ntt_consts, dim3(1), dim3(64), 0, s, consts.p, logn);
    hipLaunchKernelGGL(k_ntt_tables, dim3(ceil_div(n, 256)), dim3(256), 0, s, consts.p, logn, tw_fwd.p, tw_inv.p, coset_pre.p, coset_post.p);
    // Opt-in experiment (ZKPROVE_NTT_F9=1): bit-exact, but measured SLOWER than the weak 8 x 32 butterflies (27.5 vs 24.3 ms per 1024-batch
    // of 2^16, 58.7 vs 52.0 ms at 2^17): a butterfly is one product and two additions, and what the 9-limb product saves (162
    // multiply-adds against 256 multiply-add / carry instructions) the additions give back -- a sum has to be brought back under 4M
    // and re-normalised (44 instructions against 32), the table entry re-limbed (17), and the tile holds 2 wavefronts per SIMD, where
    // the single dependent accumulator chain of a lone product is not hidden.  The bucket kernels (msm.hip), which are products
    // almost only, gain 25 % from the same field.  (Also tried: the weak 8 x 32 additions around the 9-limb product, re-limbing both
    // operands and packing the result -- 23.5 vs 23.3 ms; the same with two butterflies per call so that the two products run in lockstep --
    // 24.5 vs 22.3 ms on the pipelined passes, the callee's registers push the kernel past 256 VGPRs; both dropped.  ZKPROVE_NTT_DIAG=2 runs the passes without any arithmetic: 13.8 ms of
    // the 23.3 are loads, LDS exchange and stores at 3.7 TB/s, hidden under the butterflies.)
    // (Built only with -DZK_EXPERIMENTS -- `make experiments`, libzkprove_exp.so -- where the parity tests keep it bit-exact.)
#ifdef ZK_EXPERIMENTS
    static const bool want9 = [] { const char *e = getenv("ZKPROVE_NTT_F9"); return e && e[0] == '1'; }();
    f9 = want9;
#else
    f9 = false;
#endif
    if (f9) {
        tw_fwd9.alloc(n / 2 + 1); tw_inv9.alloc(n / 2 + 1); coset_pre9.alloc(n); coset_post9.alloc(n);
        const size_t h = n / 2 ? n / 2 : 1;
        hipLaunchKernelGGL(k_ntt_table9, dim3(ceil_div(h, 256)), dim3(256), 0, s, tw_fwd.p, tw_fwd9.p, h);
        hipLaunchKernelGGL(k_ntt_table9, dim3(ceil_div(h, 256)), dim3(256), 0, s, tw_inv.p, tw_inv9.p, h);
        hipLaunchKernelGGL(k_ntt_table9, dim3(ceil_div(n, 256)), dim3(256), 0, s, coset_pre.p, coset_pre9.p, n);
        hipLaunchKernelGGL(k_ntt_table9, dim3(ceil_div(n, 256)), dim3(256), 0, s, coset_post.p, coset_post9.p, n);
    }
    ZK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Element policies of the butterfly kernels.
//   NttWords : 8 x 32 weak form (field.cuh), carry-chain add / sub, out-of-line weak multiplier.
//   Ntt9     : 9 x 29-bit signed limbs (field9.cuh).  The DATA keep their 2^256-Montgomery form -- only the twiddle / coset
//              tables are stored in the 2^261 form (x * 2^256 * w * 2^261 / 2^261 = xw * 2^256) --, so HBM holds ordinary words.
//              Invariant between butterflies: strict limbs, value in [0, 4M + eps).  A product returns [0, 1.06M) because its
//              left operand is made non-negative first (u - v + 4M); a sum is followed by a conditional - 4M decided on the top
//              limbs (exact up to one unit of 2^232, i.e. eps = 2^-21 M) and one carry propagation.  Everything < 5.29M = 2^256
//              fits the 8 words of an intermediate pass as it is; the last pass of a transform stores the canonical value.
struct NttWords {
    using E = Fr;
    static constexpr bool F9 = false;
    static __device__ __forceinline__ E load(const Fr &w) { return w; }
    static __device__ __forceinline__ Fr store(const E &x, bool) { return x; }
    static __device__ __forceinline__ E zero() { return Fr::zero(); }
    static __device__ __forceinline__ E mul(const E &x, const Fr &t) { return Fr::mul_wc(x, t); }
    static __device__ __forceinline__ void dif(E &u, E &v, const Fr *t) {
        const E s = Fr::add_w(u, v), d = Fr::sub_w(u, v);
        u = s;
        v = t ? Fr::mul_wc(d, *t) : d;
    }
    static __device__ __forceinline__ void dit(E &u, E &v, const Fr *t) {
        const E w = t ? Fr::mul_wc(v, *t) : v;
        const E a = Fr::add_w(u, w);
        v = Fr::sub_w(u, w);
        u = a;
    }
};
#ifdef ZK_EXPERIMENTS
// timing diagnostics (ZKPROVE_NTT_DIAG, results are WRONG): 1 = butterflies without the product, 2 = without anything (loads,
// LDS exchange and stores only) -- what the passes cost apart from the arithmetic
template <int LEVEL>
struct NttDiag {
    using E = Fr;
    static constexpr bool F9 = false;
    static __device__ __forceinline__ E load(const Fr &w) { return w; }
    static __device__ __forceinline__ Fr store(const E &x, bool) { return x; }
    static __device__ __forceinline__ E zero() { return Fr::zero(); }
    static __device__ __forceinline__ E mul(const E &x, const Fr &t) { E r = x; r.l[0] ^= t.l[0]; return r; }
    static __device__ __forceinline__ void dif(E &u, E &v, const Fr *t) {
        if (LEVEL == 1) { const E s = Fr::add_w(u, v), d = Fr::sub_w(u, v); u = s; v = d; }
        if (t) v.l[0] ^= t->l[0];
        if (LEVEL == 2) { const uint32_t x = u.l[1]; u.l[1] = v.l[1]; v.l[1] = x; }
    }
    static __device__ __forceinline__ void dit(E &u, E &v, const Fr *t) { dif(u, v, t); }
};
struct Ntt9 {
    using E = Fr9;
    static constexpr bool F9 = true;
    static __device__ __forceinline__ E load(const Fr &w) { return Fr9::from_words(w); }
    static __device__ __forceinline__ E zero() { return Fr9::zero(); }
    // x in [0, 4M + eps), strict -> words; the last pass of a transform writes the canonical value
    static __device__ __forceinline__ Fr store(const E &x, bool last) {
        if (last) return Fr9::to_words(x);
        Fr r;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int lo = (32 * j) / 29, sh = 32 * j - 29 * lo;
            uint32_t v = (uint32_t)x.l[lo] >> sh;
            if (lo + 1 < 9) v |= (uint32_t)x.l[lo + 1] << (29 - sh);
            r.l[j] = v;
        }
        return r;
    }
    // s - 4M when the top limbs say s > 4M, then carry propagation.  s = lazy sum of strict values; the test reads the
    // un-normalised top limb (the lower limbs carry at most 2 units into it), so the result is < 4M + 3 * 2^232 when nothing is
    // subtracted and < (bound of s) - 4M otherwise: eps at most doubles per stage from 2^-19.6 M, < 0.06 M after the 16 stages of
    // a transform (a product resets it)
    static __device__ __forceinline__ E fold(E s) {
        const int32_t mask = (Fr9Params::M4[8] - s.l[8]) >> 31;          // all ones <=> top limb > that of 4M
#pragma unroll
        for (int i = 0; i < 9; i++) s.l[i] -= Fr9Params::M4[i] & mask;
        return Fr9::norm(s);
    }
    // x * t, t = a table entry (words of the 2^261 form).  ONE out-of-line copy: a radix-8 group has 12 products, two groups per
    // kernel -- expanded in place they make 150 KB of code per kernel, far beyond the instruction cache
    // (operands and result as vectors: a struct of more than 32 bytes would travel through scratch memory, vectors use VGPRs)
    typedef int32_t v9i __attribute__((ext_vector_type(9)));
    typedef uint32_t v8u __attribute__((ext_vector_type(8)));
    static __device__ __noinline__ v9i mul_v(v9i xv, v8u tv) {
        E x;
        Fr t;
#pragma unroll
        for (int i = 0; i < 9; i++) x.l[i] = xv[i];
#pragma unroll
        for (int i = 0; i < 8; i++) t.l[i] = tv[i];
        const E r = Fr9::mul(x, Fr9::from_words(t));
        v9i rv;
#pragma unroll
        for (int i = 0; i < 9; i++) rv[i] = r.l[i];
        return rv;
    }
    static __device__ __forceinline__ E mul(const E &x, const Fr &t) {
        v9i xv;
        v8u tv;
#pragma unroll
        for (int i = 0; i < 9; i++) xv[i] = x.l[i];
#pragma unroll
        for (int i = 0; i < 8; i++) tv[i] = t.l[i];
        const v9i rv = mul_v(xv, tv);
        E r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = rv[i];
        return r;
    }
    // (u, v) -> (u + v, (u - v) * t); t is never null here (the twiddle 1 is multiplied like any other: the product is what
    // brings the difference back into range)
    static __device__ __forceinline__ void dif(E &u, E &v, const Fr *t) {
        E d;
#pragma unroll
        for (int i = 0; i < 9; i++) d.l[i] = u.l[i] - v.l[i] + Fr9Params::M5[i];     // u - v + 5M in (M - eps, 9M + eps): positive; limbs within (-2^29, 2^30)
        u = fold(Fr9::add(u, v));
        v = mul(d, *t);                                                             // [0, 1.06 M)
    }
    // (u, v) -> (u + v t, u - v t)
    static __device__ __forceinline__ void dit(E &u, E &v, const Fr *t) {
        const E w = mul(v, *t);                                                     // [0, 1.03 M)
        E d;
#pragma unroll
        for (int i = 0; i < 9; i++) d.l[i] = u.l[i] - w.l[i] + Fr9Params::M2[i];     // u - w + 2M in (0.9 M, 6M + eps)
        u = fold(Fr9::add(u, w));
        v = fold(d);
    }
};

// Timing diagnostic (ZKPROVE_NTT_DIAG=9, results are WRONG once a lazy sum leaves the multiplier's range): the butterflies VERDICT r4
// task 6 asks about -- 9 signed limbs kept LAZY across the stages of a pass (sum and difference limb by limb, no carry, no fold; the
// product inline against a twiddle re-limbed from its 2^261-form words), one range reduction + carry propagation + conversion per
// element and HBM pass -- in the production pass kernels.  What the transforms would cost on them at best (the bound-correct version
// adds a reduction of two of eight elements at the LDS exchange): scripts/debug/ntt_bench.py, profiles/r05*_ntt_lazy9.txt.
struct Ntt9Lazy {
    using E = Fr9;
    static constexpr bool F9 = true;
    static __device__ __forceinline__ E load(const Fr &w) { return Fr9::from_words(w); }
    static __device__ __forceinline__ E zero() { return Fr9::zero(); }
    static __device__ __forceinline__ Fr store(const E &x, bool) {
        E v = x;
        const int32_t q = v.l[8] >> 21;            // value / 2^253, estimated on the top limb: v - q M lands within a few M of zero
        int64_t c = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            c += (int64_t)v.l[j] - (int64_t)q * Fr9Params::M[j];
            if (j < 8) { v.l[j] = (int32_t)(c & 0x1fffffff); c >>= 29; } else v.l[j] = (int32_t)c;
        }
        return Fr9::to_words(v);
    }
    static __device__ __forceinline__ E mul(const E &x, const Fr &t) { return Fr9::mul(x, Fr9::from_words(t)); }
    static __device__ __forceinline__ void dif(E &u, E &v, const Fr *t) {
        const E s = Fr9::add(u, v), d = Fr9::sub(u, v);
        u = s;
        v = Fr9::mul(d, Fr9::from_words(*t));
    }
    static __device__ __forceinline__ void dit(E &u, E &v, const Fr *t) {
        const E w = Fr9::mul(v, Fr9::from_words(*t));
        const E s = Fr9::add(u, w), d = Fr9::sub(u, w);
        u = s;
        v = d;
    }
};

#endif  // ZK_EXPERIMENTS

// ---------------------------------------------------------------------------------------------
// DIF pass: K stages starting at block length `len` (natural -> towards bit-reversed).
// thread (g, b): g in [0, n >> K), b in [0, B)
template <class P, int K>
__global__ __launch_bounds__(256) void k_ntt_dif(Fr *__restrict__ data, size_t B, int logn, int loglen,
                                                 const Fr *__restrict__ tw, const Fr *__restrict__ postmul, int last) {
    using E = typename P::E;
    constexpr int R = 1 << K;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = t % B;
    const size_t g = t / B;
    const size_t groups = ((size_t)1 << logn) >> K;
    if (g >= groups) return;
    const int logS = loglen - K;
    const size_t S = (size_t)1 << logS;
    const size_t blk = g >> logS, p = g & (S - 1);
    const size_t base = (blk << loglen) + p;
    E x[R];
#pragma unroll
    for (int m = 0; m < R; m++) x[m] = P::load(data[(base + ((size_t)m << logS)) * B + b]);
#pragma unroll
    for (int st = 0; st < K; st++) {
        const int half = 1 << (K - 1 - st);          // pair distance in m units
        const int logL = loglen - st;                // this stage's block length
        const int twshift = logn - logL;
#pragma unroll
        for (int m = 0; m < R; m++) {
            if ((m / half) & 1) continue;
            const size_t j = p + ((size_t)(m % half) << logS);
            P::dif(x[m], x[m + half], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
        }
    }
#pragma unroll
    for (int m = 0; m < R; m++) {
        const size_t idx = base + ((size_t)m << logS);
        E v = x[m];
        if (postmul) v = P::mul(v, postmul[idx]);
        data[idx * B + b] = P::store(v, last != 0);
    }
}

// DIT pass: K stages starting at half-distance S = 2^logS (bit-reversed -> natural).
template <class P, int K>
__global__ __launch_bounds__(256) void k_ntt_dit(Fr *__restrict__ data, size_t B, int logn, int logS,
                                                 const Fr *__restrict__ tw, const Fr *__restrict__ premul, int last) {
    using E = typename P::E;
    constexpr int R = 1 << K;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = t % B;
    const size_t g = t / B;
    const size_t groups = ((size_t)1 << logn) >> K;
    if (g >= groups) return;
    const size_t S = (size_t)1 << logS;
    const size_t blk = g >> logS, p = g & (S - 1);
    const size_t base = (blk << (logS + K)) + p;
    E x[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        const size_t idx = base + ((size_t)m << logS);
        E v = P::load(data[idx * B + b]);
        if (premul) v = P::mul(v, premul[idx]);
        x[m] = v;
    }
#pragma unroll
    for (int st = 0; st < K; st++) {
        const int h = 1 << st;                       // pair distance in m units
        const int logL = logS + st + 1;              // block length of this stage
        const int twshift = logn - logL;
#pragma unroll
        for (int m = 0; m < R; m++) {
            if ((m >> st) & 1) continue;
            const size_t j = p + ((size_t)(m & (h - 1)) << logS);
            P::dit(x[m], x[m + h], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
        }
    }
#pragma unroll
    for (int m = 0; m < R; m++) data[(base + ((size_t)m << logS)) * B + b] = P::store(x[m], last != 0);
}

// ---------------------------------------------------------------------------------------------
// Two-level passes through LDS: a workgroup of 256 threads owns a tile of 2^(K1+K2) elements x TI instances
// (TI * 2^(K1+K2) * 32 B = 64 KiB of LDS), does K1 butterfly stages in registers, exchanges through LDS and does
// K2 more -- K1+K2 = 5..6 stages per HBM round trip instead of 3 (N = 2^16: 3 passes instead of 6).
// Global accesses stay instance-contiguous (TI * 32 B segments).
// TOTAL = field elements per tile (x 32 B of LDS); the workgroup has one thread per 8 of them, i.e. TOTAL / 8 threads.
// Smaller tiles = more, smaller workgroups per CU in different phases (load / butterflies / store) at the same LDS footprint.
#ifndef ZK_NTT_TILE
#define ZK_NTT_TILE 2048
#endif
// The transforms are the critical chain of the window between two quotient walks (transforms -> patterns -> walk) while the wire MSMs, the
// solver and the tail beside them are not: the pass kernels raise the issue priority of their wavefronts (s_setprio 3), so that they win the
// arbitration against the other streams' wavefronts on their SIMD.  Same box, twice each (profiles/r05zh_*): ChaCha20 11 420 / 11 399 -> 11 527 /
// 11 504 proofs/s (the window 32.3 -> 31.3 ms), AES-256 unchanged.  -DZK_NTT_PRIO=0: A/B build without it.
#ifndef ZK_NTT_PRIO
#define ZK_NTT_PRIO 3
#endif
// (bit 1 of the kernels' `last` argument: the domain wants it -- NttDomain::hi_prio, set for the programs whose copy stream carries the
// critical chain, i.e. the short ones; for the AES programs the main stream's wire MSMs are the longer chain and the raise costs 0.2 %)
#if ZK_NTT_PRIO > 0
#define ZK_NTT_SETPRIO() do { if (last & 2) __builtin_amdgcn_s_setprio(ZK_NTT_PRIO); } while (0)
#else
#define ZK_NTT_SETPRIO() ((void)0)
#endif
static thread_local int t_ntt_prio_bit = 0;      // set by NttDomain::dif / dit around their launches
#ifndef ZK_NTT_TPW
#define ZK_NTT_TPW 4                                // tiles per workgroup of the pipelined kernels
#endif
template <class P, int K1, int K2>
struct LdsTile {
    static constexpr int K = K1 + K2;
    static constexpr int E = 1 << K;               // elements per tile
    static constexpr int TOTAL = ZK_NTT_TILE;
    static constexpr int TI = TOTAL / E;           // instances per tile
    static constexpr int THREADS = TOTAL / 8;
    static constexpr int LDS_BYTES = TOTAL * (int)sizeof(typename P::E);      // 64 KiB of words / 72 KiB of 9-limb elements
    static_assert(E * TI == TOTAL && TI >= 1, "tile shape");
};

// DIF: stages loglen .. loglen-K+1 (block lengths len, len/2, ...).  Element e of the tile sits at
// pos = blk*len + p + e*S with S = len >> K.
template <class P, int K1, int K2>
__global__ __launch_bounds__(ZK_NTT_TILE / 8) void k_ntt_dif_lds(Fr *__restrict__ data, size_t B, int logn, int loglen, const Fr *__restrict__ tw,
                                                     const Fr *__restrict__ postmul, int last) {
    using T = LdsTile<P, K1, K2>;
    using El = typename P::E;
    constexpr int K = T::K, E = T::E, TI = T::TI, R1 = 1 << K1, R2 = 1 << K2, NT = T::THREADS;
    ZK_NTT_SETPRIO();
    extern __shared__ uint4 lds_raw[];
    El *lds = (El *)lds_raw;                       // [E][TI]
    const int tid = threadIdx.x;
    const int bl = tid % TI;                       // instance within the tile
    // the instance tiles of one element tile are neighbours in the grid: workgroups resident at the same time then read
    // adjacent 1 KiB segments of the same 2^K rows (the rows of a tile lie n / 2^K elements = tens of MiB apart in the first
    // pass; with the element tile as the fast grid axis every resident workgroup touched 2^K pages of its own)
    const size_t b = (size_t)blockIdx.x * TI + bl;
    const int logS = loglen - K;
    const size_t S = (size_t)1 << logS;
    const size_t g = blockIdx.y;                   // tile index in [0, n >> K)
    const size_t blk = g >> logS, p = g & (S - 1);
    const size_t base = (blk << loglen) + p;
    const bool active = b < B;
    // ---- stage group A: thread holds e = q + R2' * m (R2' = E / R1), distances E/2 .. R2'
    {
        constexpr int STR = E / R1;                // element stride inside the thread's set
        for (int q = tid / TI; q < STR; q += NT / TI) {
            El x[R1];
#pragma unroll
            for (int m = 0; m < R1; m++) {
                const size_t e = (size_t)q + (size_t)STR * m;
                x[m] = active ? P::load(data[(base + (e << logS)) * B + b]) : P::zero();
            }
#pragma unroll
            for (int st = 0; st < K1; st++) {
                const int half = 1 << (K1 - 1 - st);
                const int twshift = logn - (loglen - st);
#pragma unroll
                for (int m = 0; m < R1; m++) {
                    if ((m / half) & 1) continue;
                    const size_t j = p + (((size_t)q + (size_t)STR * (m % half)) << logS);
                    P::dif(x[m], x[m + half], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
                }
            }
#pragma unroll
            for (int m = 0; m < R1; m++) lds[((size_t)q + (size_t)STR * m) * TI + bl] = x[m];
        }
    }
    __syncthreads();
    // ---- stage group B: thread holds e = q2 * R2 + m, distances R2/2 .. 1
    {
        for (int q2 = tid / TI; q2 < E / R2; q2 += NT / TI) {
            El x[R2];
#pragma unroll
            for (int m = 0; m < R2; m++) x[m] = lds[((size_t)q2 * R2 + m) * TI + bl];
#pragma unroll
            for (int st = 0; st < K2; st++) {
                const int half = 1 << (K2 - 1 - st);
                const int twshift = logn - (loglen - K1 - st);
#pragma unroll
                for (int m = 0; m < R2; m++) {
                    if ((m / half) & 1) continue;
                    const size_t j = p + ((size_t)(m % half) << logS);
                    P::dif(x[m], x[m + half], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
                }
            }
            if (active) {
#pragma unroll
                for (int m = 0; m < R2; m++) {
                    const size_t idx = base + (((size_t)q2 * R2 + m) << logS);
                    El v = x[m];
                    if (postmul) v = P::mul(v, postmul[idx]);
                    data[idx * B + b] = P::store(v, (last & 1) != 0);
                }
            }
        }
    }
}

// DIT: stages with half-distances S, 2S, ... (S = 2^logS), tile element e at pos = blk*(S<<K) + p + e*S.
template <class P, int K1, int K2>
__global__ __launch_bounds__(ZK_NTT_TILE / 8) void k_ntt_dit_lds(Fr *__restrict__ data, size_t B, int logn, int logS, const Fr *__restrict__ tw,
                                                     const Fr *__restrict__ premul, int last) {
    using T = LdsTile<P, K1, K2>;
    using El = typename P::E;
    constexpr int K = T::K, E = T::E, TI = T::TI, R1 = 1 << K1, R2 = 1 << K2, NT = T::THREADS;
    ZK_NTT_SETPRIO();
    extern __shared__ uint4 lds_raw[];
    El *lds = (El *)lds_raw;
    const int tid = threadIdx.x;
    const int bl = tid % TI;
    const size_t b = (size_t)blockIdx.x * TI + bl;
    const size_t S = (size_t)1 << logS;
    const size_t g = blockIdx.y;
    const size_t blk = g >> logS, p = g & (S - 1);
    const size_t base = (blk << (logS + K)) + p;
    const bool active = b < B;
    // ---- group A: consecutive elements e = q * R1 + m, distances 1 .. R1/2
    {
        for (int q = tid / TI; q < E / R1; q += NT / TI) {
            El x[R1];
#pragma unroll
            for (int m = 0; m < R1; m++) {
                const size_t idx = base + (((size_t)q * R1 + m) << logS);
                El v = active ? P::load(data[idx * B + b]) : P::zero();
                if (premul) v = P::mul(v, premul[idx]);
                x[m] = v;
            }
#pragma unroll
            for (int st = 0; st < K1; st++) {
                const int h = 1 << st;
                const int twshift = logn - (logS + st + 1);
#pragma unroll
                for (int m = 0; m < R1; m++) {
                    if ((m >> st) & 1) continue;
                    const size_t j = p + ((size_t)(m & (h - 1)) << logS);
                    P::dit(x[m], x[m + h], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
                }
            }
#pragma unroll
            for (int m = 0; m < R1; m++) lds[((size_t)q * R1 + m) * TI + bl] = x[m];
        }
    }
    __syncthreads();
    // ---- group B: strided elements e = q2 + R1 * m, distances R1 .. E/2
    {
        for (int q2 = tid / TI; q2 < R1; q2 += NT / TI) {
            El x[R2];
#pragma unroll
            for (int m = 0; m < R2; m++) x[m] = lds[((size_t)q2 + (size_t)R1 * m) * TI + bl];
#pragma unroll
            for (int st = 0; st < K2; st++) {
                const int h = 1 << st;                          // in units of R1 elements
                const int twshift = logn - (logS + K1 + st + 1);
#pragma unroll
                for (int m = 0; m < R2; m++) {
                    if ((m >> st) & 1) continue;
                    const size_t j = p + (((size_t)q2 + (size_t)R1 * (m & (h - 1))) << logS);
                    P::dit(x[m], x[m + h], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
                }
            }
            if (active) {
#pragma unroll
                for (int m = 0; m < R2; m++) data[(base + (((size_t)q2 + (size_t)R1 * m) << logS)) * B + b] = P::store(x[m], (last & 1) != 0);
            }
        }
    }
}

// Software-pipelined forms of the two kernels above: a workgroup walks TPW consecutive tiles and requests the NEXT tile's
// elements from HBM before it runs stage group B of the current one, so the loads fly under 12 butterflies instead of in front of
// them (two resident workgroups per CU are not enough to hide a tile load: the SQ counters show the VALU 69 % busy in the plain
// kernels, and ZKPROVE_NTT_DIAG=2 that the passes cost 13.8 of 23.3 ms without any arithmetic).  Requires one group-A element set
// per thread (E / R1 == THREADS / TI: the (3,3) and (3,2) shapes).
template <class P, int K1, int K2, int TPW>
__global__ __launch_bounds__(ZK_NTT_TILE / 8) void k_ntt_dif_lds_pipe(Fr *__restrict__ data, size_t B, int logn, int loglen, const Fr *__restrict__ tw,
                                                          const Fr *__restrict__ postmul, int last) {
    using T = LdsTile<P, K1, K2>;
    using El = typename P::E;
    constexpr int K = T::K, E = T::E, TI = T::TI, R1 = 1 << K1, R2 = 1 << K2, NT = T::THREADS, STR = E / R1;
    static_assert(STR == NT / TI, "one group-A element set per thread");
    ZK_NTT_SETPRIO();
    extern __shared__ uint4 lds_raw[];
    El *lds = (El *)lds_raw;                       // [E][TI]
    const int tid = threadIdx.x;
    const int bl = tid % TI, q = tid / TI;
    const size_t b = (size_t)blockIdx.x * TI + bl;
    const int logS = loglen - K;
    const size_t S = (size_t)1 << logS;
    const bool active = b < B;
    const size_t g0 = (size_t)blockIdx.y * TPW;
    auto tile_base = [&](size_t g) { return ((g >> logS) << loglen) + (g & (S - 1)); };
    Fr raw[R1];                                    // the tile in flight, as loaded
    {
        const size_t base = tile_base(g0);
#pragma unroll
        for (int m = 0; m < R1; m++) raw[m] = active ? data[(base + (((size_t)q + (size_t)STR * m) << logS)) * B + b] : Fr::zero();
    }
#pragma unroll 1
    for (int it = 0; it < TPW; it++) {
        const size_t g = g0 + it;
        const size_t p = g & (S - 1), base = tile_base(g);
        El x[R1];
#pragma unroll
        for (int m = 0; m < R1; m++) x[m] = P::load(raw[m]);
#pragma unroll
        for (int st = 0; st < K1; st++) {
            const int half = 1 << (K1 - 1 - st);
            const int twshift = logn - (loglen - st);
#pragma unroll
            for (int m = 0; m < R1; m++) {
                if ((m / half) & 1) continue;
                const size_t j = p + (((size_t)q + (size_t)STR * (m % half)) << logS);
                P::dif(x[m], x[m + half], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
            }
        }
#pragma unroll
        for (int m = 0; m < R1; m++) lds[((size_t)q + (size_t)STR * m) * TI + bl] = x[m];
        __syncthreads();
        if (it + 1 < TPW) {                        // next tile: requested now, consumed after group B
            const size_t nbase = tile_base(g + 1);
#pragma unroll
            for (int m = 0; m < R1; m++) raw[m] = active ? data[(nbase + (((size_t)q + (size_t)STR * m) << logS)) * B + b] : Fr::zero();
        }
        for (int q2 = q; q2 < E / R2; q2 += NT / TI) {
            El y[R2];
#pragma unroll
            for (int m = 0; m < R2; m++) y[m] = lds[((size_t)q2 * R2 + m) * TI + bl];
#pragma unroll
            for (int st = 0; st < K2; st++) {
                const int half = 1 << (K2 - 1 - st);
                const int twshift = logn - (loglen - K1 - st);
#pragma unroll
                for (int m = 0; m < R2; m++) {
                    if ((m / half) & 1) continue;
                    const size_t j = p + ((size_t)(m % half) << logS);
                    P::dif(y[m], y[m + half], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
                }
            }
            if (active) {
#pragma unroll
                for (int m = 0; m < R2; m++) {
                    const size_t idx = base + (((size_t)q2 * R2 + m) << logS);
                    El v = y[m];
                    if (postmul) v = P::mul(v, postmul[idx]);
                    data[idx * B + b] = P::store(v, (last & 1) != 0);
                }
            }
        }
        __syncthreads();                           // the tile buffer is rewritten by the next iteration's group A
    }
}
template <class P, int K1, int K2, int TPW>
__global__ __launch_bounds__(ZK_NTT_TILE / 8) void k_ntt_dit_lds_pipe(Fr *__restrict__ data, size_t B, int logn, int logS, const Fr *__restrict__ tw,
                                                          const Fr *__restrict__ premul, int last) {
    using T = LdsTile<P, K1, K2>;
    using El = typename P::E;
    constexpr int K = T::K, E = T::E, TI = T::TI, R1 = 1 << K1, R2 = 1 << K2, NT = T::THREADS;
    static_assert(E / R1 == NT / TI, "one group-A element set per thread");
    ZK_NTT_SETPRIO();
    extern __shared__ uint4 lds_raw[];
    El *lds = (El *)lds_raw;
    const int tid = threadIdx.x;
    const int bl = tid % TI, q = tid / TI;
    const size_t b = (size_t)blockIdx.x * TI + bl;
    const size_t S = (size_t)1 << logS;
    const bool active = b < B;
    const size_t g0 = (size_t)blockIdx.y * TPW;
    auto tile_base = [&](size_t g) { return ((g >> logS) << (logS + K)) + (g & (S - 1)); };
    Fr raw[R1];
    {
        const size_t base = tile_base(g0);
#pragma unroll
        for (int m = 0; m < R1; m++) raw[m] = active ? data[(base + (((size_t)q * R1 + m) << logS)) * B + b] : Fr::zero();
    }
#pragma unroll 1
    for (int it = 0; it < TPW; it++) {
        const size_t g = g0 + it;
        const size_t p = g & (S - 1), base = tile_base(g);
        El x[R1];
#pragma unroll
        for (int m = 0; m < R1; m++) {
            El v = P::load(raw[m]);
            if (premul) v = P::mul(v, premul[base + (((size_t)q * R1 + m) << logS)]);
            x[m] = v;
        }
#pragma unroll
        for (int st = 0; st < K1; st++) {
            const int h = 1 << st;
            const int twshift = logn - (logS + st + 1);
#pragma unroll
            for (int m = 0; m < R1; m++) {
                if ((m >> st) & 1) continue;
                const size_t j = p + ((size_t)(m & (h - 1)) << logS);
                P::dit(x[m], x[m + h], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
            }
        }
#pragma unroll
        for (int m = 0; m < R1; m++) lds[((size_t)q * R1 + m) * TI + bl] = x[m];
        __syncthreads();
        if (it + 1 < TPW) {
            const size_t nbase = tile_base(g + 1);
#pragma unroll
            for (int m = 0; m < R1; m++) raw[m] = active ? data[(nbase + (((size_t)q * R1 + m) << logS)) * B + b] : Fr::zero();
        }
        for (int q2 = q; q2 < R1; q2 += NT / TI) {
            El y[R2];
#pragma unroll
            for (int m = 0; m < R2; m++) y[m] = lds[((size_t)q2 + (size_t)R1 * m) * TI + bl];
#pragma unroll
            for (int st = 0; st < K2; st++) {
                const int h = 1 << st;                          // in units of R1 elements
                const int twshift = logn - (logS + K1 + st + 1);
#pragma unroll
                for (int m = 0; m < R2; m++) {
                    if ((m >> st) & 1) continue;
                    const size_t j = p + (((size_t)q2 + (size_t)R1 * (m & (h - 1))) << logS);
                    P::dit(y[m], y[m + h], (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
                }
            }
            if (active) {
#pragma unroll
                for (int m = 0; m < R2; m++) data[(base + (((size_t)q2 + (size_t)R1 * m) << logS)) * B + b] = P::store(y[m], (last & 1) != 0);
            }
        }
        __syncthreads();
    }
}

template <class P, int K1, int K2>
static void launch_dif_lds(Fr *data, size_t B, int logn, int loglen, const Fr *tw, const Fr *postmul, bool last, hipStream_t s) {
    using T = LdsTile<P, K1, K2>;
    static PerDeviceOnce attr;
    attr.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_dif_lds<P, K1, K2>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)); });
    dim3 grid(ceil_div(B, T::TI), (unsigned)(((size_t)1 << logn) >> T::K));
    if constexpr (T::E / (1 << K1) == T::THREADS / T::TI) {
        constexpr int TPW = ZK_NTT_TPW;
        static const bool pipe = [] { const char *e = ZK_AB_ENV("ZKPROVE_NTT_PIPE"); return !(e && e[0] == '0'); }();
        if (pipe && grid.y % TPW == 0 && grid.y >= (unsigned)TPW) {
            static PerDeviceOnce attr2;
            attr2.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_dif_lds_pipe<P, K1, K2, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)); });
            grid.y /= TPW;
            hipLaunchKernelGGL((k_ntt_dif_lds_pipe<P, K1, K2, TPW>), grid, dim3(T::THREADS), T::LDS_BYTES, s, data, B, logn, loglen, tw, postmul, (int)last | t_ntt_prio_bit);
            return;
        }
    }
    hipLaunchKernelGGL((k_ntt_dif_lds<P, K1, K2>), grid, dim3(T::THREADS), T::LDS_BYTES, s, data, B, logn, loglen, tw, postmul, (int)last | t_ntt_prio_bit);
}
template <class P, int K1, int K2>
static void launch_dit_lds(Fr *data, size_t B, int logn, int logS, const Fr *tw, const Fr *premul, bool last, hipStream_t s) {
    using T = LdsTile<P, K1, K2>;
    static PerDeviceOnce attr;
    attr.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_dit_lds<P, K1, K2>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)); });
    dim3 grid(ceil_div(B, T::TI), (unsigned)(((size_t)1 << logn) >> T::K));
    if constexpr (T::E / (1 << K1) == T::THREADS / T::TI) {
        constexpr int TPW = ZK_NTT_TPW;
        static const bool pipe = [] { const char *e = ZK_AB_ENV("ZKPROVE_NTT_PIPE"); return !(e && e[0] == '0'); }();
        if (pipe && grid.y % TPW == 0 && grid.y >= (unsigned)TPW) {
            static PerDeviceOnce attr2;
            attr2.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_dit_lds_pipe<P, K1, K2, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)); });
            grid.y /= TPW;
            hipLaunchKernelGGL((k_ntt_dit_lds_pipe<P, K1, K2, TPW>), grid, dim3(T::THREADS), T::LDS_BYTES, s, data, B, logn, logS, tw, premul, (int)last | t_ntt_prio_bit);
            return;
        }
    }
    hipLaunchKernelGGL((k_ntt_dit_lds<P, K1, K2>), grid, dim3(T::THREADS), T::LDS_BYTES, s, data, B, logn, logS, tw, premul, (int)last | t_ntt_prio_bit);
}

template <class P, int K>
static void launch_dif(Fr *data, size_t B, int logn, int loglen, const Fr *tw, const Fr *postmul, bool last, hipStream_t s) {
    size_t threads = (((size_t)1 << logn) >> K) * B;
    hipLaunchKernelGGL((k_ntt_dif<P, K>), dim3(ceil_div(threads, 256)), dim3(256), 0, s, data, B, logn, loglen, tw, postmul, (int)last);
}
template <class P, int K>
static void launch_dit(Fr *data, size_t B, int logn, int logS, const Fr *tw, const Fr *premul, bool last, hipStream_t s) {
    size_t threads = (((size_t)1 << logn) >> K) * B;
    hipLaunchKernelGGL((k_ntt_dit<P, K>), dim3(ceil_div(threads, 256)), dim3(256), 0, s, data, B, logn, logS, tw, premul, (int)last);
}

// stage schedule: as many 6-stage LDS passes as possible, then one 5- or 4-stage LDS pass, then register passes
// Experiment (-DZK_EXPERIMENTS, ZKPROVE_NTT_EIGHT=1; VERDICT r2 task 5): 8-stage (4,4) LDS passes -- 256 elements x 8 instances per 64 KiB
// tile, 16 elements per thread -- so that N = 2^16 takes TWO round trips through HBM per transform instead of three.  Bit-exact and
// SLOWER: 38.0 vs 22.4 ms for the four transforms of a 1024-batch (the tile's rows shrink to 256-byte segments, 225-245 VGPRs leave two
// wavefronts per SIMD, no software pipelining), 9 414 vs 10 918 proofs/s on one box.  The step is bound by VALU work, of which the
// transforms' share (~16 ms) does not depend on the pass structure, so the three-pass (3,3) tiling stays.
static int next_pass(int rem) {
#ifdef ZK_EXPERIMENTS
    static const bool eight = [] { const char *e = getenv("ZKPROVE_NTT_EIGHT"); return e && e[0] == '1'; }();
    if (eight && rem >= 8 && rem % 8 == 0) return 8;
#endif
    if (rem >= 6) return 6;
    return rem;     // 5, 4 -> LDS (3,2)/(2,2); 3, 2, 1 -> register kernels
}
template <class P>
static void dif_passes(int logn, Fr *data, size_t B, const Fr *tw, const Fr *postmul, hipStream_t s) {
    int loglen = logn;
    while (loglen > 0) {
        const int k = next_pass(loglen);
        const bool last = loglen - k == 0;
        const Fr *pm = last ? postmul : nullptr;
        switch (k) {
#ifdef ZK_EXPERIMENTS
            case 8: launch_dif_lds<P, 4, 4>(data, B, logn, loglen, tw, pm, last, s); break;
#endif
            case 6: launch_dif_lds<P, 3, 3>(data, B, logn, loglen, tw, pm, last, s); break;
            case 5: launch_dif_lds<P, 3, 2>(data, B, logn, loglen, tw, pm, last, s); break;
            case 4: launch_dif_lds<P, 2, 2>(data, B, logn, loglen, tw, pm, last, s); break;
            case 3: launch_dif<P, 3>(data, B, logn, loglen, tw, pm, last, s); break;
            case 2: launch_dif<P, 2>(data, B, logn, loglen, tw, pm, last, s); break;
            default: launch_dif<P, 1>(data, B, logn, loglen, tw, pm, last, s); break;
        }
        loglen -= k;
    }
}
template <class P>
static void dit_passes(int logn, Fr *data, size_t B, const Fr *tw, const Fr *premul, hipStream_t s) {
    int logS = 0;
    while (logS < logn) {
        const int k = next_pass(logn - logS);
        const Fr *pm = (logS == 0) ? premul : nullptr;
        const bool last = logS + k == logn;
        switch (k) {
#ifdef ZK_EXPERIMENTS
            case 8: launch_dit_lds<P, 4, 4>(data, B, logn, logS, tw, pm, last, s); break;
#endif
            case 6: launch_dit_lds<P, 3, 3>(data, B, logn, logS, tw, pm, last, s); break;
            case 5: launch_dit_lds<P, 3, 2>(data, B, logn, logS, tw, pm, last, s); break;
            case 4: launch_dit_lds<P, 2, 2>(data, B, logn, logS, tw, pm, last, s); break;
            case 3: launch_dit<P, 3>(data, B, logn, logS, tw, pm, last, s); break;
            case 2: launch_dit<P, 2>(data, B, logn, logS, tw, pm, last, s); break;
            default: launch_dit<P, 1>(data, B, logn, logS, tw, pm, last, s); break;
        }
        logS += k;
    }
}
#ifdef ZK_EXPERIMENTS
// ---------------------------------------------------------------------------------------------
// EXPERIMENT (VERDICT r3 task 6; ZKPROVE_NTT_FUSE=1 in the experiments build): five HBM round trips per transform pair instead of six.
// Bit-exact and SLOWER: 24.0 vs 22.3 ms for the four transforms of a 1024-batch at N = 2^16, 51.3 vs 46.8 ms at 2^17 (one box,
// scripts/debug/ntt_prof.sh, profiles/r04d_ntt_fuse*.txt).  Per butterfly stage (a pair is 32 stages + the scaling = 2 more stages'
// worth of products): the (3,3) passes 0.36 ms, the (3,2) passes 0.32, this 12-stage pass 0.34 -- every pass runs at the rate of its
// products (2.3 G Fr products per batch at ~120 G/s), so a saved round trip through HBM saves nothing; the 5 + 5 split that makes
// room for it replaces the cheap (2,2) passes (0.2 ms per stage: half of their twiddles are 1 and cost no product).
// The prover only ever runs the PAIR inverse (DIF) + coset-forward (DIT) on a vector (to_coset).  The last six DIF stages (block
// lengths 64 .. 2) and the first six DIT stages (half-distances 1 .. 32) both live inside blocks of 64 CONSECUTIVE positions, so one
// tile visit does all twelve, with the scaling g^rev(i) / n between them: 5 HBM round trips per pair at N = 2^16 / 2^17
// (DIF 5+5 | 6+6 fused | DIT 5+5; DIF 6+5 | fused | DIT 6+5) instead of 6 (6+6+4 twice).  Inside the tile: DIF distances 32,16,8 with
// the thread holding e = q + 8m, LDS exchange, then DIF distances 4,2,1, the scaling and DIT distances 1,2,4 all on the SAME eight
// elements e = 8q + m (no exchange between the two transforms), LDS exchange, DIT distances 8,16,32 on e = q + 8m.  Same butterflies,
// same twiddle indices and the same weak-form arithmetic as the separate passes, hence bit-identical results (fr/fft difFFT /
// ditFFT of gnark-crypto).  Software-pipelined like the passes above: the next tile's loads fly under the middle section.
template <class P, int TPW>
__global__ __launch_bounds__(ZK_NTT_TILE / 8) void k_ntt_mid_lds_pipe(Fr *__restrict__ data, size_t B, int logn, const Fr *__restrict__ tw_inv,
                                                          const Fr *__restrict__ tw_fwd, const Fr *__restrict__ premul) {
    using T = LdsTile<P, 3, 3>;
    using El = typename P::E;
    constexpr int E = T::E, TI = T::TI, NT = T::THREADS, R = 8;
    static_assert(E == 64 && NT / TI == 8 && !P::F9, "tile shape of the fused middle pass");
    extern __shared__ uint4 lds_raw[];
    El *lds = (El *)lds_raw;                       // [E][TI]
    const int tid = threadIdx.x;
    const int bl = tid % TI, q = tid / TI;         // q in [0, 8)
    const size_t b = (size_t)blockIdx.x * TI + bl;
    const bool active = b < B;
    const size_t g0 = (size_t)blockIdx.y * TPW;
    Fr raw[R];
#pragma unroll
    for (int m = 0; m < R; m++) raw[m] = active ? data[((g0 << 6) + (size_t)(q + 8 * m)) * B + b] : Fr::zero();
#pragma unroll 1
    for (int it = 0; it < TPW; it++) {
        const size_t base = (g0 + it) << 6;
        El x[R];
#pragma unroll
        for (int m = 0; m < R; m++) x[m] = P::load(raw[m]);
        // inverse transform, block lengths 64, 32, 16: element e = q + 8 m
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int half = 4 >> st;
            const int twshift = logn - (6 - st);
#pragma unroll
            for (int m = 0; m < R; m++) {
                if ((m / half) & 1) continue;
                const size_t j = (size_t)q + 8 * (size_t)(m % half);
                P::dif(x[m], x[m + half], j == 0 ? nullptr : &tw_inv[j << twshift]);
            }
        }
#pragma unroll
        for (int m = 0; m < R; m++) lds[(q + 8 * m) * TI + bl] = x[m];
        __syncthreads();
        if (it + 1 < TPW) {
            const size_t nbase = (g0 + it + 1) << 6;
#pragma unroll
            for (int m = 0; m < R; m++) raw[m] = active ? data[(nbase + (size_t)(q + 8 * m)) * B + b] : Fr::zero();
        }
        // element e = 8 q + m: inverse block lengths 8, 4, 2; scaling; forward half-distances 1, 2, 4
        El y[R];
#pragma unroll
        for (int m = 0; m < R; m++) y[m] = lds[(q * 8 + m) * TI + bl];
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int half = 4 >> st;
            const int twshift = logn - (3 - st);
#pragma unroll
            for (int m = 0; m < R; m++) {
                if ((m / half) & 1) continue;
                const size_t j = (size_t)(m % half);
                P::dif(y[m], y[m + half], j == 0 ? nullptr : &tw_inv[j << twshift]);
            }
        }
#pragma unroll
        for (int m = 0; m < R; m++) y[m] = P::mul(y[m], premul[base + (size_t)(q * 8 + m)]);
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int h = 1 << st;
            const int twshift = logn - (st + 1);
#pragma unroll
            for (int m = 0; m < R; m++) {
                if ((m >> st) & 1) continue;
                const size_t j = (size_t)(m & (h - 1));
                P::dit(y[m], y[m + h], j == 0 ? nullptr : &tw_fwd[j << twshift]);
            }
        }
        // (each thread writes back exactly the rows it read: no barrier in front of the stores)
#pragma unroll
        for (int m = 0; m < R; m++) lds[(q * 8 + m) * TI + bl] = y[m];
        __syncthreads();
        // forward half-distances 8, 16, 32: element e = q + 8 m
        El z[R];
#pragma unroll
        for (int m = 0; m < R; m++) z[m] = lds[(q + 8 * m) * TI + bl];
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int h = 1 << st;
            const int twshift = logn - (3 + st + 1);
#pragma unroll
            for (int m = 0; m < R; m++) {
                if ((m >> st) & 1) continue;
                const size_t j = (size_t)q + 8 * (size_t)(m & (h - 1));
                P::dit(z[m], z[m + h], j == 0 ? nullptr : &tw_fwd[j << twshift]);
            }
        }
        if (active) {
#pragma unroll
            for (int m = 0; m < R; m++) data[(base + (size_t)(q + 8 * m)) * B + b] = P::store(z[m], false);
        }
        __syncthreads();                           // the tile buffer is rewritten by the next iteration
    }
}
// pass sizes for `rem` butterfly stages: as few passes as 6 stages each allow, balanced (10 -> 5 + 5, 11 -> 6 + 5)
static int balanced_passes(int rem, int (&sizes)[8]) {
    if (rem <= 0) return 0;
    const int k = (rem + 5) / 6;
    for (int i = 0; i < k; i++) sizes[i] = rem / k + (i < rem % k ? 1 : 0);
    return k;
}
template <class P>
static void launch_dif_pass(int k, Fr *data, size_t B, int logn, int loglen, const Fr *tw, const Fr *pm, bool last, hipStream_t s) {
    switch (k) {
        case 6: launch_dif_lds<P, 3, 3>(data, B, logn, loglen, tw, pm, last, s); break;
        case 5: launch_dif_lds<P, 3, 2>(data, B, logn, loglen, tw, pm, last, s); break;
        case 4: launch_dif_lds<P, 2, 2>(data, B, logn, loglen, tw, pm, last, s); break;
        case 3: launch_dif<P, 3>(data, B, logn, loglen, tw, pm, last, s); break;
        case 2: launch_dif<P, 2>(data, B, logn, loglen, tw, pm, last, s); break;
        default: launch_dif<P, 1>(data, B, logn, loglen, tw, pm, last, s); break;
    }
}
template <class P>
static void launch_dit_pass(int k, Fr *data, size_t B, int logn, int logS, const Fr *tw, const Fr *pm, bool last, hipStream_t s) {
    switch (k) {
        case 6: launch_dit_lds<P, 3, 3>(data, B, logn, logS, tw, pm, last, s); break;
        case 5: launch_dit_lds<P, 3, 2>(data, B, logn, logS, tw, pm, last, s); break;
        case 4: launch_dit_lds<P, 2, 2>(data, B, logn, logS, tw, pm, last, s); break;
        case 3: launch_dit<P, 3>(data, B, logn, logS, tw, pm, last, s); break;
        case 2: launch_dit<P, 2>(data, B, logn, logS, tw, pm, last, s); break;
        default: launch_dit<P, 1>(data, B, logn, logS, tw, pm, last, s); break;
    }
}
// inverse + coset-forward of one [n][B] vector with the fused middle pass; false: shape not covered (the caller runs the two transforms)
static bool to_coset_fused(const NttDomain &dom, Fr *data, size_t B, hipStream_t s) {
    using T = LdsTile<NttWords, 3, 3>;
    constexpr int TPW = ZK_NTT_TPW;
    const int logn = dom.logn;
    const size_t tiles = dom.n >> 6;
    if (logn < 12 || tiles % TPW != 0) return false;
    int sizes[8];
    const int np = balanced_passes(logn - 6, sizes);
    int loglen = logn;
    for (int i = 0; i < np; i++) { launch_dif_pass<NttWords>(sizes[i], data, B, logn, loglen, dom.tw_inv.p, nullptr, false, s); loglen -= sizes[i]; }
    static PerDeviceOnce attr;
    attr.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_mid_lds_pipe<NttWords, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)); });
    hipLaunchKernelGGL((k_ntt_mid_lds_pipe<NttWords, TPW>), dim3(ceil_div(B, T::TI), (unsigned)(tiles / TPW)), dim3(T::THREADS), T::LDS_BYTES, s, data, B, logn, dom.tw_inv.p,
                       dom.tw_fwd.p, dom.coset_pre.p);
    int logS = 6;
    for (int i = np - 1; i >= 0; i--) { launch_dit_pass<NttWords>(sizes[i], data, B, logn, logS, dom.tw_fwd.p, nullptr, logS + sizes[i] == logn, s); logS += sizes[i]; }
    return true;
}
#endif  // ZK_EXPERIMENTS

// ---------------------------------------------------------------------------------------------
// Small batches (a single Prove() call: B = 1 .. 16).  The tiled kernels above put 32 INSTANCES on the lanes of a tile, so one proof
// keeps 1 lane in 32 busy and the four transforms of a Prove() call cost 2.2 ms of pure launch latency.  Here the lanes are the
// butterflies: a workgroup owns a tile of 2^K elements (stride S) of ONE instance in LDS and runs all K stages on it, 2^(K-1) threads,
// one butterfly per thread and stage; N = 2^16 is two launches per transform (8 + 8 stages).  Same stage / twiddle indexing as
// k_ntt_dif / k_ntt_dit, same weak-form arithmetic, hence bit-identical results.
template <class P, int K>
__global__ __launch_bounds__(1 << (K > 1 ? K - 1 : 0)) void k_ntt_dif_small(Fr *__restrict__ data, size_t B, int logn, int loglen, const Fr *__restrict__ tw,
                                                                           const Fr *__restrict__ postmul, int last) {
    using E = typename P::E;
    constexpr int N = 1 << K, HALF = N / 2 ? N / 2 : 1;
    __shared__ E lds[N];
    const int tid = threadIdx.x;
    const size_t g = blockIdx.x, b = blockIdx.y;
    const int logS = loglen - K;
    const size_t S = (size_t)1 << logS;
    const size_t blk = g >> logS, p = g & (S - 1);
    const size_t base = (blk << loglen) + p;
    for (int e = tid; e < N; e += HALF) lds[e] = P::load(data[(base + ((size_t)e << logS)) * B + b]);
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < K; st++) {
        const int half = 1 << (K - 1 - st);
        const int m = ((tid / half) * 2 * half) + (tid % half);
        const size_t j = p + ((size_t)(m % half) << logS);
        const int twshift = logn - (loglen - st);
        E u = lds[m], v = lds[m + half];
        P::dif(u, v, (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
        lds[m] = u; lds[m + half] = v;
        __syncthreads();
    }
    for (int e = tid; e < N; e += HALF) {
        const size_t idx = base + ((size_t)e << logS);
        E v = lds[e];
        if (postmul) v = P::mul(v, postmul[idx]);
        data[idx * B + b] = P::store(v, last != 0);
    }
}
template <class P, int K>
__global__ __launch_bounds__(1 << (K > 1 ? K - 1 : 0)) void k_ntt_dit_small(Fr *__restrict__ data, size_t B, int logn, int logS, const Fr *__restrict__ tw,
                                                                           const Fr *__restrict__ premul, int last) {
    using E = typename P::E;
    constexpr int N = 1 << K, HALF = N / 2 ? N / 2 : 1;
    __shared__ E lds[N];
    const int tid = threadIdx.x;
    const size_t g = blockIdx.x, b = blockIdx.y;
    const size_t S = (size_t)1 << logS;
    const size_t blk = g >> logS, p = g & (S - 1);
    const size_t base = (blk << (logS + K)) + p;
    for (int e = tid; e < N; e += HALF) {
        const size_t idx = base + ((size_t)e << logS);
        E v = P::load(data[idx * B + b]);
        if (premul) v = P::mul(v, premul[idx]);
        lds[e] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < K; st++) {
        const int h = 1 << st;
        const int m = ((tid >> st) << (st + 1)) | (tid & (h - 1));
        const size_t j = p + ((size_t)(m & (h - 1)) << logS);
        const int twshift = logn - (logS + st + 1);
        E u = lds[m], v = lds[m + h];
        P::dit(u, v, (j == 0 && !P::F9) ? nullptr : &tw[j << twshift]);
        lds[m] = u; lds[m + h] = v;
        __syncthreads();
    }
    for (int e = tid; e < N; e += HALF) data[(base + ((size_t)e << logS)) * B + b] = P::store(lds[e], last != 0);
}
static constexpr size_t NTT_SMALL_B = 16;
static constexpr int NTT_SMALL_K = 9;           // stages per small-batch pass: 512 elements = 16 KiB of LDS, 256 threads
template <class P, int K>
static void launch_dif_small(Fr *data, size_t B, int logn, int loglen, const Fr *tw, const Fr *postmul, bool last, hipStream_t s) {
    hipLaunchKernelGGL((k_ntt_dif_small<P, K>), dim3((unsigned)(((size_t)1 << logn) >> K), (unsigned)B), dim3(K > 1 ? 1 << (K - 1) : 1), 0, s, data, B, logn, loglen, tw,
                       postmul, (int)last);
}
template <class P, int K>
static void launch_dit_small(Fr *data, size_t B, int logn, int logS, const Fr *tw, const Fr *premul, bool last, hipStream_t s) {
    hipLaunchKernelGGL((k_ntt_dit_small<P, K>), dim3((unsigned)(((size_t)1 << logn) >> K), (unsigned)B), dim3(K > 1 ? 1 << (K - 1) : 1), 0, s, data, B, logn, logS, tw,
                       premul, (int)last);
}
// balanced split of the stages into passes of at most NTT_SMALL_K
static int next_small_pass(int rem) {
    const int passes = (rem + NTT_SMALL_K - 1) / NTT_SMALL_K;
    return (rem + passes - 1) / passes;
}
template <class P>
static void dif_passes_small(int logn, Fr *data, size_t B, const Fr *tw, const Fr *postmul, hipStream_t s) {
    int loglen = logn;
    while (loglen > 0) {
        const int k = next_small_pass(loglen);
        const bool last = loglen - k == 0;
        const Fr *pm = last ? postmul : nullptr;
        switch (k) {
            case 9: launch_dif_small<P, 9>(data, B, logn, loglen, tw, pm, last, s); break;
            case 8: launch_dif_small<P, 8>(data, B, logn, loglen, tw, pm, last, s); break;
            case 7: launch_dif_small<P, 7>(data, B, logn, loglen, tw, pm, last, s); break;
            case 6: launch_dif_small<P, 6>(data, B, logn, loglen, tw, pm, last, s); break;
            case 5: launch_dif_small<P, 5>(data, B, logn, loglen, tw, pm, last, s); break;
            case 4: launch_dif_small<P, 4>(data, B, logn, loglen, tw, pm, last, s); break;
            case 3: launch_dif_small<P, 3>(data, B, logn, loglen, tw, pm, last, s); break;
            case 2: launch_dif_small<P, 2>(data, B, logn, loglen, tw, pm, last, s); break;
            default: launch_dif_small<P, 1>(data, B, logn, loglen, tw, pm, last, s); break;
        }
        loglen -= k;
    }
}
template <class P>
static void dit_passes_small(int logn, Fr *data, size_t B, const Fr *tw, const Fr *premul, hipStream_t s) {
    int logS = 0;
    while (logS < logn) {
        const int k = next_small_pass(logn - logS);
        const Fr *pm = (logS == 0) ? premul : nullptr;
        const bool last = logS + k == logn;
        switch (k) {
            case 9: launch_dit_small<P, 9>(data, B, logn, logS, tw, pm, last, s); break;
            case 8: launch_dit_small<P, 8>(data, B, logn, logS, tw, pm, last, s); break;
            case 7: launch_dit_small<P, 7>(data, B, logn, logS, tw, pm, last, s); break;
            case 6: launch_dit_small<P, 6>(data, B, logn, logS, tw, pm, last, s); break;
            case 5: launch_dit_small<P, 5>(data, B, logn, logS, tw, pm, last, s); break;
            case 4: launch_dit_small<P, 4>(data, B, logn, logS, tw, pm, last, s); break;
            case 3: launch_dit_small<P, 3>(data, B, logn, logS, tw, pm, last, s); break;
            case 2: launch_dit_small<P, 2>(data, B, logn, logS, tw, pm, last, s); break;
            default: launch_dit_small<P, 1>(data, B, logn, logS, tw, pm, last, s); break;
        }
        logS += k;
    }
}

// the multiplier tables of the 9-limb butterflies: same values in the 2^261-Montgomery form
const Fr *NttDomain::table9(const Fr *t) const {
    if (t == nullptr) return nullptr;
    if (t == tw_fwd.p) return tw_fwd9.p;
    if (t == tw_inv.p) return tw_inv9.p;
    if (t == coset_pre.p) return coset_pre9.p;
    if (t == coset_post.p) return coset_post9.p;
    throw ZkError("ntt: multiplier table without a 9-limb copy");
}
void NttDomain::dif(Fr *data, size_t B, bool inverse, const Fr *postmul, hipStream_t s) const {
    const Fr *tw = inverse ? tw_inv.p : tw_fwd.p;
    t_ntt_prio_bit = hi_prio ? 2 : 0;
#ifdef ZK_EXPERIMENTS
    static const int diag = [] { const char *e = getenv("ZKPROVE_NTT_DIAG"); return e ? atoi(e) : 0; }();
    if (diag == 1 || diag == 2 || diag == 9 || f9) {
        if (diag == 1) dif_passes<NttDiag<1>>(logn, data, B, tw, postmul, s);
        else if (diag == 2) dif_passes<NttDiag<2>>(logn, data, B, tw, postmul, s);
        else if (diag == 9) dif_passes<Ntt9Lazy>(logn, data, B, tw, postmul, s);
        else dif_passes<Ntt9>(logn, data, B, table9(tw), table9(postmul), s);
        ZK_HIP(hipGetLastError());
        return;
    }
#endif
    if (B <= NTT_SMALL_B) dif_passes_small<NttWords>(logn, data, B, tw, postmul, s);
    else dif_passes<NttWords>(logn, data, B, tw, postmul, s);
    ZK_HIP(hipGetLastError());
}
void NttDomain::dit(Fr *data, size_t B, bool inverse, const Fr *premul, hipStream_t s) const {
    const Fr *tw = inverse ? tw_inv.p : tw_fwd.p;
    t_ntt_prio_bit = hi_prio ? 2 : 0;
#ifdef ZK_EXPERIMENTS
    static const int diag = [] { const char *e = getenv("ZKPROVE_NTT_DIAG"); return e ? atoi(e) : 0; }();
    if (diag == 1 || diag == 2 || diag == 9 || f9) {
        if (diag == 1) dit_passes<NttDiag<1>>(logn, data, B, tw, premul, s);
        else if (diag == 2) dit_passes<NttDiag<2>>(logn, data, B, tw, premul, s);
        else if (diag == 9) dit_passes<Ntt9Lazy>(logn, data, B, tw, premul, s);
        else dit_passes<Ntt9>(logn, data, B, table9(tw), table9(premul), s);
        ZK_HIP(hipGetLastError());
        return;
    }
#endif
    if (B <= NTT_SMALL_B) dit_passes_small<NttWords>(logn, data, B, tw, premul, s);
    else dit_passes<NttWords>(logn, data, B, tw, premul, s);
    ZK_HIP(hipGetLastError());
}

#ifdef ZK_EXPERIMENTS   // gnark's unfolded computeH (7 transforms + the pointwise quotient): parity-tested in the experiments build, not part of the product
// a <- (a*b - c) * den_inv   (all n*B elements)
__global__ __launch_bounds__(256) void k_quotient_pointwise(Fr *__restrict__ a, const Fr *__restrict__ b, const Fr *__restrict__ c,
                                                            const Fr *__restrict__ consts, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    a[i] = (a[i] * b[i] - c[i].canon()) * consts[4];          // the transforms leave weak-form values (see below); '*' canonicalises, '-' needs it done
}

// computeH: a,b,c hold per-constraint values (rows >= nb_constraints zero), layout [n][B].
// On return a[rev(k)][b] = k-th coefficient of H for instance b.
void NttDomain::compute_h(Fr *a, Fr *b, Fr *c, size_t B, hipStream_t s) const {
    dif(a, B, true, nullptr, s);
    dif(b, B, true, nullptr, s);
    dif(c, B, true, nullptr, s);
    dit(a, B, false, coset_pre.p, s);     // premul = g^rev(i) / n : finishes the inverse and shifts to the coset
    dit(b, B, false, coset_pre.p, s);
    dit(c, B, false, coset_pre.p, s);
    size_t total = n * B;
    hipLaunchKernelGGL(k_quotient_pointwise, dim3(ceil_div(total, 256)), dim3(256), 0, s, a, b, c, consts.p, total);
    dif(a, B, true, coset_post.p, s);     // postmul = g^-rev(i) / n
    ZK_HIP(hipGetLastError());
}

#endif  // ZK_EXPERIMENTS

// The prover's quotient stage (see keyfold.hip): only a and b are transformed; the remaining linear maps of computeH
// live in the folded bases.
void NttDomain::to_coset(Fr *a, Fr *b, size_t B, hipStream_t s) const {
#ifdef ZK_EXPERIMENTS
    static const bool fuse = [] { const char *e = getenv("ZKPROVE_NTT_FUSE"); return e && e[0] == '1'; }();
    static const int diag = [] { const char *e = getenv("ZKPROVE_NTT_DIAG"); return e ? atoi(e) : 0; }();
    static const bool eight = [] { const char *e = getenv("ZKPROVE_NTT_EIGHT"); return e && e[0] == '1'; }();
    if (fuse && B > NTT_SMALL_B && !f9 && diag == 0 && !eight && to_coset_fused(*this, a, B, s) && to_coset_fused(*this, b, B, s)) {
        ZK_HIP(hipGetLastError());
        return;
    }
#endif
    dif(a, B, true, nullptr, s);
    dif(b, B, true, nullptr, s);
    dit(a, B, false, coset_pre.p, s);
    dit(b, B, false, coset_pre.p, s);
    ZK_HIP(hipGetLastError());
}

// plain scaling by 1/n (used by the stand-alone inverse transform entry point)
__global__ void k_scale_ninv(Fr *a, const Fr *consts, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) a[i] = a[i] * consts[3];
}
void NttDomain::scale_ninv(Fr *a, size_t B, hipStream_t s) const {
    size_t total = n * B;
    hipLaunchKernelGGL(k_scale_ninv, dim3(ceil_div(total, 256)), dim3(256), 0, s, a, consts.p, total);
}

// layout / representation helpers ---------------------------------------------------------------
__global__ void k_fr_to_mont(Fr *a, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) a[i] = a[i].to_mont();
}
__global__ void k_fr_from_mont(Fr *a, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) a[i] = a[i].from_mont();
}
void fr_to_mont(Fr *a, size_t total, hipStream_t s) { hipLaunchKernelGGL(k_fr_to_mont, dim3(ceil_div(total, 256)), dim3(256), 0, s, a, total); }
void fr_from_mont(Fr *a, size_t total, hipStream_t s) { hipLaunchKernelGGL(k_fr_from_mont, dim3(ceil_div(total, 256)), dim3(256), 0, s, a, total); }

}  // namespace zk
