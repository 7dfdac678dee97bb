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
// compute_h (all 7 transforms) is kept as the kernel-level entry point zkp_compute_h for parity tests.
#include "kernels.h"

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
    ZK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// DIF pass: K stages starting at block length `len` (natural -> towards bit-reversed).
// thread (g, b): g in [0, n >> K), b in [0, B)
template <int K>
__global__ __launch_bounds__(256) void k_ntt_dif(Fr *__restrict__ data, size_t B, int logn, int loglen,
                                                 const Fr *__restrict__ tw, const Fr *__restrict__ postmul) {
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
    Fr x[R];
#pragma unroll
    for (int m = 0; m < R; m++) x[m] = data[(base + ((size_t)m << logS)) * B + b];
#pragma unroll
    for (int st = 0; st < K; st++) {
        const int half = 1 << (K - 1 - st);          // pair distance in m units
        const int logL = loglen - st;                // this stage's block length
        const int twshift = logn - logL;
#pragma unroll
        for (int m = 0; m < R; m++) {
            if ((m / half) & 1) continue;
            const size_t j = p + ((size_t)(m % half) << logS);
            Fr u = x[m], v = x[m + half];
            x[m] = Fr::add_w(u, v);
            Fr d = Fr::sub_w(u, v);
            x[m + half] = (j == 0) ? d : Fr::mul_wc(d, tw[j << twshift]);
        }
    }
#pragma unroll
    for (int m = 0; m < R; m++) {
        const size_t idx = base + ((size_t)m << logS);
        Fr v = x[m];
        if (postmul) v = Fr::mul_wc(v, postmul[idx]);
        data[idx * B + b] = v;
    }
}

// DIT pass: K stages starting at half-distance S = 2^logS (bit-reversed -> natural).
template <int K>
__global__ __launch_bounds__(256) void k_ntt_dit(Fr *__restrict__ data, size_t B, int logn, int logS,
                                                 const Fr *__restrict__ tw, const Fr *__restrict__ premul) {
    constexpr int R = 1 << K;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = t % B;
    const size_t g = t / B;
    const size_t groups = ((size_t)1 << logn) >> K;
    if (g >= groups) return;
    const size_t S = (size_t)1 << logS;
    const size_t blk = g >> logS, p = g & (S - 1);
    const size_t base = (blk << (logS + K)) + p;
    Fr x[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        const size_t idx = base + ((size_t)m << logS);
        Fr v = data[idx * B + b];
        if (premul) v = Fr::mul_wc(v, premul[idx]);
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
            Fr u = x[m];
            Fr v = (j == 0) ? x[m + h] : Fr::mul_wc(x[m + h], tw[j << twshift]);
            x[m] = Fr::add_w(u, v);
            x[m + h] = Fr::sub_w(u, v);
        }
    }
#pragma unroll
    for (int m = 0; m < R; m++) data[(base + ((size_t)m << logS)) * B + b] = x[m];
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
template <int K1, int K2>
struct LdsTile {
    static constexpr int K = K1 + K2;
    static constexpr int E = 1 << K;               // elements per tile
    static constexpr int TOTAL = ZK_NTT_TILE;
    static constexpr int TI = TOTAL / E;           // instances per tile
    static constexpr int THREADS = TOTAL / 8;
    static constexpr int LDS_BYTES = TOTAL * 32;
    static_assert(E * TI == TOTAL && TI >= 1, "tile shape");
};

// DIF: stages loglen .. loglen-K+1 (block lengths len, len/2, ...).  Element e of the tile sits at
// pos = blk*len + p + e*S with S = len >> K.
template <int K1, int K2>
__global__ __launch_bounds__(ZK_NTT_TILE / 8) void k_ntt_dif_lds(Fr *__restrict__ data, size_t B, int logn, int loglen, const Fr *__restrict__ tw,
                                                     const Fr *__restrict__ postmul) {
    using T = LdsTile<K1, K2>;
    constexpr int K = T::K, E = T::E, TI = T::TI, R1 = 1 << K1, R2 = 1 << K2, NT = T::THREADS;
    extern __shared__ uint4 lds_raw[];
    Fr *lds = (Fr *)lds_raw;                       // [E][TI]
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
            Fr x[R1];
#pragma unroll
            for (int m = 0; m < R1; m++) {
                const size_t e = (size_t)q + (size_t)STR * m;
                x[m] = active ? data[(base + (e << logS)) * B + b] : Fr::zero();
            }
#pragma unroll
            for (int st = 0; st < K1; st++) {
                const int half = 1 << (K1 - 1 - st);
                const int twshift = logn - (loglen - st);
#pragma unroll
                for (int m = 0; m < R1; m++) {
                    if ((m / half) & 1) continue;
                    const size_t j = p + (((size_t)q + (size_t)STR * (m % half)) << logS);
                    Fr u = x[m], v = x[m + half];
                    x[m] = Fr::add_w(u, v);
                    Fr d = Fr::sub_w(u, v);
                    x[m + half] = (j == 0) ? d : Fr::mul_wc(d, tw[j << twshift]);
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
            Fr x[R2];
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
                    Fr u = x[m], v = x[m + half];
                    x[m] = Fr::add_w(u, v);
                    Fr d = Fr::sub_w(u, v);
                    x[m + half] = (j == 0) ? d : Fr::mul_wc(d, tw[j << twshift]);
                }
            }
            if (active) {
#pragma unroll
                for (int m = 0; m < R2; m++) {
                    const size_t idx = base + (((size_t)q2 * R2 + m) << logS);
                    Fr v = x[m];
                    if (postmul) v = Fr::mul_wc(v, postmul[idx]);
                    data[idx * B + b] = v;
                }
            }
        }
    }
}

// DIT: stages with half-distances S, 2S, ... (S = 2^logS), tile element e at pos = blk*(S<<K) + p + e*S.
template <int K1, int K2>
__global__ __launch_bounds__(ZK_NTT_TILE / 8) void k_ntt_dit_lds(Fr *__restrict__ data, size_t B, int logn, int logS, const Fr *__restrict__ tw,
                                                     const Fr *__restrict__ premul) {
    using T = LdsTile<K1, K2>;
    constexpr int K = T::K, E = T::E, TI = T::TI, R1 = 1 << K1, R2 = 1 << K2, NT = T::THREADS;
    extern __shared__ uint4 lds_raw[];
    Fr *lds = (Fr *)lds_raw;
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
            Fr x[R1];
#pragma unroll
            for (int m = 0; m < R1; m++) {
                const size_t idx = base + (((size_t)q * R1 + m) << logS);
                Fr v = active ? data[idx * B + b] : Fr::zero();
                if (premul) v = Fr::mul_wc(v, premul[idx]);
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
                    Fr u = x[m];
                    Fr v = (j == 0) ? x[m + h] : Fr::mul_wc(x[m + h], tw[j << twshift]);
                    x[m] = Fr::add_w(u, v);
                    x[m + h] = Fr::sub_w(u, v);
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
            Fr x[R2];
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
                    Fr u = x[m];
                    Fr v = (j == 0) ? x[m + h] : Fr::mul_wc(x[m + h], tw[j << twshift]);
                    x[m] = Fr::add_w(u, v);
                    x[m + h] = Fr::sub_w(u, v);
                }
            }
            if (active) {
#pragma unroll
                for (int m = 0; m < R2; m++) data[(base + (((size_t)q2 + (size_t)R1 * m) << logS)) * B + b] = x[m];
            }
        }
    }
}

template <int K1, int K2>
static void launch_dif_lds(Fr *data, size_t B, int logn, int loglen, const Fr *tw, const Fr *postmul, hipStream_t s) {
    using T = LdsTile<K1, K2>;
    static PerDeviceOnce attr;
    attr.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_dif_lds<K1, K2>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)); });
    dim3 grid(ceil_div(B, T::TI), (unsigned)(((size_t)1 << logn) >> T::K));
    hipLaunchKernelGGL((k_ntt_dif_lds<K1, K2>), grid, dim3(T::THREADS), T::LDS_BYTES, s, data, B, logn, loglen, tw, postmul);
}
template <int K1, int K2>
static void launch_dit_lds(Fr *data, size_t B, int logn, int logS, const Fr *tw, const Fr *premul, hipStream_t s) {
    using T = LdsTile<K1, K2>;
    static PerDeviceOnce attr;
    attr.run([] { ZK_HIP(hipFuncSetAttribute((const void *)k_ntt_dit_lds<K1, K2>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES)); });
    dim3 grid(ceil_div(B, T::TI), (unsigned)(((size_t)1 << logn) >> T::K));
    hipLaunchKernelGGL((k_ntt_dit_lds<K1, K2>), grid, dim3(T::THREADS), T::LDS_BYTES, s, data, B, logn, logS, tw, premul);
}

template <int K>
static void launch_dif(Fr *data, size_t B, int logn, int loglen, const Fr *tw, const Fr *postmul, hipStream_t s) {
    size_t threads = (((size_t)1 << logn) >> K) * B;
    hipLaunchKernelGGL(k_ntt_dif<K>, dim3(ceil_div(threads, 256)), dim3(256), 0, s, data, B, logn, loglen, tw, postmul);
}
template <int K>
static void launch_dit(Fr *data, size_t B, int logn, int logS, const Fr *tw, const Fr *premul, hipStream_t s) {
    size_t threads = (((size_t)1 << logn) >> K) * B;
    hipLaunchKernelGGL(k_ntt_dit<K>, dim3(ceil_div(threads, 256)), dim3(256), 0, s, data, B, logn, logS, tw, premul);
}


// stage schedule: as many 6-stage LDS passes as possible, then one 5- or 4-stage LDS pass, then register passes
static int next_pass(int rem) {
    if (rem >= 6) return 6;
    return rem;     // 5, 4 -> LDS (3,2)/(2,2); 3, 2, 1 -> register kernels
}
void NttDomain::dif(Fr *data, size_t B, bool inverse, const Fr *postmul, hipStream_t s) const {
    const Fr *tw = inverse ? tw_inv.p : tw_fwd.p;
    int loglen = logn;
    while (loglen > 0) {
        const int k = next_pass(loglen);
        const Fr *pm = (loglen - k == 0) ? postmul : nullptr;
        switch (k) {
            case 6: launch_dif_lds<3, 3>(data, B, logn, loglen, tw, pm, s); break;
            case 5: launch_dif_lds<3, 2>(data, B, logn, loglen, tw, pm, s); break;
            case 4: launch_dif_lds<2, 2>(data, B, logn, loglen, tw, pm, s); break;
            case 3: launch_dif<3>(data, B, logn, loglen, tw, pm, s); break;
            case 2: launch_dif<2>(data, B, logn, loglen, tw, pm, s); break;
            default: launch_dif<1>(data, B, logn, loglen, tw, pm, s); break;
        }
        loglen -= k;
    }
    ZK_HIP(hipGetLastError());
}
void NttDomain::dit(Fr *data, size_t B, bool inverse, const Fr *premul, hipStream_t s) const {
    const Fr *tw = inverse ? tw_inv.p : tw_fwd.p;
    int logS = 0;
    while (logS < logn) {
        const int k = next_pass(logn - logS);
        const Fr *pm = (logS == 0) ? premul : nullptr;
        switch (k) {
            case 6: launch_dit_lds<3, 3>(data, B, logn, logS, tw, pm, s); break;
            case 5: launch_dit_lds<3, 2>(data, B, logn, logS, tw, pm, s); break;
            case 4: launch_dit_lds<2, 2>(data, B, logn, logS, tw, pm, s); break;
            case 3: launch_dit<3>(data, B, logn, logS, tw, pm, s); break;
            case 2: launch_dit<2>(data, B, logn, logS, tw, pm, s); break;
            default: launch_dit<1>(data, B, logn, logS, tw, pm, s); break;
        }
        logS += k;
    }
    ZK_HIP(hipGetLastError());
}

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

// The prover's quotient stage (see keyfold.hip): only a and b are transformed; the remaining linear maps of computeH
// live in the folded bases.
void NttDomain::to_coset(Fr *a, Fr *b, size_t B, hipStream_t s) const {
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
