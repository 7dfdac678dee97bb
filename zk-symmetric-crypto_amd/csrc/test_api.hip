// C-ABI entry points that expose single kernels of the prover for parity tests and micro-benchmarks
// (declared in include/zkprove.h, section "kernel-level entry points").  Inputs/outputs are host
// buffers of canonical little-endian 32-byte field elements in [instance][index] order; the entry
// points transpose to the device layout [index][B], convert to Montgomery form on the GPU, run the
// production kernels and convert back.
#include <string.h>
#include <vector>
#include "kernels.h"
#include "curve9.cuh"

using namespace zk;
void fp_to_mont_dev(void *p, size_t count, hipStream_t s);
void fp_from_mont_dev(void *p, size_t count, hipStream_t s);


#define ZKP_TRY try { zk::bind_default_device();
#define ZKP_CATCH(rv)                         \
    }                                         \
    catch (const std::exception &e) {         \
        zk::set_last_error(e.what());         \
        fprintf(stderr, "zkprove: %s\n", e.what()); \
        return rv;                            \
    }

static void transpose_in(std::vector<Fr> &dst, const uint8_t *src, size_t B, size_t n) {
    dst.resize(n * B);
    for (size_t b = 0; b < B; b++)
        for (size_t i = 0; i < n; i++) memcpy(&dst[i * B + b], src + (b * n + i) * 32, 32);
}
static void transpose_out(uint8_t *dst, const std::vector<Fr> &src, size_t B, size_t n) {
    for (size_t b = 0; b < B; b++)
        for (size_t i = 0; i < n; i++) memcpy(dst + (b * n + i) * 32, &src[i * B + b], 32);
}
static uint32_t brev(uint32_t v, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) if (v & (1u << i)) r |= 1u << (bits - 1 - i);
    return r;
}


// mode 0: forward via DIF, 1: inverse via DIF, 2: forward via DIT, 3: inverse via DIT.  natural order in/out.
extern "C" int zkp_ntt(int logn, long batch, uint8_t *data, int mode) {
    ZKP_TRY
    const size_t n = (size_t)1 << logn, B = (size_t)batch;
    NttDomain dom;
    dom.init(logn, 0);
    std::vector<Fr> h, tmp(n * B);
    transpose_in(h, data, B, n);
    const bool inverse = mode & 1, use_dit = mode & 2;
    if (use_dit) {   // DIT consumes bit-reversed input
        for (size_t i = 0; i < n; i++) memcpy(&tmp[(size_t)brev((uint32_t)i, logn) * B], &h[i * B], B * sizeof(Fr));
        h.swap(tmp);
    }
    DevBuf<Fr> d(n * B);
    d.upload(h.data(), n * B);
    fr_to_mont(d.p, n * B, 0);
    if (use_dit) dom.dit(d.p, B, inverse, nullptr, 0);
    else dom.dif(d.p, B, inverse, nullptr, 0);
    if (inverse) dom.scale_ninv(d.p, B, 0);
    fr_from_mont(d.p, n * B, 0);
    d.download(h.data(), n * B);
    ZK_HIP(hipDeviceSynchronize());
    if (!use_dit) {  // DIF produces bit-reversed output
        for (size_t i = 0; i < n; i++) memcpy(&tmp[(size_t)brev((uint32_t)i, logn) * B], &h[i * B], B * sizeof(Fr));
        h.swap(tmp);
    }
    transpose_out(data, h, B, n);
    return 0;
    ZKP_CATCH(-1)
}

#ifdef ZK_EXPERIMENTS
// a,b,c: [batch][ncons] canonical; h_out: [batch][N] canonical, natural coefficient order
extern "C" int zkp_compute_h(int logn, long batch, long ncons, const uint8_t *a, const uint8_t *b, const uint8_t *c, uint8_t *h_out) {
    ZKP_TRY
    const size_t n = (size_t)1 << logn, B = (size_t)batch;
    NttDomain dom;
    dom.init(logn, 0);
    DevBuf<Fr> d[3];
    const uint8_t *src[3] = {a, b, c};
    std::vector<Fr> h(n * B);
    for (int k = 0; k < 3; k++) {
        memset(h.data(), 0, n * B * sizeof(Fr));
        for (size_t bb = 0; bb < B; bb++)
            for (size_t i = 0; i < (size_t)ncons; i++) memcpy(&h[i * B + bb], src[k] + (bb * ncons + i) * 32, 32);
        d[k].alloc(n * B);
        d[k].upload(h.data(), n * B);
        ZK_HIP(hipDeviceSynchronize());
        fr_to_mont(d[k].p, n * B, 0);
    }
    dom.compute_h(d[0].p, d[1].p, d[2].p, B, 0);
    fr_from_mont(d[0].p, n * B, 0);
    d[0].download(h.data(), n * B);
    ZK_HIP(hipDeviceSynchronize());
    std::vector<Fr> nat(n * B);
    for (size_t i = 0; i < n; i++) memcpy(&nat[(size_t)brev((uint32_t)i, logn) * B], &h[i * B], B * sizeof(Fr));
    transpose_out(h_out, nat, B, n);
    return 0;
    ZKP_CATCH(-1)
}
#endif  // ZK_EXPERIMENTS

template <class F>
static int msm_test(long n_, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c, bool dense = false, const uint8_t *wcomb_cls = nullptr,
                    int wcomb_k = 0) {
    const size_t n = (size_t)n_, B = (size_t)batch;
    constexpr size_t PB = sizeof(Affine<F>);          // 64 or 128
    // points arrive canonical LE (x|y) resp. (x0|x1|y0|y1): upload and convert to Montgomery on device
    DevBuf<Affine<F>> dp(n ? n : 1);
    if (n) ZK_HIP(hipMemcpy(dp.p, points, n * PB, hipMemcpyHostToDevice));
    fp_to_mont_dev(dp.p, n * (PB / 32), 0);
    std::vector<uint32_t> rows(n);
    for (size_t j = 0; j < n; j++) rows[j] = (uint32_t)j;
    MsmBases<F> mb;
    msm_bases_build(mb, dp.p, n, dense ? nullptr : rows.data(), c, 0);
    mb.dense_stride = n;
    std::vector<Fr> hs;
    if (dense) { hs.resize(n * B); memcpy(hs.data(), scalars, n * B * 32); }      // dense path reads the instance-major layout directly
    else transpose_in(hs, scalars, B, n);
    DevBuf<Fr> ds(n * B + 1);
    ds.upload(hs.data(), n * B);
    fr_to_mont(ds.p, n * B, 0);
    MsmScratch sc;
    sc.reserve(B, n * mb.W, (size_t)1 << (c - 1), msm_part_stride(n, c));
    sc.reserve_compact(B, n);
    if (dense) { sc.reserve_dense(B, n * mb.W); sc.reserve_tree(B, n * mb.W, (size_t)1 << (c - 1)); }
    sc.reserve_sums(B, sizeof(F) == 32 ? (size_t)1 << (c - 1) : 1, sizeof(F) == 32 ? 1 : (size_t)1 << (c - 1));
    if (wcomb_cls) {           // wire comb over the positions of class 1..3 (msm.hip, msm_attach_wire_comb)
        if (!msm_attach_wire_comb(mb, wcomb_cls, wcomb_k, 0)) return 1;
        size_t pw = 0, ab = 0, rb = 0;
        msm_wire_comb_needs(mb, B, pw, ab, rb);
        sc.reserve_wire_comb(pw, ab, rb);
    }
    DevBuf<XYZZ<F>> res(B);
    DevBuf<Affine<F>> aff(B);
    msm_batch(mb, ds.p, B, sc, res.p, 0);
    if constexpr (sizeof(F) == 32) g1_to_affine_canonical((const G1XYZZ *)res.p, (G1Affine *)aff.p, B, 0);
    else g2_to_affine_canonical((const G2XYZZ *)res.p, (G2Affine *)aff.p, B, 0);
    ZK_HIP(hipMemcpy(out, aff.p, B * PB, hipMemcpyDeviceToHost));
    return 0;
}
// points: n*64 canonical LE (x|y); scalars [batch][n]; out: batch*64 canonical LE affine, (0,0)=inf
extern "C" int zkp_msm_g1(long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c) {
    ZKP_TRY
    return msm_test<Fp>(n, batch, points, scalars, out, c);
    ZKP_CATCH(-1)
}
// wire MSM with a wire comb attached: cls[n] = width class per base (0: 0 / +-1, 1: < 2^8, 2: < 2^32, 3: any), clumps of k bases.
// group 1 = G1 (points n*64), 2 = G2 (points n*128).  returns 1 when a table cannot be built (degenerate bases)
extern "C" int zkp_msm_wire_comb(int group, long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c, int k, const uint8_t *cls) {
    ZKP_TRY
    if (group == 2) return msm_test<Fp2>(n, batch, points, scalars, out, c, false, cls, k);
    return msm_test<Fp>(n, batch, points, scalars, out, c, false, cls, k);
    ZKP_CATCH(-1)
}
// dense (quotient-MSM) path: every scalar goes through the bucket pipeline, scalars read instance-major
extern "C" int zkp_msm_g1_dense(long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c) {
    ZKP_TRY
    return msm_test<Fp>(n, batch, points, scalars, out, c, true);
    ZKP_CATCH(-1)
}
// quotient-MSM comb path (msm_comb.hip) on its own: points n*64 canonical LE, scalars [batch][n], clumps of k bases.
// returns 0 ok, 1 = the bases are degenerate for the comb (a point at infinity / a coinciding subset sum): nothing computed
// mode 1: the scalars are passed as the products of two matrices (x = scalars, y = all ones), the prover's call shape
extern "C" int zkp_msm_comb(long n_, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int k, int mode) {
    ZKP_TRY
    const size_t n = (size_t)n_, B = (size_t)batch;
    DevBuf<G1Affine> dp(n ? n : 1);
    if (n) ZK_HIP(hipMemcpy(dp.p, points, n * 64, hipMemcpyHostToDevice));
    fp_to_mont_dev(dp.p, n * 2, 0);
    CombBases cb;
    // mode bits 8.. : n1 = number of leading clumps that hold k + 1 bases (mixed tables, CombShape::n1)
    const size_t n1 = (size_t)(mode >> 8);
    const bool split_fold = (mode & 0x10) != 0;          // comb_msm_split: two walk launches, the upper planes' fold on a second stream
    mode &= 0x0f;
    if (n1 * (size_t)(k + 1) > n) return -1;
    if (!comb_build(cb, dp.p, n, k, 0, n1)) return 1;
    std::vector<Fr> hs;
    transpose_in(hs, scalars, B, n);
    DevBuf<Fr> ds(n * B + 1), ones;
    ds.upload(hs.data(), n * B);
    fr_to_mont(ds.p, n * B, 0);
    if (mode == 1) {
        std::vector<Fr> h1(n * B, Fr::one());
        ones.upload(h1.data(), n * B);
    }
    CombScratch sc;
    sc.reserve(cb, B, 1);
    DevBuf<G1XYZZ> res(B);
    DevBuf<G1Affine> aff(B);
    comb_patterns(cb, ds.p, mode == 1 ? ones.p : nullptr, B, sc.pat[0].p, 0);
    if (split_fold) {
        hipStream_t side;
        hipEvent_t ev[2];
        ZK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        ZK_HIP(hipEventCreate(&ev[0])); ZK_HIP(hipEventCreate(&ev[1]));
        ZK_HIP(hipDeviceSynchronize());                  // (the patterns above ran on the null stream; `side` does not wait for it by itself)
        comb_msm_split(cb, sc.pat[0].p, B, sc, res.p, 0, side, ev);
        ZK_HIP(hipDeviceSynchronize());
        (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]); (void)hipStreamDestroy(side);
    } else
        comb_msm(cb, sc.pat[0].p, B, sc, res.p, 0);
    g1_to_affine_canonical(res.p, aff.p, B, 0);
    ZK_HIP(hipMemcpy(out, aff.p, B * 64, hipMemcpyDeviceToHost));
    return 0;
    ZKP_CATCH(-1)
}
// timing of the comb path at production shape without a key: n = 2^logn bases j * G, pseudo-random 254-bit scalars.
// out_ms[0..4] = table build, patterns, walk, fold (both stages), table bytes / 2^30.  Returns 0, 1 = degenerate, -1 error.
namespace zk {
__global__ void k_bench_points(G1Affine *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine g{Fp::one(), Fp::one() + Fp::one()};
    uint32_t k[8];
    uint64_t x = i * 0xD6E8FEB86659FD93ull + 0xABCDEF01ull;          // unstructured 253-bit multiples of G (structured ones make subset sums coincide)
#pragma unroll
    for (int w = 0; w < 8; w++) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        k[w] = (uint32_t)x;
    }
    k[7] &= 0x1fffffffu;
    out[i] = scalar_mul(g, k, 254).to_affine();
}
__global__ void k_bench_scalars(Fr *out, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    Fr v;
    uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        v.l[k] = (uint32_t)x;
    }
    v.l[7] &= 0x1fffffffu;
    out[i] = v;
}
}
extern "C" int zkp_bench_comb(int logn, long batch, int k, int iters, double *out_ms) {
    ZKP_TRY
    const size_t n = (size_t)1 << logn, B = (size_t)batch;
    DevBuf<G1Affine> pts(n);
    hipLaunchKernelGGL(k_bench_points, dim3(ceil_div(n, 64)), dim3(64), 0, 0, pts.p, n);
    DevBuf<Fr> x(n * B);
    hipLaunchKernelGGL(k_bench_scalars, dim3(ceil_div(n * B, 256)), dim3(256), 0, 0, x.p, n * B);
    hipEvent_t e[5];
    for (auto &ev : e) ZK_HIP(hipEventCreate(&ev));
    CombBases cb;
    ZK_HIP(hipEventRecord(e[0], 0));
    const bool ok = comb_build(cb, pts.p, n, k, 0);
    ZK_HIP(hipEventRecord(e[1], 0));
    ZK_HIP(hipEventSynchronize(e[1]));
    float ms;
    ZK_HIP(hipEventElapsedTime(&ms, e[0], e[1]));
    out_ms[0] = ms;
    out_ms[4] = (double)comb_table_bytes(n, k) / (double)(1ull << 30);
    if (!ok) return 1;
    CombScratch sc;
    sc.reserve(cb, B, 1);
    DevBuf<G1XYZZ> res(B);
    double acc[3] = {0, 0, 0};
    for (int it = 0; it < iters + 1; it++) {
        ZK_HIP(hipEventRecord(e[0], 0));
        comb_patterns(cb, x.p, x.p, B, sc.pat[0].p, 0);          // the prover's call shape: products of two matrices
        ZK_HIP(hipEventRecord(e[1], 0));
        comb_msm(cb, sc.pat[0].p, B, sc, res.p, 0, &e[2]);
        ZK_HIP(hipEventRecord(e[4], 0));
        ZK_HIP(hipEventSynchronize(e[4]));
        if (it == 0) continue;          // warm-up
        ZK_HIP(hipEventElapsedTime(&ms, e[0], e[1])); acc[0] += ms;
        ZK_HIP(hipEventElapsedTime(&ms, e[2], e[3])); acc[1] += ms;
        ZK_HIP(hipEventElapsedTime(&ms, e[3], e[4])); acc[2] += ms;
    }
    for (int i = 0; i < 3; i++) out_ms[1 + i] = acc[i] / iters;
    for (auto &ev : e) (void)hipEventDestroy(ev);
    return 0;
    ZKP_CATCH(-1)
}
// points: n*128 canonical LE (x.a0|x.a1|y.a0|y.a1)
extern "C" int zkp_msm_g2(long n, long batch, const uint8_t *points, const uint8_t *scalars, uint8_t *out, int c) {
    ZKP_TRY
    return msm_test<Fp2>(n, batch, points, scalars, out, c);
    ZKP_CATCH(-1)
}

namespace zk {
__global__ void k_fp_to_mont(Fp *a, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) a[i] = a[i].to_mont();
}
}
namespace zk {
__global__ void k_fp_from_mont(Fp *a, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) a[i] = a[i].from_mont();
}
}
void fp_from_mont_dev(void *p, size_t count, hipStream_t s) {
    if (count) hipLaunchKernelGGL(zk::k_fp_from_mont, dim3(ceil_div(count, 256)), dim3(256), 0, s, (Fp *)p, count);
}
void fp_to_mont_dev(void *p, size_t count, hipStream_t s) {
    if (count) hipLaunchKernelGGL(zk::k_fp_to_mont, dim3(ceil_div(count, 256)), dim3(256), 0, s, (Fp *)p, count);
}

// key folding (keyfold.hip): z = pk.G1.Z, (2^logn - 1) * 64 canonical LE (x|y) -> ze: 2^logn * 64, zc: ncons * 64
extern "C" int zkp_fold_quotient_bases(int logn, long ncons, const uint8_t *z, uint8_t *ze, uint8_t *zc) {
    ZKP_TRY
    const size_t N = (size_t)1 << logn, nc = (size_t)ncons;
    if (nc > N || nc == 0) throw ZkError("zkp_fold_quotient_bases: bad constraint count");
    NttDomain dom;
    dom.init(logn, 0);
    DevBuf<G1Affine> dz(N), dze(N), dzc(nc);
    ZK_HIP(hipMemcpy(dz.p, z, (N - 1) * 64, hipMemcpyHostToDevice));
    fp_to_mont_dev(dz.p, (N - 1) * 2, 0);
    fold_quotient_bases(dz.p, dom, nc, dze.p, dzc.p, 0);
    hipLaunchKernelGGL(zk::k_fp_from_mont, dim3(ceil_div(N * 2, 256)), dim3(256), 0, 0, (Fp *)dze.p, N * 2);
    hipLaunchKernelGGL(zk::k_fp_from_mont, dim3(ceil_div(nc * 2, 256)), dim3(256), 0, 0, (Fp *)dzc.p, nc * 2);
    ZK_HIP(hipMemcpy(ze, dze.p, N * 64, hipMemcpyDeviceToHost));
    ZK_HIP(hipMemcpy(zc, dzc.p, nc * 64, hipMemcpyDeviceToHost));
    return 0;
    ZKP_CATCH(-1)
}

// key folding, second half (keyfold.hip): K'_w = K_w + sum over the terms of wire w of coeff * Zc[cons].
// zc: nzc*64, k: nk*64 canonical LE points of the wires rows_k[nk] (ascending); terms: nterms*(wire, cons, coeff index) u32 triples;
// coeffs: ncoeff*32 canonical LE scalars.  out_rows (cap nk + nterms) / out_points receive the folded set; returns its size or -1.
extern "C" long zkp_fold_output_matrix(long nzc, const uint8_t *zc, long nk, const uint8_t *k, const uint32_t *rows_k, long nterms, const uint32_t *terms,
                                       long ncoeff, const uint8_t *coeffs, uint32_t *out_rows, uint8_t *out_points) {
    ZKP_TRY
    DevBuf<G1Affine> dzc((size_t)nzc + 1), dk((size_t)nk + 1);
    ZK_HIP(hipMemcpy(dzc.p, zc, (size_t)nzc * 64, hipMemcpyHostToDevice));
    if (nk) ZK_HIP(hipMemcpy(dk.p, k, (size_t)nk * 64, hipMemcpyHostToDevice));
    fp_to_mont_dev(dzc.p, (size_t)nzc * 2, 0);
    fp_to_mont_dev(dk.p, (size_t)nk * 2, 0);
    DevBuf<Fr> dco((size_t)ncoeff + 1);
    ZK_HIP(hipMemcpy(dco.p, coeffs, (size_t)ncoeff * 32, hipMemcpyHostToDevice));
    fr_to_mont(dco.p, (size_t)ncoeff, 0);
    std::vector<OutputTerm> tv((size_t)nterms);
    for (long i = 0; i < nterms; i++) tv[i] = OutputTerm{terms[3 * i], terms[3 * i + 1], terms[3 * i + 2]};
    std::vector<uint32_t> rk(rows_k, rows_k + nk), rows;
    DevBuf<G1Affine> outp;
    fold_output_matrix(dzc.p, dk.p, rk, std::move(tv), dco.p, outp, rows, 0);
    hipLaunchKernelGGL(zk::k_fp_from_mont, dim3(ceil_div(rows.size() * 2 + 1, 256)), dim3(256), 0, 0, (Fp *)outp.p, rows.size() * 2);
    ZK_HIP(hipMemcpy(out_points, outp.p, rows.size() * 64, hipMemcpyDeviceToHost));
    memcpy(out_rows, rows.data(), rows.size() * 4);
    return (long)rows.size();
    ZKP_CATCH(-1)
}

// -------- micro-benchmarks: raw field-multiplication / point-addition throughput on the device
namespace zk {
template <class F>
struct Nine9;
template <>
struct Nine9<Fp> { using F9 = Fp9; using X9 = G1X9; };
template <>
struct Nine9<Fp2> { using F9 = Fp29; using X9 = G2X9; };
template <class F>
__global__ __launch_bounds__(256) void k_bench_mul(F *x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    F a = x[i], b = x[i + 1];
    for (int k = 0; k < iters; k++) { a = a * b; b = b * a; }
    x[i] = a + b;
}
// issue rate of the multiplier's two instructions: PAIRS = false -> v_mad_u64_u32 alone, true -> the v_mad_u64_u32 + v_addc_co_u32
// carry pair of the Montgomery block; 4 independent accumulators, 64 of them per loop iteration
template <bool PAIRS>
__global__ __launch_bounds__(256) void k_bench_mad(uint32_t *x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = x[i], b = x[i] * 2654435761u + 1u;
    uint64_t acc0 = a, acc1 = b, acc2 = a ^ b, acc3 = a + b;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (PAIRS) {
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc0), "+v"(c0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc1), "+v"(c1) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc2), "+v"(c2) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc3), "+v"(c3) : "v"(a), "v"(b) : "vcc");
            } else {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    x[i] = (uint32_t)(acc0 ^ acc1 ^ acc2 ^ acc3) ^ (uint32_t)((acc0 ^ acc1 ^ acc2 ^ acc3) >> 32) ^ c0 ^ c1 ^ c2 ^ c3;
}
// the 9-limb field of the bucket kernels (field9.cuh): product chain, G1 / G2 mixed-addition chains
__global__ __launch_bounds__(256) void k_bench_mul9(Fp *x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fp9 a = Fp9::from_words(x[i]), b = Fp9::from_words(x[i + 1]);
    for (int k = 0; k < iters; k++) { a = Fp9::mul(a, b); b = Fp9::mul(b, a); }
    x[i] = Fp9::to_words(Fp9::norm(Fp9::add(a, b)));
}
template <class F>
__global__ __launch_bounds__(64, sizeof(F) == 64 ? 2 : 4) void k_bench_madd9(typename Nine9<F>::X9 *acc, const Affine<F> *pts, int iters) {
    using F9 = typename Nine9<F>::F9;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typename Nine9<F>::X9 a = acc[i];
    for (int k = 0; k < iters; k++) {
        const Affine<F> p = pts[(i * 7 + k) & 1023];
        a.madd(F9::from_words(p.x), F9::cneg(F9::from_words(p.y), (k & 4) ? -1 : 0));
    }
    acc[i] = a;
}
// a radix-8 DIF butterfly group (3 stages, 12 butterflies) in registers, as the transform kernels run it: kind 11 on the 8 x 32 weak
// form (ntt.hip, NttWords), kind 12 on 9 signed 29-bit limbs kept across stages (sum: limb-wise + one carry propagation, difference:
// limb-wise, product: Field9::mul against a pre-limbed twiddle) -- the question is the ratio of their butterfly rates
__global__ __launch_bounds__(256) void k_bench_bfly_words(Fr *x, const Fr *tw, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = x[i * 8 + m];
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int half = 4 >> st;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if ((m / half) & 1) continue;
                const Fr s = Fr::add_w(v[m], v[m + half]), d = Fr::sub_w(v[m], v[m + half]);
                v[m] = s;
                v[m + half] = Fr::mul_wc(d, tw[(k + st * 4 + (m % half)) & 63]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; m++) x[i * 8 + m] = v[m];
}
__device__ __noinline__ Fr9 bench_mul9_call(Fr9 a, Fr9 b) { return Fr9::mul(a, b); }
template <bool CALL>
__global__ __launch_bounds__(256) void k_bench_bfly_9(Fr *x, const Fr9 *tw, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr9 v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = Fr9::from_words(x[i * 8 + m]);
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int st = 0; st < 3; st++) {
            const int half = 4 >> st;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if ((m / half) & 1) continue;
                const Fr9 s = Fr9::norm(Fr9::add(v[m], v[m + half])), d = Fr9::sub(v[m], v[m + half]);
                v[m] = s;
                const Fr9 t = tw[(k + st * 4 + (m % half)) & 63];
                v[m + half] = CALL ? bench_mul9_call(d, t) : Fr9::mul(d, t);
            }
        }
        // once per "pass" (here: per group) the sums are brought back near [0, M): top limb estimate, q * M subtracted
        if ((k & 1) == 1) {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int32_t q = v[m].l[8] >> 21;
                int64_t c = 0;
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    c += (int64_t)v[m].l[j] - (int64_t)q * Fr9Params::M[j];
                    if (j < 8) { v[m].l[j] = (int32_t)(c & 0x1fffffff); c >>= 29; } else v[m].l[j] = (int32_t)c;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; m++) x[i * 8 + m] = Fr9::to_words(Fr9::norm(v[m]));
}
// radix-4 groups (2 stages, 4 butterflies per 4 elements): what VERDICT r4 task 6 asks to try before any pass kernel -- four 9-limb
// elements are 36 registers (the radix-8 group's eight were parked in scratch by the compiler).  kind 14: 9 signed limbs kept across the
// two stages, inline product; kind 15: the 8 x 32 weak form with the same grouping (the reference for the ratio).  4 butterflies / iteration.
__global__ __launch_bounds__(256) void k_bench_bfly4_words(Fr *x, const Fr *tw, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr v[4];
#pragma unroll
    for (int m = 0; m < 4; m++) v[m] = x[i * 4 + m];
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int st = 0; st < 2; st++) {
            const int half = 2 >> st;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if ((m / half) & 1) continue;
                const Fr s = Fr::add_w(v[m], v[m + half]), d = Fr::sub_w(v[m], v[m + half]);
                v[m] = s;
                v[m + half] = Fr::mul_wc(d, tw[(k + st * 2 + (m % half)) & 63]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) x[i * 4 + m] = v[m];
}
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_bench_bfly4_9(Fr *x, const Fr9 *tw, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr9 v[4];
#pragma unroll
    for (int m = 0; m < 4; m++) v[m] = Fr9::from_words(x[i * 4 + m]);
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int st = 0; st < 2; st++) {
            const int half = 2 >> st;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if ((m / half) & 1) continue;
                const Fr9 s = Fr9::norm(Fr9::add(v[m], v[m + half])), d = Fr9::sub(v[m], v[m + half]);
                v[m] = s;
                v[m + half] = Fr9::mul(d, tw[(k + st * 2 + (m % half)) & 63]);
            }
        }
        // once per three groups (= one 6-stage tile visit of the transform kernels) the sums are brought back near [0, M)
        if (k % 3 == 2) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int32_t q = v[m].l[8] >> 21;
                int64_t c = 0;
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    c += (int64_t)v[m].l[j] - (int64_t)q * Fr9Params::M[j];
                    if (j < 8) { v[m].l[j] = (int32_t)(c & 0x1fffffff); c >>= 29; } else v[m].l[j] = (int32_t)c;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) x[i * 4 + m] = Fr9::to_words(Fr9::norm(v[m]));
}
__global__ __launch_bounds__(256) void k_bench_mad_i64(uint32_t *x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t a = (int32_t)x[i], b = (int32_t)(x[i] * 2654435761u + 1u);
    int64_t acc0 = a, acc1 = b, acc2 = a ^ b, acc3 = a + b;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b) : "vcc");
        }
    }
    x[i] = (uint32_t)(acc0 ^ acc1 ^ acc2 ^ acc3) ^ (uint32_t)((acc0 ^ acc1 ^ acc2 ^ acc3) >> 32);
}
// iters dependent inversions per thread: GCD = false -> Fermat ladder (inv), true -> binary GCD (inv_gcd)
template <class F, bool GCD>
__global__ __launch_bounds__(256) void k_bench_inv(F *x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    F a = x[i];
    for (int k = 0; k < iters; k++) a = (GCD ? a.inv_gcd() : a.inv_fermat()) + F::one();
    x[i] = a;
}
// out[i] = in[i]^-1 by the binary-GCD inversion (Montgomery form in and out)
template <class F>
__global__ __launch_bounds__(64) void k_field_inv_gcd(const F *in, F *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i].inv_gcd();
}
template <class F>
__global__ __launch_bounds__(256) void k_bench_madd(XYZZ<F> *acc, const Affine<F> *pts, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ<F> a = acc[i];
    for (int k = 0; k < iters; k++) a.madd(pts[(i + k) & 1023]);
    acc[i] = a;
}
}
// in / out: n canonical little-endian 32-byte elements of Fp (field 0) or Fr (field 1); out[i] = in[i]^-1 mod m (0 -> 0), computed on
// the device by Field::inv_gcd (conversion to and from Montgomery form included)
extern "C" int zkp_field_inv(int field, long n, const uint8_t *in, uint8_t *out) {
    ZKP_TRY
    const size_t N = (size_t)n;
    if (field == 0) {
        DevBuf<Fp> d(N), o(N);
        ZK_HIP(hipMemcpy(d.p, in, N * 32, hipMemcpyHostToDevice));
        fp_to_mont_dev(d.p, N, 0);
        hipLaunchKernelGGL(k_field_inv_gcd<Fp>, dim3(ceil_div(N, 64)), dim3(64), 0, 0, d.p, o.p, N);
        fp_from_mont_dev(o.p, N, 0);
        ZK_HIP(hipMemcpy(out, o.p, N * 32, hipMemcpyDeviceToHost));
    } else {
        DevBuf<Fr> d(N), o(N);
        ZK_HIP(hipMemcpy(d.p, in, N * 32, hipMemcpyHostToDevice));
        fr_to_mont(d.p, N, 0);
        hipLaunchKernelGGL(k_field_inv_gcd<Fr>, dim3(ceil_div(N, 64)), dim3(64), 0, 0, d.p, o.p, N);
        fr_from_mont(o.p, N, 0);
        ZK_HIP(hipMemcpy(out, o.p, N * 32, hipMemcpyDeviceToHost));
    }
    ZK_HIP(hipDeviceSynchronize());
    return 0;
    ZKP_CATCH(-1)
}
// returns milliseconds for `threads` threads x `iters` iterations; kind 0: 2 Fp mul / iter, 1: G1 madd, 2: G2 madd, 3: Fp Fermat inversion,
// 4: Fp binary-GCD inversion, 5: 64 v_mad_u64_u32 per iteration, 6: 64 v_mad_u64_u32 + v_addc_co_u32 pairs per iteration,
// 7: 2 products of the 9-limb field / iter, 8: G1 mixed addition on the 9-limb field, 9: G2 on its tower, 10: 64 v_mad_i64_i32 per iteration
extern "C" double zkp_microbench(int kind, long threads, int iters) {
    ZKP_TRY
    size_t T = (size_t)threads;
    hipEvent_t e0, e1;
    ZK_HIP(hipEventCreate(&e0)); ZK_HIP(hipEventCreate(&e1));
    float ms = 0;
    if (kind == 0) {
        DevBuf<Fp> x(T + 1);
        std::vector<Fp> h(T + 1);
        for (size_t i = 0; i <= T; i++) for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
        x.upload(h.data(), T + 1);
        hipLaunchKernelGGL(k_bench_mul<Fp>, dim3(T / 256), dim3(256), 0, 0, x.p, 4);
        ZK_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bench_mul<Fp>, dim3(T / 256), dim3(256), 0, 0, x.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 5 || kind == 6) {
        DevBuf<uint32_t> x(T);
        ZK_HIP(hipMemset(x.p, 0x5a, T * 4));
        ZK_HIP(hipEventRecord(e0, 0));
        if (kind == 5) hipLaunchKernelGGL((k_bench_mad<false>), dim3(T / 256), dim3(256), 0, 0, x.p, iters);
        else hipLaunchKernelGGL((k_bench_mad<true>), dim3(T / 256), dim3(256), 0, 0, x.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 7) {
        DevBuf<Fp> x(T + 1);
        std::vector<Fp> h(T + 1);
        for (size_t i = 0; i <= T; i++) for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
        x.upload(h.data(), T + 1);
        hipLaunchKernelGGL(k_bench_mul9, dim3(T / 256), dim3(256), 0, 0, x.p, 4);
        ZK_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bench_mul9, dim3(T / 256), dim3(256), 0, 0, x.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 14 || kind == 15 || kind == 16) {
        DevBuf<Fr> x(T * 4);
        std::vector<Fr> h(T * 4);
        for (size_t i = 0; i < T * 4; i++) for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
        x.upload(h.data(), T * 4);
        DevBuf<Fr> tw(64);
        DevBuf<Fr9> tw9(64);
        std::vector<Fr9> h9(64);
        for (int i = 0; i < 64; i++) h9[i] = Fr9::from_words(h[i]);
        tw.upload(h.data(), 64); tw9.upload(h9.data(), 64);
        ZK_HIP(hipEventRecord(e0, 0));
        if (kind == 15) hipLaunchKernelGGL(k_bench_bfly4_words, dim3(T / 256), dim3(256), 0, 0, x.p, tw.p, iters);
        else if (kind == 14) hipLaunchKernelGGL(k_bench_bfly4_9<1>, dim3(T / 256), dim3(256), 0, 0, x.p, tw9.p, iters);
        else hipLaunchKernelGGL(k_bench_bfly4_9<2>, dim3(T / 256), dim3(256), 0, 0, x.p, tw9.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 11 || kind == 12 || kind == 13) {
        DevBuf<Fr> x(T * 8);
        std::vector<Fr> h(T * 8);
        for (size_t i = 0; i < T * 8; i++) for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
        x.upload(h.data(), T * 8);
        DevBuf<Fr> tw(64);
        DevBuf<Fr9> tw9(64);
        std::vector<Fr9> h9(64);
        for (int i = 0; i < 64; i++) h9[i] = Fr9::from_words(h[i]);
        tw.upload(h.data(), 64); tw9.upload(h9.data(), 64);
        ZK_HIP(hipEventRecord(e0, 0));
        if (kind == 11) hipLaunchKernelGGL(k_bench_bfly_words, dim3(T / 256), dim3(256), 0, 0, x.p, tw.p, iters);
        else if (kind == 12) hipLaunchKernelGGL(k_bench_bfly_9<false>, dim3(T / 256), dim3(256), 0, 0, x.p, tw9.p, iters);
        else hipLaunchKernelGGL(k_bench_bfly_9<true>, dim3(T / 256), dim3(256), 0, 0, x.p, tw9.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 10) {
        DevBuf<uint32_t> x(T);
        ZK_HIP(hipMemset(x.p, 0x5a, T * 4));
        ZK_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bench_mad_i64, dim3(T / 256), dim3(256), 0, 0, x.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 8) {
        DevBuf<G1X9> acc(T); acc.zero();
        // 1024 distinct points k * (1, 2), in the 2^261-Montgomery form the 9-limb kernels read
        std::vector<G1Affine> hp(1024);
        G1Affine g{Fp::one(), Fp::one() + Fp::one()};
        G1XYZZ run = G1XYZZ::inf();
        Fp k32;
        for (int j = 0; j < 8; j++) k32.l[j] = Fp9Params::SCALE32[j];
        for (auto &p : hp) { run.madd(g); const G1Affine a = run.to_affine(); p.x = a.x * k32; p.y = a.y * k32; }
        DevBuf<G1Affine> pts(1024); pts.upload(hp.data(), 1024);
        hipLaunchKernelGGL(k_bench_madd9<Fp>, dim3(T / 64), dim3(64), 0, 0, acc.p, pts.p, 2);
        ZK_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bench_madd9<Fp>, dim3(T / 64), dim3(64), 0, 0, acc.p, pts.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 9) {
        DevBuf<G2X9> acc(T); acc.zero();
        std::vector<G2Affine> hp(1024);      // arbitrary coordinate pairs: the formulas do not care, the timing neither
        for (size_t i = 0; i < hp.size(); i++) {
            Fp v = Fp::one();
            for (size_t k = 0; k < (i & 15) + 1; k++) v = v + v + Fp::one();
            hp[i] = G2Affine{Fp2{v, v + Fp::one()}, Fp2{v + v, v * v}};
        }
        DevBuf<G2Affine> pts(1024); pts.upload(hp.data(), 1024);
        hipLaunchKernelGGL(k_bench_madd9<Fp2>, dim3(T / 64), dim3(64), 0, 0, acc.p, pts.p, 2);
        ZK_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bench_madd9<Fp2>, dim3(T / 64), dim3(64), 0, 0, acc.p, pts.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 3 || kind == 4) {
        DevBuf<Fp> x(T);
        std::vector<Fp> h(T);
        for (size_t i = 0; i < T; i++) for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u + 17u) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
        x.upload(h.data(), T);
        ZK_HIP(hipEventRecord(e0, 0));
        if (kind == 3) hipLaunchKernelGGL((k_bench_inv<Fp, false>), dim3(T / 256), dim3(256), 0, 0, x.p, iters);
        else hipLaunchKernelGGL((k_bench_inv<Fp, true>), dim3(T / 256), dim3(256), 0, 0, x.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else if (kind == 1) {
        DevBuf<G1XYZZ> acc(T); acc.zero();
        std::vector<G1Affine> hp(1024);
        // points = small multiples of the generator computed on device is overkill here: use (1,2) for all (madd hits the doubling path once, then generic)
        for (auto &p : hp) { p.x = Fp::one(); p.y = Fp::one() + Fp::one(); }
        DevBuf<G1Affine> pts(1024); pts.upload(hp.data(), 1024);
        hipLaunchKernelGGL(k_bench_madd<Fp>, dim3(T / 256), dim3(256), 0, 0, acc.p, pts.p, 2);
        ZK_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bench_madd<Fp>, dim3(T / 256), dim3(256), 0, 0, acc.p, pts.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    } else {
        DevBuf<G2XYZZ> acc(T); acc.zero();
        std::vector<G2Affine> hp(1024);
        for (auto &p : hp) { p.x = Fp2::one(); p.y = Fp2{Fp::one() + Fp::one(), Fp::one()}; }
        DevBuf<G2Affine> pts(1024); pts.upload(hp.data(), 1024);
        hipLaunchKernelGGL(k_bench_madd<Fp2>, dim3(T / 256), dim3(256), 0, 0, acc.p, pts.p, 2);
        ZK_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bench_madd<Fp2>, dim3(T / 256), dim3(256), 0, 0, acc.p, pts.p, iters);
        ZK_HIP(hipEventRecord(e1, 0));
    }
    ZK_HIP(hipEventSynchronize(e1));
    ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms;
    ZKP_CATCH(-1.0)
}

// milliseconds per to_coset() of a and b (the prover's 4 transforms) over `batch` instances of 2^logn elements, mean of iters
extern "C" double zkp_bench_to_coset(int logn, long batch, int iters) {
    ZKP_TRY
    const size_t n = (size_t)1 << logn, B = (size_t)batch;
    NttDomain dom;
    dom.init(logn, 0);
    DevBuf<Fr> a(n * B), b(n * B);
    ZK_HIP(hipMemset(a.p, 0x11, n * B * sizeof(Fr)));
    ZK_HIP(hipMemset(b.p, 0x07, n * B * sizeof(Fr)));
    hipEvent_t e0, e1;
    ZK_HIP(hipEventCreate(&e0)); ZK_HIP(hipEventCreate(&e1));
    dom.to_coset(a.p, b.p, B, 0);
    ZK_HIP(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) dom.to_coset(a.p, b.p, B, 0);
    ZK_HIP(hipEventRecord(e1, 0));
    ZK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / iters;
    ZKP_CATCH(-1.0)
}

#ifdef ZK_EXPERIMENTS
// ---- debug harness: device tree kernel vs. host emulation of the same code on host-sorted input (prints per-bucket diffs)
#include "msm_tree.cuh"
#include <random>
#include <array>
#include <algorithm>
namespace zk {
template <class F, bool DENSE>
__global__ void k_msm_tree(const Affine<F> *__restrict__ table, const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ entries,
                           size_t max_entries, int c, size_t B, uint32_t nlanes, Affine<F> *__restrict__ bufA, size_t capA,
                           Affine<F> *__restrict__ bufB, size_t capB, Affine<F> *__restrict__ fin, uint8_t *__restrict__ scratch, uint32_t dbg);
}
static int dbg_signed_digit(const uint32_t *l, int w, int c, int &carry) {
    const int bit = w * c; const int li = bit >> 5, sh = bit & 31;
    uint64_t v = li < 8 ? l[li] : 0; if (li + 1 < 8) v |= (uint64_t)l[li + 1] << 32;
    int raw = (int)((v >> sh) & ((1u << c) - 1)) + carry;
    if (raw > (1 << (c - 1))) { carry = 1; return raw - (1 << c); }
    carry = 0; return raw;
}
extern "C" int zkp_debug_tree(int n, int c, int piece, int dbg) {
    ZKP_TRY
    const int W = (254 + c - 1) / c; const uint32_t nb = 1u << (c - 1);
    std::mt19937_64 rng(42);
    G1Affine g{Fp::one(), Fp::one() + Fp::one()};
    std::vector<G1Affine> pts(n);
    for (int j = 0; j < n; j++) { uint32_t k[8] = {(uint32_t)rng(), (uint32_t)rng(), 0, 0, 0, 0, 0, 0}; pts[j] = scalar_mul(g, k, 64).to_affine(); }
    std::vector<G1Affine> table((size_t)n * W);
    for (int j = 0; j < n; j++) { XYZZ<Fp> cur = XYZZ<Fp>::from_affine(pts[j]); for (int w = 0; w < W; w++) { table[(size_t)j * W + w] = cur.to_affine(); for (int k = 0; k < c; k++) cur = cur.dbl(); } }
    std::vector<std::array<uint32_t, 8>> sc(n);
    for (auto &s : sc) { for (int i = 0; i < 8; i++) s[i] = (uint32_t)rng(); s[7] &= 0x0fffffff; }
    std::vector<std::vector<uint32_t>> buckets(nb);
    for (int j = 0; j < n; j++) { int carry = 0; for (int w = 0; w < W; w++) { int d = dbg_signed_digit(sc[j].data(), w, c, carry); if (!d) continue; uint32_t neg = d < 0; buckets[(neg ? -d : d) - 1].push_back((uint32_t)(j * W + w) | (neg << 31)); } }
    std::vector<uint32_t> bs(nb + 1), ent;
    for (uint32_t k = 0; k < nb; k++) { bs[k] = (uint32_t)ent.size(); ent.insert(ent.end(), buckets[k].begin(), buckets[k].end()); }
    bs[nb] = (uint32_t)ent.size();
    const size_t M = ent.size();
    const uint32_t nlanes = (uint32_t)((M + piece - 1) / piece);
    const size_t capA = tree_capacity(M, nb, nlanes, 1), capB = tree_capacity(M, nb, nlanes, 2);
    std::vector<G1Affine> A(capA), Bf(capB), fin(nb), fin_dev(nb);
    std::vector<Fp> pref(TREE_CH); std::vector<uint4> desc(TREE_CH);
    for (uint32_t lane = 0; lane < nlanes; lane++) {
        uint32_t k0, k1; tree_piece_bounds(bs.data(), nb, lane, nlanes, k0, k1);
        tree_piece<Fp>(table.data(), bs.data(), ent.data(), k0, k1, lane, A.data(), Bf.data(), fin.data(), pref.data(), desc.data(), 1);
    }
    DevBuf<G1Affine> dt(table.size()), dA(capA), dB(capB), dfin(nb);
    DevBuf<uint32_t> dbs(nb + 1), dent(M + 1);
    dt.upload(table.data(), table.size()); dbs.upload(bs.data(), nb + 1); dent.upload(ent.data(), M);
    const unsigned grid = (dbg & 8) ? nlanes : ceil_div(nlanes, 64);
    DevBuf<uint8_t> dscr((size_t)grid * TREE_CH * 64 * tree_slot_bytes<Fp>());
    ZK_HIP(hipMemset(dfin.p, 0xff, nb * sizeof(G1Affine)));
    hipLaunchKernelGGL((k_msm_tree<Fp, true>), dim3(grid), dim3(64), 0, 0, dt.p, dbs.p, dent.p, M, c, (size_t)1, nlanes, dA.p, capA, dB.p, capB, dfin.p, dscr.p, (uint32_t)dbg);
    ZK_HIP(hipDeviceSynchronize());
    dfin.download(fin_dev.data(), nb);
    ZK_HIP(hipDeviceSynchronize());
    int bad = 0, hist[16] = {};
    for (uint32_t k = 0; k < nb; k++) {
        if (buckets[k].size() < 2) continue;
        if (!(fin[k].x == fin_dev[k].x && fin[k].y == fin_dev[k].y)) {
            bad++;
            int lg = 0; while ((1u << lg) < buckets[k].size()) lg++;
            hist[lg < 15 ? lg : 15]++;
            if (bad <= 8) fprintf(stderr, "bucket %u n=%zu differs (dev x limb0 %08x host %08x)\n", k, buckets[k].size(), fin_dev[k].x.l[0], fin[k].x.l[0]);
        }
    }
    fprintf(stderr, "debug_tree n=%d c=%d piece=%d dbg=%d entries=%zu lanes=%u: %d bad buckets; by ceil(log2 n):", n, c, piece, dbg, M, nlanes, bad);
    for (int i = 0; i < 16; i++) fprintf(stderr, " %d", hist[i]);
    fprintf(stderr, "\n");
    return bad;
    ZKP_CATCH(-1)
}
#endif  // ZK_EXPERIMENTS

// ------------------------------------------------------------------------------------ pins (tests/test_pins.py, tests/test_gpu_pins.py)
#include "hostutil.h"
#include "xmd.cuh"
namespace zk {
int pairing_product_check(const uint8_t *g1_raw, const uint8_t *g2_raw, size_t n, bool on_device);      // verifier.cpp
int vk_pairing_identities(const uint8_t *vk_bytes, size_t len, bool on_device);
}
// prod e(P_i, Q_i) == 1 through the verifier's host pairing (device = 0) or with the Miller loops in the batch verifier's kernel
// (device = 1); points in gnark-crypto's uncompressed encoding.  1 / 0, -1: undecodable / error
extern "C" int zkp_pairing_product_check(const uint8_t *g1_raw, const uint8_t *g2_raw, long n, int device) {
    try {
        return zk::pairing_product_check(g1_raw, g2_raw, (size_t)n, device != 0);
    } catch (const std::exception &e) { zk::set_last_error(e.what()); return -1; }
}
// bit 0: e(beta1, G2) == e(G1, beta2), bit 1: e(delta1, G2) == e(G1, delta2), for a key in gnark's VerifyingKey.WriteTo layout; -1: does not parse
extern "C" int zkp_vk_pairing_identities(const uint8_t *vk, long len, int device) {
    try {
        return zk::vk_pairing_identities(vk, (size_t)len, device != 0);
    } catch (const std::exception &e) { zk::set_last_error(e.what()); return -1; }
}
__global__ void k_xmd_test(const uint8_t *msg, uint32_t mlen, const uint8_t *dst, uint32_t dlen, uint8_t *out, uint32_t out_len) {
    if (threadIdx.x == 0 && blockIdx.x == 0) expand_message_xmd_dev(msg, mlen, dst, dlen, out, out_len);
}
// RFC 9380 expand_message_xmd(SHA-256): device = 0 the host function behind hostutil.cpp::hash_to_fr (verifier), device = 1 the device
// function behind k_bsb22_challenge (prover)
extern "C" int zkp_expand_message_xmd(const uint8_t *msg, long mlen, const uint8_t *dst, long dlen, uint8_t *out, long out_len, int device) {
    if (mlen < 0 || dlen < 0 || dlen > 255 || out_len < 0 || out_len > 255 * 32) return -1;
    if (!device) return zk::expand_message_xmd(msg, (size_t)mlen, dst, (size_t)dlen, out, (size_t)out_len) ? 0 : -1;
    ZKP_TRY
    DevBuf<uint8_t> dm((size_t)mlen + 1), dd((size_t)dlen + 1), dout((size_t)out_len + 1);
    if (mlen) ZK_HIP(hipMemcpy(dm.p, msg, (size_t)mlen, hipMemcpyHostToDevice));
    if (dlen) ZK_HIP(hipMemcpy(dd.p, dst, (size_t)dlen, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_xmd_test, dim3(1), dim3(64), 0, 0, dm.p, (uint32_t)mlen, dd.p, (uint32_t)dlen, dout.p, (uint32_t)out_len);
    ZK_HIP(hipGetLastError());
    if (out_len) ZK_HIP(hipMemcpy(out, dout.p, (size_t)out_len, hipMemcpyDeviceToHost));
    return 0;
    ZKP_CATCH(-1)
}
// the prover's challenge kernel on n commitments: com = n x (x | y) canonical little-endian 32-byte coordinates -> n canonical LE Fr
extern "C" int zkp_bsb22_challenge(const uint8_t *com, long n, uint8_t *out) {
    ZKP_TRY
    const size_t N = (size_t)n;
    DevBuf<G1Affine> dc(N);
    DevBuf<Fr> dr(N);
    ZK_HIP(hipMemcpy(dc.p, com, N * 64, hipMemcpyHostToDevice));
    bsb22_challenge_device(dc.p, dr.p, N, 0);
    fr_from_mont(dr.p, N, 0);
    ZK_HIP(hipMemcpy(out, dr.p, N * 32, hipMemcpyDeviceToHost));
    return 0;
    ZKP_CATCH(-1)
}
// host-side counterpart (verifier.cpp's path): hash_to_fr(X || Y big-endian, "bsb22-commitment")
extern "C" int zkp_bsb22_challenge_host(const uint8_t *com_be64, long n, uint8_t *out) {
    for (long i = 0; i < n; i++) {
        const zk::U256 v = zk::hash_to_fr(com_be64 + 64 * i, 64, "bsb22-commitment");
        memcpy(out + 32 * i, v.l, 32);
    }
    return 0;
}

// ---- request layer (cabi.cpp / request.cpp): decoder agreement and the device-free host-stage benchmark
namespace zk {
int request_decode_both(const char *p, size_t n);
int host_stage_bench(const char *data, size_t len, int iters, double *out);
void debug_leftovers(int id, long *out3);
void debug_fetch(int id, const uint32_t *instances, size_t n, uint8_t *wires, uint8_t *h);
void debug_solver_plan(int id, long *out4);
}
extern "C" int zkp_debug_solver_plan(int id, long *out4) {
    ZKP_TRY
    zk::debug_solver_plan(id, out4);
    return 0;
    ZKP_CATCH(-1)
}
// ---- parity-test views of the current resident batch (state of THIS library: the batch must have been uploaded and stepped through it)
extern "C" int zkp_debug_leftovers(int id, long *out3) {
    ZKP_TRY
    zk::debug_leftovers(id, out3);
    return 0;
    ZKP_CATCH(-1)
}
extern "C" int zkp_debug_fetch(int id, const uint32_t *instances, long n, uint8_t *wires, uint8_t *h) {
    ZKP_TRY
    zk::debug_fetch(id, instances, (size_t)(n < 0 ? 0 : n), wires, h);
    return 0;
    ZKP_CATCH(-1)
}
// 1: the single-pass scanner took the request and agrees with the DOM reader, 0: it declined, -1: the two disagree
extern "C" int zkp_request_decode_both(const uint8_t *json, long len) { return zk::request_decode_both((const char *)json, (size_t)(len < 0 ? 0 : len)); }
// host stages of a streamed ProveBatch without a device: out4 = {decode stage requests/s, serialise stage requests/s, requests the scanner
// took, requests that produced a record}
extern "C" int zkp_host_stage_bench(const uint8_t *json_array, long len, int iters, double *out4) {
    try {
        return zk::host_stage_bench((const char *)json_array, (size_t)len, iters, out4);
    } catch (const std::exception &e) { zk::set_last_error(e.what()); return -1; }
}
