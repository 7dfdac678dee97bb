// Experiment: BN254 Fp on 9 signed 29-bit limbs (Montgomery radix 2^261) in plain C++ -- column sums of v_mad_i64_i32 with no
// carry instructions, lazy (carry-free) additions / subtractions, and products that share ONE reduction (a*b - c*d).
// Checks the arithmetic against the 8 x 32 field of field.cuh and times both (multiplication chain and the XYZZ mixed addition).
#include "../../zk-symmetric-crypto_amd/csrc/curve.cuh"
#include "../../zk-symmetric-crypto_amd/csrc/field9_consts.inc"
#define F9_P ZK_F9_FP_M
#define F9_PINV ZK_F9_FP_PINV
#define F9_CIN ZK_F9_FP_CIN
#define F9_COUT ZK_F9_FP_COUT
#define F9_ONE ZK_F9_FP_ONE
#include <stdio.h>
#include <vector>
using namespace zk;

struct F9 {
    int32_t l[9];
};
__device__ constexpr int32_t P9[9] = F9_P;
__device__ constexpr int32_t CIN9[9] = F9_CIN;
__device__ constexpr int32_t COUT9[9] = F9_COUT;
__device__ constexpr int32_t ONE9[9] = F9_ONE;
constexpr uint32_t MASK29 = 0x1fffffffu;

__device__ __forceinline__ void mac(int64_t &acc, int32_t a, int32_t b) {
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mac_s(int64_t &acc, int32_t a, int32_t b) {      // b: uniform constant
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc");
}
// (sum over t < NP of sgn_t * a_t * b_t) / 2^261 mod M, result strict (limbs 0..7 in [0, 2^29), limb 8 signed)
template <int NP>
__device__ __forceinline__ F9 mont9(const F9 *const (&a)[NP], const F9 *const (&b)[NP]) {
    int64_t acc = 0;
    int32_t m[9];
    F9 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int t = 0; t < NP; t++)
#pragma unroll
            for (int i = 0; i <= k; i++) mac(acc, a[t]->l[i], b[t]->l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mac_s(acc, m[i], P9[k - i]);
        m[k] = (int32_t)(((uint32_t)acc * F9_PINV) & MASK29);
        mac_s(acc, m[k], P9[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int t = 0; t < NP; t++)
#pragma unroll
            for (int i = k - 8; i < 9; i++) mac(acc, a[t]->l[i], b[t]->l[k - i]);
#pragma unroll
        for (int i = k - 8; i < 9; i++) mac_s(acc, m[i], P9[k - i]);
        r.l[k - 9] = (int32_t)((uint32_t)acc & MASK29);
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    return r;
}
__device__ __forceinline__ F9 mul9(const F9 &a, const F9 &b) {
    const F9 *const pa[1] = {&a};
    const F9 *const pb[1] = {&b};
    return mont9<1>(pa, pb);
}
// a*b - c*d with one reduction
__device__ __forceinline__ F9 mulsub9(const F9 &a, const F9 &b, const F9 &c, const F9 &d) {
    F9 nc;
#pragma unroll
    for (int i = 0; i < 9; i++) nc.l[i] = -c.l[i];
    const F9 *const pa[2] = {&a, &nc};
    const F9 *const pb[2] = {&b, &d};
    return mont9<2>(pa, pb);
}
__device__ __forceinline__ F9 sub9(const F9 &a, const F9 &b) {      // lazy: no carries
    F9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
__device__ __forceinline__ F9 norm9(F9 x) {                          // signed carry propagation -> strict
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t c = x.l[i] >> 29;
        x.l[i] &= (int32_t)MASK29;
        x.l[i + 1] += c;
    }
    return x;
}
// strict x: value == 0 mod M for a value in (-M, 2M)
__device__ __forceinline__ bool is_zero_mod9(const F9 &x) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { z |= (uint32_t)x.l[i]; e |= (uint32_t)(x.l[i] ^ P9[i]); }
    return z == 0 || e == 0;
}
// re-limb the 8 words of a canonical value
__device__ __forceinline__ F9 limbs9(const Fp &x) {
    F9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint32_t v;
        if (sh == 0) v = x.l[w];
        else if (w + 1 < 8) v = __builtin_amdgcn_alignbit(x.l[w + 1], x.l[w], sh);
        else v = x.l[w] >> sh;
        r.l[i] = (int32_t)(v & MASK29);
    }
    return r;
}
// strict value in (-2M, 3M) -> canonical words
__device__ Fp words9(F9 x) {
#pragma unroll
    for (int i = 0; i < 9; i++) x.l[i] += 2 * P9[i];
    x = norm9(x);                              // in (0, 5M)
#pragma unroll
    for (int s = 2; s >= 0; s--) {             // subtract 4M, 2M, M when that stays non-negative
        F9 y;
#pragma unroll
        for (int i = 0; i < 9; i++) y.l[i] = x.l[i] - (P9[i] << s);
        y = norm9(y);
        if (y.l[8] >= 0) x = y;
    }
    uint32_t out[9] = {0};
    uint64_t acc = 0;
    int have = 0, o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc |= (uint64_t)(uint32_t)x.l[i] << have;
        have += 29;
        if (have >= 32) { out[o++] = (uint32_t)acc; acc >>= 32; have -= 32; }
    }
    out[o] = (uint32_t)acc;
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = out[i];
    return r;
}
__device__ __forceinline__ F9 const9(const int32_t (&c)[9]) {
    F9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c[i];
    return r;
}
// 2^256-Montgomery words <-> 2^261-Montgomery limbs
__device__ __forceinline__ F9 to9(const Fp &x) { return mul9(limbs9(x), const9(CIN9)); }
__device__ Fp from9(const F9 &x) { return words9(mul9(x, const9(COUT9))); }

struct XYZZ9 {
    F9 X, Y, ZZ, ZZZ;
    __device__ __forceinline__ bool is_inf() const {
        uint32_t z = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) z |= (uint32_t)ZZ.l[i];
        return z == 0;
    }
    // q: canonical words of the 2^261-Montgomery form
    __device__ __forceinline__ void madd(const G1Affine &q) {
        if (q.is_inf()) return;
        const F9 x2 = limbs9(q.x), y2 = limbs9(q.y);
        if (is_inf()) { X = x2; Y = y2; ZZ = const9(ONE9); ZZZ = const9(ONE9); return; }
        const F9 U2 = mul9(x2, ZZ), S2 = mul9(y2, ZZZ);
        const F9 Pv = sub9(U2, X), R = sub9(S2, Y);
        const F9 PP = mul9(Pv, Pv);
        if (is_zero_mod9(PP)) {        // doubling / cancellation: not in the timed loop's data
            X = x2; Y = y2; ZZ = const9(ONE9); ZZZ = const9(ONE9);
            return;
        }
        const F9 PPP = mul9(Pv, PP), Q = mul9(X, PP), RR = mul9(R, R);
        F9 X3;
#pragma unroll
        for (int i = 0; i < 9; i++) X3.l[i] = RR.l[i] - PPP.l[i] - 2 * Q.l[i];
        X3 = norm9(X3);
        const F9 T = sub9(Q, X3);
        Y = mulsub9(R, T, Y, PPP);
        X = X3;
        ZZ = mul9(ZZ, PP);
        ZZZ = mul9(ZZZ, PPP);
    }
};

__global__ __launch_bounds__(256) void k_check_mul(const Fp *x, Fp *got, Fp *exp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = x[i], b = x[i + 1], c = x[i + 2], d = x[i + 3];
    F9 A = to9(a), B = to9(b), C = to9(c), D = to9(d);
    F9 u = sub9(A, C), v = sub9(B, D);                  // lazy operands
    got[2 * i] = from9(mul9(u, v));
    exp[2 * i] = (a - c) * (b - d);
    got[2 * i + 1] = from9(mulsub9(u, B, C, v));
    exp[2 * i + 1] = (a - c) * b - c * (b - d);
}
__global__ __launch_bounds__(256) void k_bench9(const Fp *x, Fp *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    F9 a = limbs9(x[i]), b = limbs9(x[i + 1]);
    for (int k = 0; k < iters; k++) { a = mul9(a, b); b = mul9(b, a); }
    out[i] = words9(sub9(a, b));
}
__global__ __launch_bounds__(256) void k_bench32(Fp *x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = x[i], b = x[i + 1];
    for (int k = 0; k < iters; k++) { a = Fp::mul_w(a, b); b = Fp::mul_w(b, a); }
    x[i] = Fp::add_w(a, b).canon();
}
// the same chain of mixed additions in both representations; pts32 = 2^256-Montgomery words, pts9 = 2^261-Montgomery words
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_madd9(G1XYZZ *out, const G1Affine *pts9, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ9 a;
    a.X = a.Y = a.ZZ = a.ZZZ = F9{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
    for (int k = 0; k < iters; k++) a.madd(pts9[(i * 7 + k) & 1023]);
    out[i] = G1XYZZ{from9(a.X), from9(a.Y), from9(a.ZZ), from9(a.ZZZ)};
}
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_madd32(G1XYZZ *out, const G1Affine *pts, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    G1XYZZ a = G1XYZZ::inf();
    for (int k = 0; k < iters; k++) a.madd_w<true>(pts[(i * 7 + k) & 1023]);
    out[i] = a.canon();
}
__global__ void k_make_points(G1Affine *pts, G1Affine *pts9, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // (i+1) * G, G = (1, 2)
    G1Affine g{Fp::one(), Fp::one() + Fp::one()};
    uint32_t k[8] = {(uint32_t)(i * 2654435761u + 12345u), (uint32_t)i + 77u, 0, 0, 0, 0, 0, 0};
    G1Affine p = scalar_mul(g, k, 64).to_affine();
    pts[i] = p;
    // 2^261 form: multiply by 2^5
    Fp c32 = Fp::one();
    for (int t = 0; t < 5; t++) c32 = c32 + c32;
    pts9[i] = G1Affine{p.x * c32, p.y * c32};
}

int main() {
    size_t T = 256 * 256 * 16;
    std::vector<Fp> h(T + 4), g(2 * T), e(2 * T);
    for (size_t i = 0; i < T + 4; i++)
        for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u + 12345u * k * k) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
    Fp *d, *dg, *de;
    hipMalloc(&d, (T + 4) * 32); hipMalloc(&dg, 2 * T * 32); hipMalloc(&de, 2 * T * 32);
    hipMemcpy(d, h.data(), (T + 4) * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check_mul, dim3(T / 256), dim3(256), 0, 0, d, dg, de);
    hipMemcpy(g.data(), dg, 2 * T * 32, hipMemcpyDeviceToHost); hipMemcpy(e.data(), de, 2 * T * 32, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < 2 * T; i++) bad += !(g[i] == e[i]);
    printf("mul9 / mulsub9 mismatches: %zu / %zu\n", bad, 2 * T);

    G1Affine *pts, *pts9; hipMalloc(&pts, 1024 * 64); hipMalloc(&pts9, 1024 * 64);
    hipLaunchKernelGGL(k_make_points, dim3(4), dim3(256), 0, 0, pts, pts9, 1024);
    size_t TM = 256 * 256 * 8;
    G1XYZZ *o9, *o32; hipMalloc(&o9, TM * 128); hipMalloc(&o32, TM * 128);
    hipLaunchKernelGGL(k_madd9<2>, dim3(TM / 256), dim3(256), 0, 0, o9, pts9, 37);
    hipLaunchKernelGGL(k_madd32<2>, dim3(TM / 256), dim3(256), 0, 0, o32, pts, 37);
    std::vector<G1XYZZ> r9(TM), r32(TM);
    hipMemcpy(r9.data(), o9, TM * 128, hipMemcpyDeviceToHost); hipMemcpy(r32.data(), o32, TM * 128, hipMemcpyDeviceToHost);
    bad = 0;
    for (size_t i = 0; i < TM; i++) bad += !(r9[i].X == r32[i].X && r9[i].Y == r32[i].Y && r9[i].ZZ == r32[i].ZZ && r9[i].ZZZ == r32[i].ZZZ);
    printf("madd9 vs madd32 mismatches after 37 additions: %zu / %zu\n", bad, TM);

    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_bench9, dim3(T / 256), dim3(256), 0, 0, d, dg, 1000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("mul9   %8.3f ms  %7.2f G mul/s\n", ms, (double)T * 2000 / ms / 1e6);
        hipEventRecord(e0); hipLaunchKernelGGL(k_bench32, dim3(T / 256), dim3(256), 0, 0, d, 1000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("mul32w %8.3f ms  %7.2f G mul/s\n", ms, (double)T * 2000 / ms / 1e6);
    }
#define BM(W) \
    hipEventRecord(e0); hipLaunchKernelGGL(k_madd9<W>, dim3(TM / 256), dim3(256), 0, 0, o9, pts9, 400); hipEventRecord(e1); hipEventSynchronize(e1); \
    hipEventElapsedTime(&ms, e0, e1); printf("madd9  waves=%d %8.3f ms  %7.2f G madd/s\n", W, ms, (double)TM * 400 / ms / 1e6); \
    hipEventRecord(e0); hipLaunchKernelGGL(k_madd32<W>, dim3(TM / 256), dim3(256), 0, 0, o32, pts, 400); hipEventRecord(e1); hipEventSynchronize(e1); \
    hipEventElapsedTime(&ms, e0, e1); printf("madd32 waves=%d %8.3f ms  %7.2f G madd/s\n", W, ms, (double)TM * 400 / ms / 1e6);
    for (int rep = 0; rep < 2; rep++) { BM(2) BM(3) BM(4) }
    return 0;
}
