// Field-multiplication variants for gfx950: correctness vs the production Field::mul_impl and throughput.
#include "../../zk-symmetric-crypto_amd/csrc/field.cuh"
#include <stdio.h>
#include <vector>
using namespace zk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---- V2: finely integrated product scanning (FIPS) Montgomery, 96-bit column accumulator:
//      v_mad_u64_u32 (carry-out -> vcc) + v_addc_co_u32 into the third word.
#define MAC(acc, hi, x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(x), "v"(y) : "vcc")
template <class P>
__device__ __attribute__((noinline)) Field<P> mul_fips(Field<P> a, Field<P> b) {
    uint32_t m[8], t[8];
    uint64_t acc = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) MAC(acc, hi, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) MAC(acc, hi, m[i], P::M[k - i]);
        m[k] = (uint32_t)acc * P::INV;
        MAC(acc, hi, m[k], P::M[0]);
        acc = (acc >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
        for (int i = k - 7; i < 8; i++) MAC(acc, hi, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 7; i < 8; i++) MAC(acc, hi, m[i], P::M[k - i]);
        t[k - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    Field<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    // value < 2M : one conditional subtraction (acc holds the 9th limb, zero for M < 2^254 operands < M)
    Field<P>::reduce_once(r.l);
    return r;
}

// ---- V1: same algorithm in plain C (let the compiler pick carries)
template <class P>
__device__ __attribute__((noinline)) Field<P> mul_fips_c(Field<P> a, Field<P> b) {
    uint32_t m[8], t[8];
    uint64_t acc = 0;
    uint32_t hi = 0;
#define MACC(x, y) { uint64_t p_ = (uint64_t)(x) * (y); acc += p_; hi += acc < p_; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) MACC(a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) MACC(m[i], P::M[k - i]);
        m[k] = (uint32_t)acc * P::INV;
        MACC(m[k], P::M[0]);
        acc = (acc >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
        for (int i = k - 7; i < 8; i++) MACC(a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 7; i < 8; i++) MACC(m[i], P::M[k - i]);
        t[k - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    Field<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    Field<P>::reduce_once(r.l);
    return r;
}

template <int V>
__device__ __forceinline__ Fp mulv(const Fp &a, const Fp &b) {
    if (V == 0) return a * b;
    if (V == 1) return mul_fips_c<FpParams>(a, b);
    return mul_fips<FpParams>(a, b);
}
template <int V>
__global__ __launch_bounds__(256) void k_bench(Fp *x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = x[i], b = x[i + 1];
    for (int k = 0; k < iters; k++) { a = mulv<V>(a, b); b = mulv<V>(b, a); }
    x[i] = a + b;
}
template <int V>
int run(const char *name, const std::vector<Fp> &h, std::vector<Fp> &out) {
    size_t T = h.size() - 1;
    Fp *d; CHECK(hipMalloc(&d, (T + 1) * sizeof(Fp)));
    CHECK(hipMemcpy(d, h.data(), (T + 1) * sizeof(Fp), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 1000;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_bench<V>, dim3(T / 256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    out.resize(T);
    CHECK(hipMemcpy(out.data(), d, T * sizeof(Fp), hipMemcpyDeviceToHost));
    printf("%-12s %8.3f ms  %7.2f G mul/s\n", name, ms, (double)T * iters * 2 / ms / 1e6);
    hipFree(d);
    return 0;
}
int main() {
    size_t T = 256 * 256 * 16;
    std::vector<Fp> h(T + 1), o0, o1, o2;
    for (size_t i = 0; i <= T; i++) for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u + 12345u * k * k) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
    run<0>("cios (prod)", h, o0); run<1>("fips C", h, o1); run<2>("fips asm", h, o2);
    run<0>("cios (prod)", h, o0); run<1>("fips C", h, o1); run<2>("fips asm", h, o2);
    size_t bad1 = 0, bad2 = 0;
    for (size_t i = 0; i < T; i++) { if (!(o0[i] == o1[i])) bad1++; if (!(o0[i] == o2[i])) bad2++; }
    printf("mismatches: fipsC %zu  fipsAsm %zu of %zu\n", bad1, bad2, T);
    return 0;
}
