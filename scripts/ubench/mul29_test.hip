#include "../../zk-symmetric-crypto_amd/csrc/field.cuh"
#include "mul29_asm.inc"
#include <stdio.h>
#include <vector>
using namespace zk;
struct F29 { uint32_t l[9]; };
__device__ __forceinline__ F29 to29(const Fp &x) {      // plain re-limbing of the 256-bit integer
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t v = x.l[w];
        if (w + 1 < 8) v |= (uint64_t)x.l[w + 1] << 32;
        r.l[i] = (uint32_t)(v >> sh) & 0x1fffffffu;
    }
    return r;
}
__device__ __forceinline__ Fp from29(const F29 &x) {     // value must be < 2^256
    uint32_t out[9] = {0};
    uint64_t acc = 0; int have = 0, o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc |= (uint64_t)x.l[i] << have; have += 29;
        if (have >= 32) { out[o++] = (uint32_t)acc; acc >>= 32; have -= 32; }
    }
    out[o] = (uint32_t)acc;
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = out[i];
    return r;
}
__device__ __attribute__((noinline)) F29 mul29(F29 a, F29 b) {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7, t8;
    asm volatile(ZK_MUL29_ASM_FP
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8)
                 : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]),
                   "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(b.l[8])
                 : "v16", "v17", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "vcc");
    F29 r; r.l[0]=t0; r.l[1]=t1; r.l[2]=t2; r.l[3]=t3; r.l[4]=t4; r.l[5]=t5; r.l[6]=t6; r.l[7]=t7; r.l[8]=t8;
    return r;
}
__global__ __launch_bounds__(256) void k_check(const Fp *x, Fp *got, Fp *exp, Fp c251) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = x[i], b = x[i + 1];
    F29 r = mul29(to29(a), to29(b));            // = a*b*2^-261 mod p, value < 2p
    Fp g = from29(r);
    Fp::reduce_once(g.l);
    got[i] = g;
    exp[i] = (a * b) * c251;                    // a*b*2^-256 * 2^251 * 2^-256 = a*b*2^-261
}
__global__ __launch_bounds__(256) void k_bench29(const Fp *x, Fp *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    F29 a = to29(x[i]), b = to29(x[i + 1]);
    for (int k = 0; k < iters; k++) { a = mul29(a, b); b = mul29(b, a); }
    F29 s; for (int k = 0; k < 9; k++) s.l[k] = (a.l[k] + b.l[k]) & 0x1fffffff;
    out[i] = from29(s);
}
__global__ __launch_bounds__(256) void k_bench32(Fp *x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = x[i], b = x[i + 1];
    for (int k = 0; k < iters; k++) { a = a * b; b = b * a; }
    x[i] = a + b;
}
int main() {
    size_t T = 256 * 256 * 16;
    std::vector<Fp> h(T + 1), g(T), e(T);
    for (size_t i = 0; i <= T; i++) for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u + 12345u * k * k) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
    Fp *d, *dg, *de; hipMalloc(&d, (T + 1) * 32); hipMalloc(&dg, T * 32); hipMalloc(&de, T * 32);
    hipMemcpy(d, h.data(), (T + 1) * 32, hipMemcpyHostToDevice);
    // 2^251 mod p in Montgomery form (R=2^256): computed on host with python-generated constant
    Fp c251;
    const uint32_t C[8] = C251_LIMBS;
    for (int k = 0; k < 8; k++) c251.l[k] = C[k];
    hipLaunchKernelGGL(k_check, dim3(T / 256), dim3(256), 0, 0, d, dg, de, c251);
    hipMemcpy(g.data(), dg, T * 32, hipMemcpyDeviceToHost); hipMemcpy(e.data(), de, T * 32, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < T; i++) bad += !(g[i] == e[i]);
    printf("mul29 mismatches: %zu / %zu\n", bad, T);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_bench29, dim3(T / 256), dim3(256), 0, 0, d, dg, 1000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("mul29  %8.3f ms  %7.2f G mul/s\n", ms, (double)T * 2000 / ms / 1e6);
        hipEventRecord(e0); hipLaunchKernelGGL(k_bench32, dim3(T / 256), dim3(256), 0, 0, d, 1000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("mul32  %8.3f ms  %7.2f G mul/s\n", ms, (double)T * 2000 / ms / 1e6);
    }
    return 0;
}
