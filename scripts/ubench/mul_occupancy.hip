// Field products per second at LOW occupancy (the transforms' LDS tiles leave 2 wavefronts per SIMD): 8 x 32 weak multiplier
// against the 9-limb product (one accumulator chain) and a 9-limb product with two interleaved accumulator chains per column.
#include "../../zk-symmetric-crypto_amd/csrc/field9.cuh"
#include "mul29_asm.inc"
#include <stdio.h>
#include <vector>
using namespace zk;
using F9 = Fr9;

// two chains: the a*b products of a column go to accA, the m*M products to accB; the column value is their sum
__device__ __forceinline__ F9 mul_two_chains(const F9 &a, const F9 &b) {
    int64_t accA = 0, accB = 0;
    int32_t m[9];
    F9 r;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            F9::mac(accA, a.l[i], b.l[k - i]);
            const int mi = (k < 9) ? (i < k ? i : -1) : i;      // m index paired with this product slot
            if (mi >= 0) F9::mac_const(accB, m[mi], Fr9Params::M[k - mi]);
        }
        int64_t acc = accA + accB;
        if (k < 9) {
            m[k] = (int32_t)(((uint32_t)acc * Fr9Params::PINV) & F9::MASK);
            F9::mac_const(acc, m[k], Fr9Params::M[0]);
        } else r.l[k - 9] = (int32_t)((uint32_t)acc & F9::MASK);
        accA = acc >> 29;
        accB = 0;
    }
    r.l[8] = (int32_t)accA;
    return r;
}
// the same product as ONE asm block (gen_mul29.py: unsigned limbs, v_lshrrev_b64 shifts): no compiler-inserted s_nop between the multiply-adds
__device__ __forceinline__ F9 mul_asm_block(const F9 &a, const F9 &b) {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7, t8;
    asm volatile(ZK_MUL29_ASM_FP
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8)
                 : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]),
                   "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(b.l[8])
                 : "v16", "v17", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "vcc");
    F9 r;
    r.l[0] = t0; r.l[1] = t1; r.l[2] = t2; r.l[3] = t3; r.l[4] = t4; r.l[5] = t5; r.l[6] = t6; r.l[7] = t7; r.l[8] = t8;
    return r;
}
template <int V>
__global__ __launch_bounds__(256) void k(const Fr *x, Fr *out, int iters) {
    extern __shared__ uint32_t pad[];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (V == 0) {
        Fr a = x[i], b = x[i + 1];
        for (int k = 0; k < iters; k++) { a = Fr::mul_w(a, b); b = Fr::mul_w(b, a); }
        out[i] = Fr::add_w(a, b).canon();
    } else {
        F9 a = F9::from_words(x[i]), b = F9::from_words(x[i + 1]);
        for (int k = 0; k < iters; k++) {
            if (V == 1) { a = F9::mul(a, b); b = F9::mul(b, a); }
            else if (V == 2) { a = mul_two_chains(a, b); b = mul_two_chains(b, a); }
            else { a = mul_asm_block(a, b); b = mul_asm_block(b, a); }
        }
        out[i] = F9::to_words(F9::norm(F9::sub(a, b)));
    }
    if (iters < 0) out[i].l[0] = pad[threadIdx.x];
}
int main() {
    size_t T = 256 * 256 * 16;
    std::vector<Fr> h(T + 1), r1(T), r2(T);
    for (size_t i = 0; i <= T; i++) for (int k = 0; k < 8; k++) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u + 12345u * k * k) & (k == 7 ? 0x0fffffffu : 0xffffffffu);
    Fr *d, *o; hipMalloc(&d, (T + 1) * 32); hipMalloc(&o, T * 32);
    hipMemcpy(d, h.data(), (T + 1) * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<1>, dim3(T / 256), dim3(256), 0, 0, d, o, 9); hipMemcpy(r1.data(), o, T * 32, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<2>, dim3(T / 256), dim3(256), 0, 0, d, o, 9); hipMemcpy(r2.data(), o, T * 32, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < T; i++) bad += !(r1[i] == r2[i]);
    printf("two-chain product vs mul: %zu mismatches\n", bad);
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    const int lds_for[4] = {150 * 1024, 72 * 1024, 36 * 1024, 0};     // 1, 2, 4, 8 workgroups of 4 wavefronts per CU
    const char *names[4] = {"8x32 weak mul_w", "9-limb one chain", "9-limb two chains", "9-limb ONE asm block (Fp)"};
    for (int li = 0; li < 4; li++)
        for (int v = 0; v < 4; v++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(T / 256), dim3(256), lds_for[li], 0, d, o, 500);
                if (v == 1) hipLaunchKernelGGL(k<1>, dim3(T / 256), dim3(256), lds_for[li], 0, d, o, 500);
                if (v == 2) hipLaunchKernelGGL(k<2>, dim3(T / 256), dim3(256), lds_for[li], 0, d, o, 500);
                if (v == 3) hipLaunchKernelGGL(k<3>, dim3(T / 256), dim3(256), lds_for[li], 0, d, o, 500);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("waves/SIMD=%d  %-18s %8.3f ms  %7.2f G mul/s\n", 1 << li, names[v], ms, (double)T * 1000 / ms / 1e6);
        }
    return 0;
}
