// Does the VGPR bank of the operands change the issue rate of v_mad_u64_u32 / v_addc_co_u32 on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define BODY8(s) s s s s s s s s
#define PRE "v_mov_b32 v16, %1\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, %1\n\tv_mov_b32 v19, 1\n\t" \
    "v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\t" \
    "v_mov_b32 v24, %1\n\tv_mov_b32 v25, %1\n\tv_mov_b32 v26, %1\n\tv_mov_b32 v27, %1\n\t" \
    "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t" \
    "s_mov_b32 s20, %2\n\t" "1:\n\t"
#define POST "s_sub_u32 s20, s20, 1\n\t" "s_cmp_lg_u32 s20, 0\n\t" "s_cbranch_scc1 1b\n\t" \
    "v_xor_b32 %0, v16, v17\n\tv_xor_b32 %0, %0, v18\n\tv_xor_b32 %0, %0, v19\n\tv_xor_b32 %0, %0, v28\n\tv_xor_b32 %0, %0, v29\n\tv_xor_b32 %0, %0, v30\n\tv_xor_b32 %0, %0, v31\n\t"
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","s20","s22","s23","vcc","scc"
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t a0, int iters) {
    uint32_t r = 0;
    uint32_t av = a0 + threadIdx.x;
    // the loop body, 32 instructions of the probed kind
    if (MODE == 0) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], vcc, v22, v23, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v22, v23, v[18:19]\n\t" "v_mad_u64_u32 v[16:17], vcc, v26, v27, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v26, v27, v[18:19]\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    if (MODE == 1) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], vcc, v20, v24, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v20, v24, v[18:19]\n\t" "v_mad_u64_u32 v[16:17], vcc, v20, v24, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v20, v24, v[18:19]\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    if (MODE == 2) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], vcc, v22, v26, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v22, v26, v[18:19]\n\t" "v_mad_u64_u32 v[16:17], vcc, v22, v26, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v22, v26, v[18:19]\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    if (MODE == 3) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], vcc, v21, v22, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v21, v22, v[18:19]\n\t" "v_mad_u64_u32 v[16:17], vcc, v25, v26, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v25, v26, v[18:19]\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    if (MODE == 4) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], vcc, v22, s20, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v22, s20, v[18:19]\n\t" "v_mad_u64_u32 v[16:17], vcc, v26, s20, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], vcc, v26, s20, v[18:19]\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // the multiplier's pair: mad into P, addc into the carry word; distinct banks / conflicting banks
    if (MODE == 5) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], vcc, v22, v23, v[16:17]\n\t" "v_addc_co_u32_e32 v19, vcc, 0, v19, vcc\n\t" "v_mad_u64_u32 v[16:17], vcc, v26, v27, v[16:17]\n\t" "v_addc_co_u32_e32 v19, vcc, 0, v19, vcc\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    if (MODE == 6) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], vcc, v20, v24, v[16:17]\n\t" "v_addc_co_u32_e32 v19, vcc, 0, v19, vcc\n\t" "v_mad_u64_u32 v[16:17], vcc, v21, v25, v[16:17]\n\t" "v_addc_co_u32_e32 v19, vcc, 0, v19, vcc\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // mad with the accumulator as a separate destination (no read-after-write on the same pair)
    if (MODE == 7) asm volatile(PRE BODY8("v_mad_u64_u32 v[28:29], vcc, v22, v23, v[16:17]\n\t" "v_mad_u64_u32 v[30:31], vcc, v22, v23, v[18:19]\n\t" "v_mad_u64_u32 v[28:29], vcc, v26, v27, v[16:17]\n\t" "v_mad_u64_u32 v[30:31], vcc, v26, v27, v[18:19]\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // mad with inline-constant addend (2 VGPR reads only)
    if (MODE == 8) asm volatile(PRE BODY8("v_mad_u64_u32 v[28:29], vcc, v22, v23, 0\n\t" "v_mad_u64_u32 v[30:31], vcc, v22, v23, 0\n\t" "v_mad_u64_u32 v[28:29], vcc, v26, v27, 0\n\t" "v_mad_u64_u32 v[30:31], vcc, v26, v27, 0\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // sdst = SGPR pair instead of vcc
    if (MODE == 9) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], s[22:23], v22, v23, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], s[22:23], v22, v23, v[18:19]\n\t" "v_mad_u64_u32 v[16:17], s[22:23], v26, v27, v[16:17]\n\t" "v_mad_u64_u32 v[18:19], s[22:23], v26, v27, v[18:19]\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // v_mul_lo_u32 + v_mul_hi_u32 (32-bit results) instead
    if (MODE == 10) asm volatile(PRE BODY8("v_mul_lo_u32 v28, v22, v23\n\t" "v_mul_hi_u32 v29, v22, v23\n\t" "v_mul_lo_u32 v30, v26, v27\n\t" "v_mul_hi_u32 v31, v26, v27\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // v_mad_u32_u24 (24-bit multiplier path)
    if (MODE == 11) asm volatile(PRE BODY8("v_mad_u32_u24 v28, v22, v23, v28\n\t" "v_mad_u32_u24 v29, v22, v23, v29\n\t" "v_mad_u32_u24 v30, v26, v27, v30\n\t" "v_mad_u32_u24 v31, v26, v27, v31\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // simple adds for calibration
    if (MODE == 12) asm volatile(PRE BODY8("v_add_u32 v28, v22, v28\n\t" "v_add_u32 v29, v23, v29\n\t" "v_add_u32 v30, v26, v30\n\t" "v_add_u32 v31, v27, v31\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // carry chain alone
    if (MODE == 13) asm volatile(PRE BODY8("v_addc_co_u32_e32 v28, vcc, v22, v28, vcc\n\t" "v_addc_co_u32_e32 v29, vcc, v23, v29, vcc\n\t" "v_addc_co_u32_e32 v30, vcc, v26, v30, vcc\n\t" "v_addc_co_u32_e32 v31, vcc, v27, v31, vcc\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // v_mad_u64_u32 interleaved with independent cheap VOP2 (does the mad leave issue slots for a simple op?)
    if (MODE == 14) asm volatile(PRE BODY8("v_mad_u64_u32 v[16:17], vcc, v22, v23, v[16:17]\n\t" "v_add_u32 v28, v26, v28\n\t" "v_mad_u64_u32 v[18:19], vcc, v22, v23, v[18:19]\n\t" "v_add_u32 v29, v27, v29\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    // v_mul_u32_u24 + v_mul_hi_u32_u24
    if (MODE == 15) asm volatile(PRE BODY8("v_mul_u32_u24 v28, v22, v23\n\t" "v_mul_hi_u32_u24 v29, v22, v23\n\t" "v_mul_u32_u24 v30, v26, v27\n\t" "v_mul_hi_u32_u24 v31, v26, v27\n\t") POST : "=&v"(r) : "v"(av), "s"(iters) : CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
void run(const char *name, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;
    uint32_t *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 3u, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 3u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)blocks * 4 * iters * 32;
    printf("%-28s w/SIMD=%d %8.3f ms  %.2f cycles per wave-instr per SIMD (2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (inst / 1024.0));
    hipFree(out);
}
int main() {
    for (int w : {2, 3, 4, 8}) {
        run<0>("mad distinct banks", w); run<1>("mad a,b bank0 (=acc.lo)", w); run<2>("mad a,b both bank2", w); run<3>("mad a bank1(=acc.hi) b bank2", w);
        run<4>("mad vgpr x sgpr", w); run<5>("mad+addc distinct", w); run<6>("mad+addc conflicting", w); run<7>("mad separate dst", w);
        run<8>("mad addend 0", w); run<9>("mad sdst sgpr", w); run<10>("mul_lo + mul_hi", w); run<11>("mad_u32_u24", w); run<12>("add_u32", w); run<13>("addc chain", w);
        run<14>("mad + add interleaved", w); run<15>("mul_u32_u24 + mul_hi_u32_u24", w);
    }
    return 0;
}
