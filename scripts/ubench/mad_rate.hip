// Raw issue-rate probes for the integer instructions the field multiplication is made of (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t a0, uint32_t b0, int iters) {
    uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
    uint64_t c0 = 1, c1 = 2, c2 = 3, c3 = 4, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
    uint32_t d0 = 1, d1 = 2, d2 = 3, d3 = 4, d4 = 5, d5 = 6, d6 = 7, d7 = 8;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // 8 independent mad_u64_u32 chains
#define M(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
            M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6) M(c7)
        } else if (MODE == 1) { // v_mul_lo_u32
#define L(d) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(d) : "v"(a));
            L(d0) L(d1) L(d2) L(d3) L(d4) L(d5) L(d6) L(d7)
        } else if (MODE == 2) { // v_mul_hi_u32
#define H(d) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d) : "v"(a));
            H(d0) H(d1) H(d2) H(d3) H(d4) H(d5) H(d6) H(d7)
        } else if (MODE == 3) { // v_add_co / full-rate reference: v_add_u32
#define A(d) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(a));
            A(d0) A(d1) A(d2) A(d3) A(d4) A(d5) A(d6) A(d7)
        } else if (MODE == 4) { // v_lshl_add_u64 (64-bit add)
#define Q(c) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c) : "v"(c7));
            Q(c0) Q(c1) Q(c2) Q(c3) Q(c4) Q(c5) Q(c6) Q(c0)
        } else if (MODE == 5) { // v_mad_u32_u24
#define T(d) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(d) : "v"(a));
            T(d0) T(d1) T(d2) T(d3) T(d4) T(d5) T(d6) T(d7)
        } else if (MODE == 6) { // v_addc_co_u32 chain
#define C(d) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(d) : "v"(a) : "vcc");
            C(d0) C(d1) C(d2) C(d3) C(d4) C(d5) C(d6) C(d7)
        } else if (MODE == 7) { // v_mov_b32
#define V(d, e) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(e));
            V(d0, d1) V(d1, d2) V(d2, d3) V(d3, d4) V(d4, d5) V(d5, d6) V(d6, d7) V(d7, a)
        } else if (MODE == 8) { // v_mad_i64_i32
#define MI(c) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
            MI(c0) MI(c1) MI(c2) MI(c3) MI(c4) MI(c5) MI(c6) MI(c7)
        } else if (MODE == 9) { // v_alignbit_b32
#define AB(d, e) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(d) : "v"(e));
            AB(d0, d1) AB(d1, d2) AB(d2, d3) AB(d3, d4) AB(d4, d5) AB(d5, d6) AB(d6, d7) AB(d7, a)
        } else if (MODE == 10) { // v_ashrrev_i32
#define AS(d) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(d));
            AS(d0) AS(d1) AS(d2) AS(d3) AS(d4) AS(d5) AS(d6) AS(d7)
        } else if (MODE == 11) { // v_lshrrev_b64
#define LS(c) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(c));
            LS(c0) LS(c1) LS(c2) LS(c3) LS(c4) LS(c5) LS(c6) LS(c7)
        } else if (MODE == 12) { // v_and_b32 with literal
#define AN(d) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(d));
            AN(d0) AN(d1) AN(d2) AN(d3) AN(d4) AN(d5) AN(d6) AN(d7)
        } else if (MODE == 13) { // ONE dependent mad_u64_u32 chain (latency probe)
            M(c0) M(c0) M(c0) M(c0) M(c0) M(c0) M(c0) M(c0)
        } else if (MODE == 14) { // mad with an SGPR multiplicand
#define MS(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "s"(b0) : "vcc");
            MS(c0) MS(c1) MS(c2) MS(c3) MS(c4) MS(c5) MS(c6) MS(c7)
        } else if (MODE == 15) { // v_sub_u32
#define SB(d) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(d) : "v"(a));
            SB(d0) SB(d1) SB(d2) SB(d3) SB(d4) SB(d5) SB(d6) SB(d7)
        } else if (MODE == 16) { // v_mul_u32_u24
#define MU(d) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(d) : "v"(a));
            MU(d0) MU(d1) MU(d2) MU(d3) MU(d4) MU(d5) MU(d6) MU(d7)
        } else if (MODE == 17) { // v_bfe_u32
#define BF(d) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(d));
            BF(d0) BF(d1) BF(d2) BF(d3) BF(d4) BF(d5) BF(d6) BF(d7)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}
template <int MODE>
int run(const char *name, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves = 1 per SIMD) x waves_per_simd
    uint64_t *out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)blocks * 4 * iters * 8;           // wave-instructions
    double per_simd_cycles = ms * 1e-3 * 2.4e9 / (inst / 1024.0);
    printf("%-18s waves/SIMD=%d  %8.3f ms  %7.2f T lane-op/s  ~%.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, inst * 64 / ms / 1e9, per_simd_cycles);
    hipFree(out);
    return 0;
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mad_u64_u32", w); run<1>("v_mul_lo_u32", w); run<2>("v_mul_hi_u32", w); run<3>("v_add_u32", w);
        run<4>("v_lshl_add_u64", w); run<5>("v_mad_u32_u24", w); run<6>("v_addc_co_u32", w); run<7>("v_mov_b32", w);
        run<8>("v_mad_i64_i32", w); run<9>("v_alignbit_b32", w); run<10>("v_ashrrev_i32", w); run<11>("v_lshrrev_b64", w);
        run<12>("v_and_b32 lit", w); run<13>("mad_u64 dep chain", w); run<14>("mad_u64 sgpr", w); run<15>("v_sub_u32", w);
        run<16>("v_mul_u32_u24", w); run<17>("v_bfe_u32", w);
    }
    return 0;
}
