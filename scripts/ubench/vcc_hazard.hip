// Does gfx950 interlock "VALU writes VCC -> next VALU reads VCC as carry-in" in hardware?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k(const uint32_t *x, uint32_t *out_asm, uint32_t *out_ref, uint64_t *out64_asm, uint64_t *out64_ref, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = 0, ref = 0;
    uint64_t acc = 0, racc = 0; uint32_t hi = 0, rhi = 0;
    for (int j = 0; j < n; j++) {
        uint32_t a = x[(i * 7 + j) & 0xffff], b = x[(i * 13 + 3 * j + 1) & 0xffff];
        // 1-pass writer immediately followed by the carry consumer
        asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(a), "v"(b) : "vcc");
        ref += a < b;
        // multi-pass writer (the multiplier's case)
        asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        uint64_t p = (uint64_t)a * b; racc += p; rhi += racc < p;
    }
    out_asm[i] = cnt; out_ref[i] = ref; out64_asm[i] = acc + ((uint64_t)hi << 40); out64_ref[i] = racc + ((uint64_t)rhi << 40);
}
int main() {
    const int T = 256 * 1024, N = 4000;
    std::vector<uint32_t> h(65536);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i << 17) ^ 0xdeadbeef;
    uint32_t *x, *oa, *orf; uint64_t *o6a, *o6r;
    hipMalloc(&x, h.size() * 4); hipMalloc(&oa, T * 4); hipMalloc(&orf, T * 4); hipMalloc(&o6a, T * 8); hipMalloc(&o6r, T * 8);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(T / 256), dim3(256), 0, 0, x, oa, orf, o6a, o6r, N);
    std::vector<uint32_t> a(T), r(T); std::vector<uint64_t> a6(T), r6(T);
    hipMemcpy(a.data(), oa, T * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), orf, T * 4, hipMemcpyDeviceToHost);
    hipMemcpy(a6.data(), o6a, T * 8, hipMemcpyDeviceToHost); hipMemcpy(r6.data(), o6r, T * 8, hipMemcpyDeviceToHost);
    size_t bad = 0, bad6 = 0;
    for (int i = 0; i < T; i++) { bad += a[i] != r[i]; bad6 += a6[i] != r6[i]; }
    printf("v_cmp->v_addc back-to-back mismatches: %zu / %d ; v_mad_u64_u32->v_addc mismatches: %zu / %d  (%d carry events per thread)\n", bad, T, bad6, T, N);
    return 0;
}
