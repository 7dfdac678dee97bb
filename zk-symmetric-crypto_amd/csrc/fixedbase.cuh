// Fixed-base scalar multiplication with 8-bit windows: tab[w*255 + d-1] = d * 2^(8w) * base (32 windows).
// Used for r*delta, s*delta, -rs*delta in the proof tail and by the key generator (setup.hip).
#pragma once
#include "kernels.h"

namespace zk {

template <class F>
__global__ __launch_bounds__(64) void k_fixed_table(const Affine<F> *__restrict__ base, Affine<F> *__restrict__ tab) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 32 * 255) return;
    const uint32_t w = t / 255, d = t % 255 + 1;
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    k[w >> 2] = d << ((w & 3) * 8);
    tab[t] = scalar_mul(*base, k, 256).to_affine();
}

template <class F>
__device__ __forceinline__ XYZZ<F> fixed_mul(const Affine<F> *__restrict__ tab, const uint32_t *k) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = 0; w < 32; w++) {
        const uint32_t d = (k[w >> 2] >> ((w & 3) * 8)) & 0xff;
        if (d) acc.madd(tab[w * 255 + d - 1]);
    }
    return acc;
}

// The same product by 16 lanes of one workgroup (small batches: the 32 dependent mixed additions above are latency of every Prove()
// call): lane i takes windows i and i + 16, four rounds through LDS join the 16 partial sums.  Result valid in lane 0; every lane of
// the workgroup must call (barriers).  Same group element as fixed_mul.
template <class F>
__device__ __forceinline__ XYZZ<F> fixed_mul_tree(const Affine<F> *__restrict__ tab, const uint32_t *k, int lane, XYZZ<F> *lds) {
    XYZZ<F> acc = XYZZ<F>::inf();
    if (lane < 16) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int w = lane + 16 * h;
            const uint32_t d = (k[w >> 2] >> ((w & 3) * 8)) & 0xff;
            if (d) acc.madd(tab[w * 255 + d - 1]);
        }
    }
    for (int step = 8; step >= 1; step >>= 1) {
        if (lane >= step && lane < 2 * step) lds[lane] = acc;
        __syncthreads();
        if (lane < step) acc.add(lds[lane + step]);
        __syncthreads();
    }
    return acc;
}

}  // namespace zk
