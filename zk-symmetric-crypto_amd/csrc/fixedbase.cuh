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

}  // namespace zk
