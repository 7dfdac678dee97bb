// Host build of csrc/glv.cuh against the plain double-and-add of curve.cuh: same point for edge and random scalars and points,
// decomposition identity k1 + k2 lambda = k (mod r) with both parts below 2^127.
#include <stdio.h>
#include <stdint.h>
#include <random>
#include "glv.cuh"

using namespace zk;

static bool same_point(const XYZZ<Fp> &a, const XYZZ<Fp> &b) {
    const Affine<Fp> x = a.to_affine(), y = b.to_affine();
    if (x.is_inf() || y.is_inf()) return x.is_inf() && y.is_inf();
    return x.x == y.x && x.y == y.y;
}

int main() {
    std::mt19937_64 rng(20261003);
    const Affine<Fp> g{Fp::from_u32(1), Fp::from_u32(2)};
    // lambda acts as phi: lambda * G == (beta x, y)
    const uint32_t LAMBDA[8] = {0xb99c90ddu, 0x8b17ea66u, 0x8d8daaa7u, 0x5bfc4108u, 0x41a91758u, 0xb3c4d79du, 0u, 0u};
    {
        const Affine<Fp> lg = scalar_mul(g, LAMBDA, 256).to_affine();
        const uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
        uint32_t k1[4], k2[4];
        bool n1, n2;
        if (!glv::decompose(LAMBDA, k1, n1, k2, n2)) { printf("decompose(lambda) failed\n"); return 1; }
        if (!(lg.y == g.y) || lg.x == g.x) { printf("lambda is not the eigenvalue of phi\n"); return 1; }
        if (!same_point(scalar_mul_glv(g, one), XYZZ<Fp>::from_affine(g))) { printf("1 * G\n"); return 1; }
    }
    const uint32_t R[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    int checked = 0;
    for (int it = 0; it < 260; it++) {
        uint32_t k[8];
        for (int i = 0; i < 8; i++) k[i] = (uint32_t)rng();
        if (it < 200) k[7] &= 0x3fffffffu;                                  // < 2^254 like a canonical scalar; the rest: any 256-bit value
        if (it == 0) for (int i = 0; i < 8; i++) k[i] = 0;
        if (it == 1) { for (int i = 0; i < 8; i++) k[i] = 0; k[0] = 1; }
        if (it == 2) { for (int i = 0; i < 8; i++) k[i] = R[i]; k[0] -= 1; }    // r - 1
        if (it == 3) for (int i = 0; i < 8; i++) k[i] = R[i];                 // r: the point at infinity
        if (it == 4) for (int i = 0; i < 8; i++) k[i] = 0xffffffffu;
        if (it == 5) for (int i = 0; i < 8; i++) k[i] = LAMBDA[i];
        if (it == 6) { for (int i = 0; i < 8; i++) k[i] = 0; k[4] = 1; }        // 2^128
        uint32_t k1[4], k2[4];
        bool n1, n2;
        if (!glv::decompose(k, k1, n1, k2, n2)) { printf("decompose out of range at %d\n", it); return 1; }
        // the base point: a random multiple of G (cheap: 64-bit scalar)
        uint32_t m[8] = {(uint32_t)rng(), (uint32_t)rng(), 0, 0, 0, 0, 0, 0};
        const Affine<Fp> p = it == 7 ? Affine<Fp>::inf() : scalar_mul(g, m, 64).to_affine();
        if (!same_point(scalar_mul_glv(p, k), scalar_mul(p, k, 256))) { printf("mismatch at %d\n", it); return 1; }
        checked++;
    }
    printf("%d scalar multiplications\nok\n", checked);
    return 0;
}
