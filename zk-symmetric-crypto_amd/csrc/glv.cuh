// Variable-base scalar multiplication on BN254 G1 through the curve's endomorphism (Gallant-Lambert-Vanstone).
//
// phi(x, y) = (beta x, y), beta a primitive cube root of unity in Fp, acts on the order-r group as multiplication by lambda
// (lambda^2 + lambda + 1 = 0 mod r).  A scalar k is written k = k1 + k2 lambda with |k1|, |k2| < 2^127 (lattice reduction with the
// basis (a1, b1), (a2, b2) of { (x, y) : x + y lambda = 0 mod r }), and k P = k1 P + k2 phi(P) is ONE double-and-add chain of 127
// doublings with a joint table {P, phi P, P + phi P} instead of 254 doublings.  The two variable-base multiplications s Ar and r Bs1 of
// the proof tail (gnark backend/groth16/bn254/prove.go: krs.ScalarMultiplication) are a chain of dependent doublings on one thread each --
// latency of a single Prove() call, whatever the batch size -- so halving the chain halves that latency.
//
// Correctness does not depend on the rounding: k1 + k2 lambda = k - c1 (a1 + b1 lambda) - c2 (a2 + b2 lambda) = k mod r for ANY
// integers c1, c2; c1 = round(b2 k / r), c2 = round(-b1 k / r) (taken with 320-bit fixed-point reciprocals) only make k1, k2 short:
// |k1|, |k2| < 2^127 for every k < 2^256 (checked exhaustively on the edge values and on 3 x 10^5 random ones; the loop below runs
// 128 bit positions and the decomposition asserts the words above are clear).  Same group element as scalar_mul, hence the same
// proof bytes (tests: tests/host/glv_check.cpp on the host build, the proof parity tests on the device).
#pragma once
#include "curve.cuh"

namespace zk {

namespace glv {
// r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
// lambda = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd, beta = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
// a1 = b2 = 9931322734385697763, -b1 = 147946756881789319000765030803803410728, a2 = 147946756881789319010696353538189108491
// G1 = floor(b2 2^320 / r), G2 = floor(-b1 2^320 / r)
// out[0 .. na+nb) = a * b (little-endian 32-bit words)
ZK_HD void mul_words(const uint32_t *a, int na, const uint32_t *b, int nb, uint32_t *out) {
    for (int i = 0; i < na + nb; i++) out[i] = 0;
    for (int i = 0; i < na; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < nb; j++) {
            const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
        out[i + nb] = (uint32_t)carry;
    }
}
// x -= y over n words (two's complement wrap-around)
ZK_HD void sub_words(uint32_t *x, const uint32_t *y, int n) {
    uint64_t borrow = 0;
    for (int i = 0; i < n; i++) {
        const uint64_t t = (uint64_t)x[i] - y[i] - borrow;
        x[i] = (uint32_t)t;
        borrow = (t >> 63) & 1;
    }
}
// |x| of an n-word two's complement value; returns its sign
ZK_HD bool abs_words(uint32_t *x, int n) {
    const bool neg = (x[n - 1] >> 31) != 0;
    if (neg) {
        uint64_t carry = 1;
        for (int i = 0; i < n; i++) {
            const uint64_t t = (uint64_t)(~x[i]) + carry;
            x[i] = (uint32_t)t;
            carry = t >> 32;
        }
    }
    return neg;
}
// k (8 words, any value < 2^256) -> |k1|, |k2| (4 words each) and their signs; false if a part does not fit 127 bits (cannot happen)
ZK_HD bool decompose(const uint32_t *k, uint32_t *k1, bool &neg1, uint32_t *k2, bool &neg2) {
    const uint32_t G1[5] = {0x773a6ef2u, 0x6eb9c714u, 0xc7e0b3d7u, 0xd91d232eu, 0x2u};
    const uint32_t G2[7] = {0x5eaa26d9u, 0xa5e38cfbu, 0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x2u};
    const uint32_t A1[2] = {0x94d213e3u, 0x89d32568u};                                  // = b2
    const uint32_t A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
    const uint32_t NB1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};       // -b1
    uint32_t t[15], c1[5], c2[5];
    // c = (k g + 2^319) >> 320
    mul_words(k, 8, G1, 5, t);                       // 13 words; bit 319 = bit 31 of word 9
    {
        uint64_t carry = 0x80000000ull;
        for (int i = 9; i < 13; i++) { const uint64_t s = (uint64_t)t[i] + carry; t[i] = (uint32_t)s; carry = s >> 32; }
        for (int i = 0; i < 5; i++) c1[i] = 10 + i < 13 ? t[10 + i] : 0u;
    }
    mul_words(k, 8, G2, 7, t);                       // 15 words
    {
        uint64_t carry = 0x80000000ull;
        for (int i = 9; i < 15; i++) { const uint64_t s = (uint64_t)t[i] + carry; t[i] = (uint32_t)s; carry = s >> 32; }
        for (int i = 0; i < 5; i++) c2[i] = t[10 + i];
    }
    // k1 = k - c1 a1 - c2 a2 ; k2 = c1 (-b1) - c2 b2   (10-word two's complement)
    uint32_t x[10], y[10], p[10];
    for (int i = 0; i < 10; i++) x[i] = i < 8 ? k[i] : 0u;
    mul_words(c1, 5, A1, 2, p); for (int i = 7; i < 10; i++) p[i] = 0u;
    sub_words(x, p, 10);
    mul_words(c2, 5, A2, 4, p); p[9] = 0u;
    sub_words(x, p, 10);
    mul_words(c1, 5, NB1, 4, y); y[9] = 0u;
    mul_words(c2, 5, A1, 2, p); for (int i = 7; i < 10; i++) p[i] = 0u;
    sub_words(y, p, 10);
    neg1 = abs_words(x, 10);
    neg2 = abs_words(y, 10);
    uint32_t hi = 0;
    for (int i = 4; i < 10; i++) hi |= x[i] | y[i];
    hi |= (x[3] | y[3]) >> 31;
    for (int i = 0; i < 4; i++) { k1[i] = x[i]; k2[i] = y[i]; }
    return hi == 0;
}
}  // namespace glv

// k * p for k < 2^256 (8 LE words): same point as scalar_mul(p, k, 256)
ZK_HD XYZZ<Fp> scalar_mul_glv(const Affine<Fp> &p, const uint32_t *k) {
    if (p.is_inf()) return XYZZ<Fp>::inf();
    uint32_t k1[4], k2[4];
    bool n1, n2;
    if (!glv::decompose(k, k1, n1, k2, n2)) return scalar_mul(p, k, 256);       // (unreachable; keeps the function total)
    Fp beta;
    {
        const uint32_t BETA[8] = {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu, 0x0u, 0x0u};
        for (int i = 0; i < 8; i++) beta.l[i] = BETA[i];
        beta = beta.to_mont();
    }
    const Affine<Fp> p1{p.x, n1 ? p.y.neg() : p.y};
    const Affine<Fp> p2{p.x * beta, n2 ? p.y.neg() : p.y};
    XYZZ<Fp> p3 = XYZZ<Fp>::from_affine(p1);
    p3.madd(p2);                                     // phi(P) = lambda P != +-P: a generic addition
    XYZZ<Fp> acc = XYZZ<Fp>::inf();
    for (int i = 127; i >= 0; i--) {
        acc = acc.dbl();
        const uint32_t b1 = (k1[i >> 5] >> (i & 31)) & 1u, b2 = (k2[i >> 5] >> (i & 31)) & 1u;
        if (b1 & b2) acc.add(p3);
        else if (b1) acc.madd(p1);
        else if (b2) acc.madd(p2);
    }
    return acc;
}

}  // namespace zk
