// BN254 prime-field arithmetic for gfx950 (CDNA4): 8 x 32-bit limbs, Montgomery form R = 2^256.
//
// Replaces (on device) the fp/fr packages of github.com/consensys/gnark-crypto v0.19.2
// (gnark/go.mod:7) that the reference prover reaches through groth16.Prove
// (gnark/libraries/prover/impl/provers.go:179,269).  Byte layout of an element is identical to
// gnark-crypto's in-memory form (4 x u64 little-endian Montgomery limbs), so the R1CS coefficient
// table is uploaded verbatim.
//
// CDNA4 has no 64x64 multiplier; the hot instruction is v_mad_u64_u32 (32x32+64 -> 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field_mul_asm.inc"

#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__

namespace zk {

struct alignas(16) u256 {
    uint32_t l[8];
};

struct FpParams {
    static constexpr int ID = 0;
    // p = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    static constexpr uint32_t M[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xe4866389u;  // -p^-1 mod 2^32
    // R mod p, R^2 mod p
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
struct FrParams {
    static constexpr int ID = 1;
    // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    static constexpr uint32_t M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xefffffffu;
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class P>
struct Field {
    uint32_t l[8];

    static ZK_HD Field zero() {
        Field r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = 0;
        return r;
    }
    static ZK_HD Field one() {
        Field r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = P::ONE[i];
        return r;
    }
    static ZK_HD Field r2() {
        Field r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = P::R2[i];
        return r;
    }
    ZK_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i];
        return o == 0;
    }
    ZK_HD bool operator==(const Field &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    ZK_HD bool operator!=(const Field &b) const { return !(*this == b); }

    // t >= M ?
    static ZK_HD bool geq_mod(const uint32_t *t) {
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            if (t[i] > P::M[i]) return true;
            if (t[i] < P::M[i]) return false;
        }
        return true;
    }
    // conditional subtract of M (branch-free): r = t - M if t >= M
    static ZK_HD void reduce_once(uint32_t *t) {
        uint32_t d[8];
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)t[i] - P::M[i] - borrow;
            d[i] = (uint32_t)x;
            borrow = (uint32_t)(x >> 63);
        }
        // borrow == 0  <=> t >= M
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = borrow ? t[i] : d[i];
    }
    friend ZK_HD Field operator+(const Field &a, const Field &b) {
        Field r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)a.l[i] + b.l[i] + c;
            r.l[i] = (uint32_t)x;
            c = (uint32_t)(x >> 32);
        }
        // M < 2^254 so a+b < 2^255: no carry out of 256 bits
        reduce_once(r.l);
        return r;
    }
    friend ZK_HD Field operator-(const Field &a, const Field &b) {
        Field r;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)a.l[i] - b.l[i] - borrow;
            r.l[i] = (uint32_t)x;
            borrow = (uint32_t)(x >> 63);
        }
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)r.l[i] + (borrow ? P::M[i] : 0u) + c;
            r.l[i] = (uint32_t)x;
            c = (uint32_t)(x >> 32);
        }
        return r;
    }
    ZK_HD Field neg() const {
        if (is_zero()) return *this;
        Field r;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)P::M[i] - l[i] - borrow;
            r.l[i] = (uint32_t)x;
            borrow = (uint32_t)(x >> 63);
        }
        return r;
    }
    ZK_HD Field dbl() const { return *this + *this; }

    // Montgomery product (CIOS, 32-bit limbs). M < 2^254 => intermediate fits in 9 limbs.
    friend ZK_HD Field operator*(const Field &a, const Field &b) { return mul_impl(a, b); }
    // Out-of-line on purpose: one copy of the ~400-instruction body per field keeps kernels small
    // (instruction cache) and compile times sane; operands travel in VGPRs (by value).
    static __host__ __device__ __attribute__((noinline)) Field mul_impl(Field a, Field b) {
#if defined(__HIP_DEVICE_COMPILE__)
        // gfx950: one inline-asm block (generated by gen_mul_asm.py): FIPS Montgomery with
        // v_mad_u64_u32 + v_addc_co_u32 carry pairs; +44 % over the compiled CIOS loop below.
        uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
        if constexpr (P::ID == 0) {
            asm volatile(ZK_MUL_ASM_FP
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                           "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                         : "v16", "v17", "v18", "v19", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "vcc");
        } else {
            asm volatile(ZK_MUL_ASM_FR
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                           "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                         : "v16", "v17", "v18", "v19", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "vcc");
        }
        Field r;
        r.l[0] = t0; r.l[1] = t1; r.l[2] = t2; r.l[3] = t3; r.l[4] = t4; r.l[5] = t5; r.l[6] = t6; r.l[7] = t7;
        reduce_once(r.l);
        return r;
#else
        // portable CIOS (host compilation only; the product never multiplies on the host)
        uint32_t t[9];
#pragma unroll
        for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t c = 0;
            const uint32_t bi = b.l[i];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint64_t x = (uint64_t)a.l[j] * bi + t[j] + c;
                t[j] = (uint32_t)x;
                c = x >> 32;
            }
            uint64_t x8 = (uint64_t)t[8] + c;
            t[8] = (uint32_t)x8;
            const uint32_t m = t[0] * P::INV;
            uint64_t x = (uint64_t)m * P::M[0] + t[0];
            c = x >> 32;
#pragma unroll
            for (int j = 1; j < 8; j++) {
                x = (uint64_t)m * P::M[j] + t[j] + c;
                t[j - 1] = (uint32_t)x;
                c = x >> 32;
            }
            x = (uint64_t)t[8] + c;
            t[7] = (uint32_t)x;
            t[8] = (uint32_t)(x >> 32);
        }
        Field r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = t[i];
        reduce_once(r.l);
        return r;
#endif
    }
    ZK_HD Field sqr() const { return (*this) * (*this); }
    ZK_HD Field to_mont() const { return (*this) * r2(); }
    ZK_HD Field from_mont() const {
        Field o = zero();
        o.l[0] = 1;
        return (*this) * o;
    }
    // exponent: plain 256-bit integer, little-endian limbs
    ZK_HD Field pow(const uint32_t *e, int nbits = 256) const {
        Field acc = one();
        for (int i = nbits - 1; i >= 0; i--) {
            acc = acc.sqr();
            if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * (*this);
        }
        return acc;
    }
    // Fermat inverse; 0 -> 0
    ZK_HD Field inv() const {
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = P::M[i];
        e[0] -= 2;  // M[0] >= 2 for both moduli, no borrow
        return pow(e, 254);
    }
    static ZK_HD Field from_u32(uint32_t v) {
        Field o = zero();
        o.l[0] = v;
        return o.to_mont();
    }
};

using Fp = Field<FpParams>;
using Fr = Field<FrParams>;

// ---- Fp2 = Fp[u]/(u^2+1) ----
struct Fp2 {
    Fp a0, a1;
    static ZK_HD Fp2 zero() { return Fp2{Fp::zero(), Fp::zero()}; }
    static ZK_HD Fp2 one() { return Fp2{Fp::one(), Fp::zero()}; }
    ZK_HD bool is_zero() const { return a0.is_zero() && a1.is_zero(); }
    ZK_HD bool operator==(const Fp2 &b) const { return a0 == b.a0 && a1 == b.a1; }
    friend ZK_HD Fp2 operator+(const Fp2 &a, const Fp2 &b) { return Fp2{a.a0 + b.a0, a.a1 + b.a1}; }
    friend ZK_HD Fp2 operator-(const Fp2 &a, const Fp2 &b) { return Fp2{a.a0 - b.a0, a.a1 - b.a1}; }
    ZK_HD Fp2 neg() const { return Fp2{a0.neg(), a1.neg()}; }
    ZK_HD Fp2 dbl() const { return Fp2{a0.dbl(), a1.dbl()}; }
    friend ZK_HD Fp2 operator*(const Fp2 &a, const Fp2 &b) {
        Fp t0 = a.a0 * b.a0, t1 = a.a1 * b.a1;
        Fp m = (a.a0 + a.a1) * (b.a0 + b.a1);
        return Fp2{t0 - t1, m - t0 - t1};
    }
    ZK_HD Fp2 sqr() const {
        Fp s = a0 + a1, d = a0 - a1, m = a0 * a1;
        return Fp2{s * d, m.dbl()};
    }
    ZK_HD Fp2 inv() const {
        Fp n = (a0.sqr() + a1.sqr()).inv();
        return Fp2{a0 * n, (a1 * n).neg()};
    }
    ZK_HD Fp2 to_mont() const { return Fp2{a0.to_mont(), a1.to_mont()}; }
    ZK_HD Fp2 from_mont() const { return Fp2{a0.from_mont(), a1.from_mont()}; }
};

}  // namespace zk
