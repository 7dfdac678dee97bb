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
    static constexpr uint32_t R3[8] = {0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u, 0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u};   // R^3 mod p
};
struct FrParams {
    static constexpr int ID = 1;
    // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    static constexpr uint32_t M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xefffffffu;
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t R3[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};   // R^3 mod r
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
#if defined(__HIP_DEVICE_COMPILE__)
    // r = a + b as 256-bit integers (a, b < M: no carry out)
    static ZK_D void asm_add(Field &r, const Field &a, const Field &b) {
        if constexpr (P::ID == 0) {
            asm(ZK_ADD_ASM_FP
                : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7])
                : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                  "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                : "vcc");
        } else {
            asm(ZK_ADD_ASM_FR
                : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7])
                : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                  "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                : "vcc");
        }
    }
    // r < 2M -> r mod M
    static ZK_D void asm_reduce(Field &r) {
        uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
        if constexpr (P::ID == 0) {
            asm(ZK_RED_ASM_FP
                : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
                  "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                :
                : "vcc");
        } else {
            asm(ZK_RED_ASM_FR
                : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
                  "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                :
                : "vcc");
        }
    }
    // weak form: r < 4M -> r - 2M if r >= 2M
    static ZK_D void asm_reduce_w(Field &r) {
        uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
        if constexpr (P::ID == 0) {
            asm(ZK_REDW_ASM_FP
                : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
                  "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                :
                : "vcc");
        } else {
            asm(ZK_REDW_ASM_FR
                : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
                  "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                :
                : "vcc");
        }
    }
    // weak form: r = a - b, + 2M if that borrowed (a, b in [0, 2M) -> r in [0, 2M))
    static ZK_D void asm_sub_w(Field &r, const Field &a, const Field &b) {
        uint32_t m, d0, d1, d2, d3, d4, d5, d6, d7;
        if constexpr (P::ID == 0) {
            asm(ZK_SUB_ASM_FP
                : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(m)
                : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                  "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                : "vcc");
            asm(ZK_SUBFIXW_ASM_FP
                : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
                  "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                : "v"(m)
                : "vcc");
        } else {
            asm(ZK_SUB_ASM_FR
                : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(m)
                : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                  "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                : "vcc");
            asm(ZK_SUBFIXW_ASM_FR
                : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
                  "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                : "v"(m)
                : "vcc");
        }
    }
    // r = a - b mod M
    static ZK_D void asm_sub(Field &r, const Field &a, const Field &b) {
        uint32_t m, d0, d1, d2, d3, d4, d5, d6, d7;
        if constexpr (P::ID == 0) {
            asm(ZK_SUB_ASM_FP
                : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(m)
                : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                  "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                : "vcc");
            asm(ZK_SUBFIX_ASM_FP
                : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
                  "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                : "v"(m)
                : "vcc");
        } else {
            asm(ZK_SUB_ASM_FR
                : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(m)
                : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                  "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                : "vcc");
            asm(ZK_SUBFIX_ASM_FR
                : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]),
                  "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                : "v"(m)
                : "vcc");
        }
    }
#endif
    friend ZK_HD Field operator+(const Field &a, const Field &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        // gfx950: 8 add-with-carry + 8 for the trial subtraction + 8 selects (hipcc turns the portable loop below into ~100
        // instructions of 64-bit arithmetic); two asm statements because one takes at most 30 operands
        Field r;
        asm_add(r, a, b);
        asm_reduce(r);
        return r;
#else
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
#endif
    }
    friend ZK_HD Field operator-(const Field &a, const Field &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Field r;
        asm_sub(r, a, b);
        return r;
#else
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
#endif
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

    // ---- "weak" form: values in [0, 2M) instead of [0, M).  A Montgomery product of weak operands is weak again without
    // the final conditional subtraction (4M < 2^256), which is 8 % of the multiplier block; additions and subtractions cost
    // the same as in canonical form.  Only zero tests and stores need care (is_zero_w, canon).  Canonical values are weak
    // values, so the two forms mix freely as INPUTS of the *_w operations.  Host builds simply stay canonical.
    ZK_HD Field canon() const {
#if defined(__HIP_DEVICE_COMPILE__)
        Field r = *this;
        asm_reduce(r);
        return r;
#else
        Field r = *this;
        reduce_once(r.l);
        return r;
#endif
    }
    ZK_HD bool is_zero_w() const {
        uint32_t o = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { o |= l[i]; q |= l[i] ^ P::M[i]; }
        return o == 0 || q == 0;
    }
    static ZK_HD Field mul_w(const Field &a, const Field &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return mul_block_w(a, b);
#else
        return a.canon() * b.canon();
#endif
    }
    // the same through one shared out-of-line copy (kernels with dozens of multiplication sites: the transforms)
    static __host__ __device__ __attribute__((noinline)) Field mul_wc_impl(Field a, Field b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return mul_block_w(a, b);
#else
        return a.canon() * b.canon();
#endif
    }
    static ZK_HD Field mul_wc(const Field &a, const Field &b) { return mul_wc_impl(a, b); }
    static ZK_HD Field add_w(const Field &a, const Field &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Field r;
        asm_add(r, a, b);
        asm_reduce_w(r);
        return r;
#else
        return a.canon() + b.canon();
#endif
    }
    static ZK_HD Field sub_w(const Field &a, const Field &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Field r;
        asm_sub_w(r, a, b);
        return r;
#else
        return a.canon() - b.canon();
#endif
    }

    // Montgomery product (CIOS, 32-bit limbs). M < 2^254 => intermediate fits in 9 limbs.
    friend ZK_HD Field operator*(const Field &a, const Field &b) { return mul_impl(a, b); }
    // Out-of-line on purpose: one copy of the ~400-instruction body per field keeps kernels small
    // (instruction cache) and compile times sane; operands travel in VGPRs (by value).
#if defined(__HIP_DEVICE_COMPILE__)
    // the multiplier block itself (generated by gen_mul_asm.py): FIPS Montgomery with v_mad_u64_u32 + v_addc_co_u32 carry
    // pairs, ending with the conditional subtraction of the modulus; +44 % over the compiled CIOS loop
    // the same without the final conditional subtraction: operands and result in [0, 2M) ("weak" form; 4M < 2^256)
    static ZK_D Field mul_block_w(const Field &a, const Field &b) {
        uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
        if constexpr (P::ID == 0) {
            asm volatile(ZK_MULW_ASM_FP
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                           "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                         : "v16", "v17", "v18", "v19", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "vcc");
        } else {
            asm volatile(ZK_MULW_ASM_FR
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                           "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                         : "v16", "v17", "v18", "v19", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "vcc");
        }
        Field r;
        r.l[0] = t0; r.l[1] = t1; r.l[2] = t2; r.l[3] = t3; r.l[4] = t4; r.l[5] = t5; r.l[6] = t6; r.l[7] = t7;
        return r;
    }
    static ZK_D Field mul_block(const Field &a, const Field &b) {
        uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
        if constexpr (P::ID == 0) {
            asm volatile(ZK_MUL_ASM_FP
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                           "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                         : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "vcc");
        } else {
            asm volatile(ZK_MUL_ASM_FR
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
                           "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7])
                         : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "vcc");
        }
        Field r;
        r.l[0] = t0; r.l[1] = t1; r.l[2] = t2; r.l[3] = t3; r.l[4] = t4; r.l[5] = t5; r.l[6] = t6; r.l[7] = t7;
        return r;
    }
#endif
    // a * b with the multiplier block expanded in place: for the few innermost loops (bucket accumulation) where the 24 register
    // moves of a call per multiplication show; everything else goes through the shared out-of-line copy below
    static ZK_HD Field mul_inline(const Field &a, const Field &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return mul_block(a, b);
#else
        return mul_impl(a, b);
#endif
    }
    static __host__ __device__ __attribute__((noinline)) Field mul_impl(Field a, Field b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return mul_block(a, b);
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
    // Fermat inverse; 0 -> 0.  Kept as the independent check of inv_gcd (tests, micro-benchmark): ~373 multiplications.
    ZK_HD Field inv_fermat() const {
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = P::M[i];
        e[0] -= 2;  // M[0] >= 2 for both moduli, no borrow
        return pow(e, 254);
    }
    // x^-1 (0 -> 0): the binary-GCD inversion below, 5x cheaper than the ladder
    ZK_HD Field inv() const { return inv_gcd(); }
    // Inverse by the extended binary GCD on word-sized approximations (Pornin, "Optimized binary GCD for modular inversion",
    // eprint 2020/972): 17 outer rounds of 30 branch-free inner steps on 64-bit approximations of (a, b), each round applying
    // the collected 2x2 update to the 256-bit values once.  2*254 - 1 = 507 <= 510 steps suffice for a 254-bit modulus; every lane
    // runs the same instruction stream.  ~7x cheaper than the Fermat ladder above on gfx950 (no Montgomery multiplications
    // inside: 80 word multiplications per round), which is what makes the shared-inversion affine additions of msm.hip pay.
    // Same contract as inv(): Montgomery in, Montgomery out, 0 -> 0.
    ZK_HD Field inv_gcd() const {
        constexpr int K = 30, ROUNDS = 17;
        uint32_t a[8], b[8], u[8], v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { a[i] = l[i]; b[i] = P::M[i]; u[i] = 0; v[i] = 0; }
        u[0] = 1;                       // invariants: a = u * x, b = v * x  (mod M)
        for (int round = 0; round < ROUNDS; round++) {
            // n = bit length of max(a, b); approximations keep the low K bits (they decide the K parity tests) and the 32 bits below bit n
            uint32_t top = 0;           // index of the highest limb where a | b != 0
#pragma unroll
            for (int i = 1; i < 8; i++) top = (a[i] | b[i]) ? (uint32_t)i : top;
            uint64_t xa, xb;
            if (top == 0) {
                xa = a[0]; xb = b[0];
            } else {
                uint32_t ah = 0, am = 0, al = 0, bh = 0, bm = 0, bl = 0;    // limbs top, top-1, top-2 (0 below limb 0)
#pragma unroll
                for (int i = 1; i < 8; i++)
                    if ((uint32_t)i == top) { ah = a[i]; am = a[i - 1]; al = i >= 2 ? a[i - 2] : 0u; bh = b[i]; bm = b[i - 1]; bl = i >= 2 ? b[i - 2] : 0u; }
                const int lz = __builtin_clz(ah | bh);                      // top limb non-zero in a | b
                // the 32 bits ending at the top set bit of a | b
                const uint64_t a96h = ((uint64_t)ah << 32) | am, b96h = ((uint64_t)bh << 32) | bm;
                uint64_t ta = lz ? ((a96h << lz) | ((uint64_t)al >> (32 - lz))) : a96h;
                uint64_t tb = lz ? ((b96h << lz) | ((uint64_t)bl >> (32 - lz))) : b96h;
                ta >>= 32; tb >>= 32;
                const uint32_t n_bits = 32 * top + 32 - (uint32_t)lz;
                if (n_bits <= K + 32) {     // both values fit the approximation exactly
                    xa = ((uint64_t)a[1] << 32) | a[0];
                    xb = ((uint64_t)b[1] << 32) | b[0];
                } else {
                    xa = (ta << K) | (a[0] & ((1u << K) - 1));
                    xb = (tb << K) | (b[0] & ((1u << K) - 1));
                }
            }
            int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 2
            for (int j = 0; j < K; j++) {
                const uint64_t odd = 0 - (xa & 1);
                const uint64_t sw = (xa < xb ? ~(uint64_t)0 : 0) & odd;
                uint64_t t = (xa ^ xb) & sw; xa ^= t; xb ^= t;
                const int32_t sw32 = (int32_t)sw;
                int32_t tf = (f0 ^ f1) & sw32; f0 ^= tf; f1 ^= tf;
                int32_t tg = (g0 ^ g1) & sw32; g0 ^= tg; g1 ^= tg;
                xa = (xa - (xb & odd)) >> 1;
                f0 -= f1 & (int32_t)odd; g0 -= g1 & (int32_t)odd;
                f1 = (int32_t)((uint32_t)f1 << 1); g1 = (int32_t)((uint32_t)g1 << 1);
            }
            // (a, b) <- (a f0 + b g0, a f1 + b g1) / 2^K   (exact), made non-negative
            uint32_t na[8], nb_[8];
            {
                int64_t ca = 0, cb = 0;
                uint32_t pa = 0, pb = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    ca += (int64_t)a[i] * f0 + (int64_t)b[i] * g0;
                    cb += (int64_t)a[i] * f1 + (int64_t)b[i] * g1;
                    const uint32_t wa = (uint32_t)ca, wb = (uint32_t)cb;
                    if (i) { na[i - 1] = (pa >> K) | (wa << (32 - K)); nb_[i - 1] = (pb >> K) | (wb << (32 - K)); }
                    pa = wa; pb = wb;
                    ca >>= 32; cb >>= 32;
                }
                const uint32_t wa = (uint32_t)ca, wb = (uint32_t)cb;      // sign-carrying 9th word
                na[7] = (pa >> K) | (wa << (32 - K)); nb_[7] = (pb >> K) | (wb << (32 - K));
                const bool nega = ca < 0, negb = cb < 0;
                uint32_t ba = 1, bb = 1;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint64_t sa = (uint64_t)(nega ? ~na[i] : na[i]) + (nega ? ba : 0u);
                    const uint64_t sb = (uint64_t)(negb ? ~nb_[i] : nb_[i]) + (negb ? bb : 0u);
                    a[i] = (uint32_t)sa; b[i] = (uint32_t)sb;
                    ba = (uint32_t)(sa >> 32); bb = (uint32_t)(sb >> 32);
                }
                if (nega) { f0 = -f0; g0 = -g0; }
                if (negb) { f1 = -f1; g1 = -g1; }
            }
            // (u, v) <- (u f0 + v g0, u f1 + v g1) / 2^K mod M: one Montgomery step with the word 2^K, then into [0, M)
            {
                const uint32_t minv = P::INV & ((1u << K) - 1);
                const uint32_t tu = (((uint32_t)((int64_t)u[0] * f0 + (int64_t)v[0] * g0)) * minv) & ((1u << K) - 1);
                const uint32_t tv = (((uint32_t)((int64_t)u[0] * f1 + (int64_t)v[0] * g1)) * minv) & ((1u << K) - 1);
                int64_t cu = 0, cv = 0;
                uint32_t pu = 0, pv = 0, nu[8], nv[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    // three products of magnitude < 2^62: the Montgomery term is accumulated apart to keep each sum inside int64
                    const uint64_t mu = (uint64_t)P::M[i] * tu, mv = (uint64_t)P::M[i] * tv;
                    cu += (int64_t)u[i] * f0 + (int64_t)v[i] * g0;
                    cv += (int64_t)u[i] * f1 + (int64_t)v[i] * g1;
                    cu += (int64_t)(mu & 0xffffffffu); cv += (int64_t)(mv & 0xffffffffu);
                    const uint32_t wu = (uint32_t)cu, wv = (uint32_t)cv;
                    if (i) { nu[i - 1] = (pu >> K) | (wu << (32 - K)); nv[i - 1] = (pv >> K) | (wv << (32 - K)); }
                    pu = wu; pv = wv;
                    cu = (cu >> 32) + (int64_t)(mu >> 32); cv = (cv >> 32) + (int64_t)(mv >> 32);
                }
                nu[7] = (pu >> K) | ((uint32_t)cu << (32 - K)); nv[7] = (pv >> K) | ((uint32_t)cv << (32 - K));
                // value in (-M, 2M): sign from the 9th word (after the shift: bit K of it and above)
                const bool negu = cu < 0, negv = cv < 0;
                uint32_t cy_u = 0, cy_v = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint64_t su = (uint64_t)nu[i] + (negu ? P::M[i] : 0u) + cy_u;
                    const uint64_t sv = (uint64_t)nv[i] + (negv ? P::M[i] : 0u) + cy_v;
                    u[i] = (uint32_t)su; v[i] = (uint32_t)sv;
                    cy_u = (uint32_t)(su >> 32); cy_v = (uint32_t)(sv >> 32);
                }
                // a non-negative value may still be in [M, 2M): bit 256 cannot be set (2M < 2^255)
                if (!negu) reduce_once(u);
                if (!negv) reduce_once(v);
            }
        }
        // b = gcd(x, M): 1 unless x == 0; then x^-1 = v as plain integers, and (x R)^-1 R^2 = x^-1 R
        uint32_t rest = b[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < 8; i++) rest |= b[i];
        Field r, r3;
#pragma unroll
        for (int i = 0; i < 8; i++) { r.l[i] = rest ? 0u : v[i]; r3.l[i] = P::R3[i]; }
        return r * r3;
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
    // weak form (both coordinates in [0, 2p), see Field): same static interface as Field so that XYZZ::madd_w / add_w serve G2 too
    ZK_HD Fp2 canon() const { return Fp2{a0.canon(), a1.canon()}; }
    ZK_HD bool is_zero_w() const { return a0.is_zero_w() && a1.is_zero_w(); }
    static ZK_HD Fp2 add_w(const Fp2 &a, const Fp2 &b) { return Fp2{Fp::add_w(a.a0, b.a0), Fp::add_w(a.a1, b.a1)}; }
    static ZK_HD Fp2 sub_w(const Fp2 &a, const Fp2 &b) { return Fp2{Fp::sub_w(a.a0, b.a0), Fp::sub_w(a.a1, b.a1)}; }
    static ZK_HD Fp2 mul_wc(const Fp2 &a, const Fp2 &b) {
        const Fp t0 = Fp::mul_wc(a.a0, b.a0), t1 = Fp::mul_wc(a.a1, b.a1);
        const Fp m = Fp::mul_wc(Fp::add_w(a.a0, a.a1), Fp::add_w(b.a0, b.a1));
        return Fp2{Fp::sub_w(t0, t1), Fp::sub_w(Fp::sub_w(m, t0), t1)};
    }
    static ZK_HD Fp2 mul_w(const Fp2 &a, const Fp2 &b) { return mul_wc(a, b); }      // the tower never expands the block in place
    ZK_HD Fp2 to_mont() const { return Fp2{a0.to_mont(), a1.to_mont()}; }
    ZK_HD Fp2 from_mont() const { return Fp2{a0.from_mont(), a1.from_mont()}; }
};

}  // namespace zk
