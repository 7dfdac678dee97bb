// BN254 prime fields on 9 SIGNED 29-bit limbs, Montgomery radix 2^261 -- the representation of the innermost MSM loops on gfx950.
//
// Why a second representation (field.cuh keeps the 8 x 32 one for storage and for everything that is not hot):
// on MI355X every integer-multiply / carry instruction issues at the same rate (v_mad_u64_u32 = v_addc_co_u32 = v_mul_lo_u32,
// ~4.5 cycles per wavefront instruction and SIMD; plain VOP2 add / and ~2.5; scripts/ubench/bank_probe.hip), so the cost of a
// Montgomery product is its INSTRUCTION count.  With saturated 32-bit limbs each of the 128 partial products needs a carry
// instruction beside it (256 + moves).  With 29-bit limbs a whole column -- up to 9 products a_i*b_j, 9 products m_i*M_j and
// the carry from the previous column -- fits one signed 64-bit accumulator: 162 v_mad_i64_i32 and no carry instruction at all.
// Signed limbs make subtraction a per-limb v_sub (no borrow chain, no "+ k*M" correction) and let two products share one
// reduction (a*b - c*d: 243 multiply-adds instead of 324).  Measured (scripts/ubench/mul9.hip): 172 G products/s against 135,
// G1 mixed addition 16.5 G/s against 12.8.
//
// Value model.  An element is an integer v = sum l[i] * 2^(29 i) taken mod M; it is NOT kept in [0, M):
//   strict : l[0..7] in [0, 2^29), l[8] signed and small -- what every product returns; unique per integer v
//   lazy   : any limbs, |l[i]| < k * 2^29 -- what add / sub return (k = sum of the operands' k)
// A reduction (mont_lockstep) needs, per column, 9 * sum_t A_t * B_t + 9 * 2^58 < 2^63 where A_t, B_t bound the limb magnitudes of
// the t-th product's operands: one product may take (strict or 1-lazy) x (2-lazy), two fused products (1-lazy x 1-lazy) each.  Callers
// state their bounds where they use lazy operands.  Values: the result of a reduction lies in (S / 2^261, S / 2^261 + M) for the
// integer S = sum of products; all call sites keep |S| < 20 M^2, so results stay within (-M/8, M + M/8) (M / 2^261 < 2^-7.4),
// sums of a few of them within (-4M, 4M), and |v| < 2^258 everywhere (the top limb never leaves 32 bits).
//
// Replaces the same gnark-crypto v0.19.2 fp / fr element arithmetic as field.cuh (ecc/bn254/fp/element.go Mul / Add / Sub reached
// from MultiExp's bucket additions, provers.go:179,269).
#pragma once
#include "field.cuh"
#include "field9_consts.inc"

namespace zk {

struct Fp9Params {
    using Base = FpParams;
    static constexpr int32_t M[9] = ZK_F9_FP_M;
    static constexpr uint32_t PINV = ZK_F9_FP_PINV;
    static constexpr int32_t CIN[9] = ZK_F9_FP_CIN;
    static constexpr int32_t COUT[9] = ZK_F9_FP_COUT;
    static constexpr int32_t ONE[9] = ZK_F9_FP_ONE;
    static constexpr uint32_t SCALE32[8] = ZK_F9_FP_SCALE32;
};
struct Fr9Params {
    using Base = FrParams;
    static constexpr int32_t M[9] = ZK_F9_FR_M;
    static constexpr uint32_t PINV = ZK_F9_FR_PINV;
    static constexpr int32_t CIN[9] = ZK_F9_FR_CIN;
    static constexpr int32_t COUT[9] = ZK_F9_FR_COUT;
    static constexpr int32_t ONE[9] = ZK_F9_FR_ONE;
    static constexpr uint32_t SCALE32[8] = ZK_F9_FR_SCALE32;
};

template <class Q>
struct Field9 {
    using Words = Field<typename Q::Base>;
    static constexpr uint32_t MASK = 0x1fffffffu;
    int32_t l[9];

    // acc += a * b: one v_mad_i64_i32 (there is no builtin for it; a one-instruction asm leaves allocation and scheduling to the compiler)
    static ZK_HD void mac(int64_t &acc, int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
        acc += (int64_t)a * (int64_t)b;
#endif
    }
    static ZK_HD void mac_const(int64_t &acc, int32_t a, int32_t b) {      // b: a limb of the modulus (scalar register)
#if defined(__HIP_DEVICE_COMPILE__)
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc");
#else
        acc += (int64_t)a * (int64_t)b;
#endif
    }
    // One term of a reduction: a * b, or a^2 when b == nullptr (45 instead of 81 multiply-adds: the cross products are taken once
    // against 2a; same column sums, so the same limb bounds as a * a).
    struct Term {
        const Field9 *a, *b;
    };
    // K independent Montgomery reductions in lockstep; reduction j = (sum of its np[j] <= 2 terms) / 2^261 mod M, strict.
    // Product scanning: column k of reduction j collects its products and the multiples m_i * M_(k-i) in one 64-bit accumulator,
    // m_k clears the column's low 29 bits, the rest carries into column k+1.  The K accumulator chains are independent and are
    // issued interleaved (a v_mad_i64_i32 that reads the result of the one just before it costs a wait state on gfx950).
    template <int K>
    static ZK_HD void mont_lockstep(const Term (&tm)[K][2], const int (&np)[K], Field9 (&r)[K]) {
        int64_t acc[K];
        int32_t m[K][9];
        Field9 a2[K][2];
#pragma unroll
        for (int j = 0; j < K; j++) {
            acc[j] = 0;
#pragma unroll
            for (int t = 0; t < 2; t++)
                if (t < np[j] && tm[j][t].b == nullptr) a2[j][t] = add(*tm[j][t].a, *tm[j][t].a);
        }
#pragma unroll
        for (int k = 0; k < 17; k++) {
            const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
            for (int i = lo; i <= hi; i++) {
                const int jj = k - i;
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int j = 0; j < K; j++) {
                        if (t >= np[j]) continue;
                        const Term &T = tm[j][t];
                        if (T.b != nullptr) mac(acc[j], T.a->l[i], T.b->l[jj]);
                        else if (i < jj) mac(acc[j], a2[j][t].l[i], T.a->l[jj]);
                        else if (i == jj) mac(acc[j], T.a->l[i], T.a->l[i]);
                    }
            }
            if (k < 9) {
#pragma unroll
                for (int i = 0; i < k; i++)
#pragma unroll
                    for (int j = 0; j < K; j++) mac_const(acc[j], m[j][i], Q::M[k - i]);
#pragma unroll
                for (int j = 0; j < K; j++) m[j][k] = (int32_t)(((uint32_t)acc[j] * Q::PINV) & MASK);
#pragma unroll
                for (int j = 0; j < K; j++) {
                    mac_const(acc[j], m[j][k], Q::M[0]);
                    acc[j] >>= 29;
                }
            } else {
#pragma unroll
                for (int i = k - 8; i < 9; i++)
#pragma unroll
                    for (int j = 0; j < K; j++) mac_const(acc[j], m[j][i], Q::M[k - i]);
#pragma unroll
                for (int j = 0; j < K; j++) {
                    r[j].l[k - 9] = (int32_t)((uint32_t)acc[j] & MASK);
                    acc[j] >>= 29;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < K; j++) r[j].l[8] = (int32_t)acc[j];
    }
    static ZK_HD Field9 mul(const Field9 &a, const Field9 &b) {
        const Term t[1][2] = {{{&a, &b}, {nullptr, nullptr}}};
        const int np[1] = {1};
        Field9 r[1];
        mont_lockstep<1>(t, np, r);
        return r[0];
    }
    static ZK_HD Field9 sqr(const Field9 &a) {
        const Term t[1][2] = {{{&a, nullptr}, {nullptr, nullptr}}};
        const int np[1] = {1};
        Field9 r[1];
        mont_lockstep<1>(t, np, r);
        return r[0];
    }
    // a*b - c*d with ONE reduction
    static ZK_HD Field9 mulsub(const Field9 &a, const Field9 &b, const Field9 &c, const Field9 &d) {
        const Field9 nc = neg(c);
        const Term t[1][2] = {{{&a, &b}, {&nc, &d}}};
        const int np[1] = {2};
        Field9 r[1];
        mont_lockstep<1>(t, np, r);
        return r[0];
    }
    // a*b + c*d with ONE reduction
    static ZK_HD Field9 muladd(const Field9 &a, const Field9 &b, const Field9 &c, const Field9 &d) {
        const Term t[1][2] = {{{&a, &b}, {&c, &d}}};
        const int np[1] = {2};
        Field9 r[1];
        mont_lockstep<1>(t, np, r);
        return r[0];
    }
    // two / three independent products in lockstep
    static ZK_HD void mul2(Field9 &r0, const Field9 &a0, const Field9 &b0, Field9 &r1, const Field9 &a1, const Field9 &b1) {
        const Term t[2][2] = {{{&a0, &b0}, {nullptr, nullptr}}, {{&a1, &b1}, {nullptr, nullptr}}};
        const int np[2] = {1, 1};
        Field9 r[2];
        mont_lockstep<2>(t, np, r);
        r0 = r[0]; r1 = r[1];
    }
    static ZK_HD void sqr2(Field9 &r0, const Field9 &a0, Field9 &r1, const Field9 &a1) {
        const Term t[2][2] = {{{&a0, nullptr}, {nullptr, nullptr}}, {{&a1, nullptr}, {nullptr, nullptr}}};
        const int np[2] = {1, 1};
        Field9 r[2];
        mont_lockstep<2>(t, np, r);
        r0 = r[0]; r1 = r[1];
    }
    static ZK_HD void mul3(Field9 &r0, const Field9 &a0, const Field9 &b0, Field9 &r1, const Field9 &a1, const Field9 &b1, Field9 &r2, const Field9 &a2,
                           const Field9 &b2) {
        const Term t[3][2] = {{{&a0, &b0}, {nullptr, nullptr}}, {{&a1, &b1}, {nullptr, nullptr}}, {{&a2, &b2}, {nullptr, nullptr}}};
        const int np[3] = {1, 1, 1};
        Field9 r[3];
        mont_lockstep<3>(t, np, r);
        r0 = r[0]; r1 = r[1]; r2 = r[2];
    }
    // r0 = a*b - c*d (one reduction), r1 = e*f, in lockstep
    static ZK_HD void mulsub_mul(Field9 &r0, const Field9 &a, const Field9 &b, const Field9 &c, const Field9 &d, Field9 &r1, const Field9 &e, const Field9 &f) {
        const Field9 nc = neg(c);
        const Term t[2][2] = {{{&a, &b}, {&nc, &d}}, {{&e, &f}, {nullptr, nullptr}}};
        const int np[2] = {2, 1};
        Field9 r[2];
        mont_lockstep<2>(t, np, r);
        r0 = r[0]; r1 = r[1];
    }
    // lazy limb-wise operations: no carries, the limb bounds add up
    static ZK_HD Field9 add(const Field9 &a, const Field9 &b) {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
        return r;
    }
    static ZK_HD Field9 sub(const Field9 &a, const Field9 &b) {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = a.l[i] - b.l[i];
        return r;
    }
    static ZK_HD Field9 neg(const Field9 &a) {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = -a.l[i];
        return r;
    }
    // -a when s is all ones, a when s is 0
    static ZK_HD Field9 cneg(const Field9 &a, int32_t s) {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = (a.l[i] ^ s) - s;
        return r;
    }
    // signed carry propagation: any lazy value -> strict, same integer
    static ZK_HD Field9 norm(Field9 x) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t c = x.l[i] >> 29;
            x.l[i] &= (int32_t)MASK;
            x.l[i + 1] += c;
        }
        return x;
    }
    static ZK_HD Field9 zero() {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = 0;
        return r;
    }
    static ZK_HD Field9 one() {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = Q::ONE[i];
        return r;
    }
    // all limbs zero (the exact encoding of "no value": infinity flags)
    ZK_HD bool is_zero_exact() const {
        uint32_t z = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) z |= (uint32_t)l[i];
        return z == 0;
    }
    // strict value in (-M, 2M): is it 0 mod M?  (strict limbs are unique per integer, so this is v == 0 or v == M)
    ZK_HD bool is_zero_mod() const {
        uint32_t z = 0, e = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { z |= (uint32_t)l[i]; e |= (uint32_t)(l[i] ^ Q::M[i]); }
        return z == 0 || e == 0;
    }
    // (hi:lo) >> sh, low word.  The builtin, not a 64-bit expression: the optimiser fuses `(u64)l[w+1] << 32 | l[w]` into ONE 64-bit load
    // of the operand, and a struct read through overlapping 64-bit loads is no longer split into registers (it stays in scratch)
    static ZK_HD uint32_t funnel(uint32_t hi, uint32_t lo, int sh) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
    }
    // the 8 words of a value < 2^256 cut into limbs (no arithmetic: whatever Montgomery form the words are in stays)
    static ZK_HD Field9 from_words(const Words &x) {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
            uint32_t v;
            if (sh == 0) v = x.l[w];
            else if (w + 1 < 8) v = funnel(x.l[w + 1], x.l[w], sh);
            else v = x.l[w] >> sh;
            r.l[i] = (int32_t)(v & MASK);
        }
        return r;
    }
    // any value in (-4M, 4M) -> its canonical representative in [0, M) as 8 words
    static ZK_HD Words to_words(Field9 x) {
#pragma unroll
        for (int r = 0; r < 2; r++) {              // + 4M in two steps: 4 * M[i] alone already fills 31 bits
#pragma unroll
            for (int i = 0; i < 9; i++) x.l[i] += 2 * Q::M[i];
            x = norm(x);
        }                                          // in (0, 8M)
#pragma unroll
        for (int s = 2; s >= 0; s--) {             // subtract 4M, 2M, M whenever that stays non-negative
            Field9 y;
#pragma unroll
            for (int i = 0; i < 9; i++) y.l[i] = x.l[i] - (Q::M[i] << s);
            y = norm(y);
            const bool keep = y.l[8] >= 0;
#pragma unroll
            for (int i = 0; i < 9; i++) x.l[i] = keep ? y.l[i] : x.l[i];
        }
        Words r;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // word j = bits 32j .. 32j+31 = limb lo (from its bit s) | next limb(s) shifted up
            const int lo = (32 * j) / 29, s = 32 * j - 29 * lo;
            uint32_t v = (uint32_t)x.l[lo] >> s;
            if (lo + 1 < 9) v |= (uint32_t)x.l[lo + 1] << (29 - s);
            if (lo + 2 < 9 && 58 - s < 32) v |= (uint32_t)x.l[lo + 2] << (58 - s);
            r.l[j] = v;
        }
        return r;
    }
    // 2^256-Montgomery words (field.cuh) <-> 2^261-Montgomery limbs: one product with a constant each way
    static ZK_HD Field9 constant(const int32_t (&c)[9]) {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = c[i];
        return r;
    }
    static ZK_HD Field9 from_field(const Words &x) { return mul(from_words(x), constant(Q::CIN)); }
    static ZK_HD Words to_field(const Field9 &x) { return to_words(mul(x, constant(Q::COUT))); }
};
using Fp9 = Field9<Fp9Params>;
using Fr9 = Field9<Fr9Params>;

}  // namespace zk
