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
// integer S = sum of products; all call sites keep |S| < 22 M^2 (the largest: P^2 in XYZZ9::madd / add, P = U2 - X with the state X in
// (-3.52 M, 1.5 M) and U2 in (-M/8, 9M/8), |P| < 4.65 M), so results stay within (-0.14 M, M + 0.14 M) (M / 2^261 < 2^-7.4), sums of
// a few of them within (-4M, 4M) -- the input range of to_words; X, the widest state, stays within (-3.52 M, 1.5 M) -- and
// |v| < 2^258 everywhere (the top limb never leaves 32 bits).  tests/host/field9_check.cpp drives these extremes on the host build.
//
// Replaces the same gnark-crypto v0.19.2 fp / fr element arithmetic as field.cuh (ecc/bn254/fp/element.go Mul / Add / Sub reached
// from MultiExp's bucket additions, provers.go:179,269).
#pragma once
#include <utility>
#include "field.cuh"
#include "field9_consts.inc"

namespace zk {

// ZK_F9_COLUMN_ASM=0 (A/B builds): one asm statement per multiply-add, as before round 4
#ifndef ZK_F9_COLUMN_ASM
#define ZK_F9_COLUMN_ASM 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#include "field9_mac.inc"
#endif

struct Fp9Params {
    using Base = FpParams;
    static constexpr int32_t M[9] = ZK_F9_FP_M;
    static constexpr uint32_t PINV = ZK_F9_FP_PINV;
    static constexpr int32_t CIN[9] = ZK_F9_FP_CIN;
    static constexpr int32_t COUT[9] = ZK_F9_FP_COUT;
    static constexpr int32_t ONE[9] = ZK_F9_FP_ONE;
    static constexpr uint32_t SCALE32[8] = ZK_F9_FP_SCALE32;
    static constexpr int32_t M2[9] = ZK_F9_FP_2M, M4[9] = ZK_F9_FP_4M, M5[9] = ZK_F9_FP_5M;      // strict limbs of 2M, 4M, 5M
};
struct Fr9Params {
    using Base = FrParams;
    static constexpr int32_t M[9] = ZK_F9_FR_M;
    static constexpr uint32_t PINV = ZK_F9_FR_PINV;
    static constexpr int32_t CIN[9] = ZK_F9_FR_CIN;
    static constexpr int32_t COUT[9] = ZK_F9_FR_COUT;
    static constexpr int32_t ONE[9] = ZK_F9_FR_ONE;
    static constexpr uint32_t SCALE32[8] = ZK_F9_FR_SCALE32;
    static constexpr int32_t M2[9] = ZK_F9_FR_2M, M4[9] = ZK_F9_FR_4M, M5[9] = ZK_F9_FR_5M;
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
    // One term of a reduction: a * b.  sym promises a_i * b_j == a_j * b_i for all i, j (a == b: a square; a == -b: a negated
    // square); the cross products are then taken once against 2a -- 45 instead of 81 multiply-adds, same column sums and therefore
    // the same limb bounds as the full product.  a == nullptr: unused slot.
    struct Term {
        const Field9 *a, *b;
        bool sym;
    };
    // K independent Montgomery reductions in lockstep; reduction j = (sum of its np[j] <= 2 terms) / 2^261 mod M, strict.
    // Product scanning: column k of reduction j collects its products and the multiples m_i * M_(k-i) in one 64-bit accumulator,
    // m_k clears the column's low 29 bits, the rest carries into column k+1.  The K accumulator chains are independent and are
    // issued interleaved (a v_mad_i64_i32 that reads the result of the one just before it costs a wait state on gfx950).
    // On the device a column's multiply-adds on one accumulator are ONE asm statement (field9_mac.inc, Mac9<N>): with a statement per
    // multiply-add hipcc pads every one with an s_nop -- 1 203 of the 3 900 instructions of the comb walk's loop.
    template <int N>
    static ZK_HD void macn(int64_t &acc, const int32_t (&a)[N], const int32_t (&b)[N]) {
#if defined(__HIP_DEVICE_COMPILE__) && ZK_F9_COLUMN_ASM
        Mac9<N>::vv(acc, a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < N; i++) mac(acc, a[i], b[i]);
#else
        for (int i = 0; i < N; i++) acc += (int64_t)a[i] * (int64_t)b[i];
#endif
    }
    template <int N>
    static ZK_HD void macn_const(int64_t &acc, const int32_t (&a)[N], const int32_t (&b)[N]) {      // b: limbs of the modulus (scalar registers)
#if defined(__HIP_DEVICE_COMPILE__) && ZK_F9_COLUMN_ASM
        Mac9<N>::vs(acc, a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < N; i++) mac_const(acc, a[i], b[i]);
#else
        for (int i = 0; i < N; i++) acc += (int64_t)a[i] * (int64_t)b[i];
#endif
    }
    template <int K, int k>
    static ZK_HD void lockstep_column(const Term (&tm)[K][2], const int (&np)[K], const Field9 (&a2)[K][2], int64_t (&acc)[K], int32_t (&m)[K][9], Field9 (&r)[K]) {
        constexpr int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8, N = hi - lo + 1;
        // symmetric terms: cross products i < k - i once against 2a, the diagonal i == k - i once
        constexpr int half = (k - 1) / 2, chi = k == 0 ? -1 : (hi < half ? hi : half);
        constexpr int NC = chi >= lo ? chi - lo + 1 : 0, ND = (k % 2 == 0 && k / 2 >= lo && k / 2 <= hi) ? 1 : 0, NS = NC + ND;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int j = 0; j < K; j++) {
                if (t >= np[j]) continue;
                const Term &T = tm[j][t];
                if (!T.sym) {
                    int32_t xa[N], xb[N];
#pragma unroll
                    for (int i = lo; i <= hi; i++) { xa[i - lo] = T.a->l[i]; xb[i - lo] = T.b->l[k - i]; }
                    macn<N>(acc[j], xa, xb);
                } else {
                    int32_t xa[NS], xb[NS];
#pragma unroll
                    for (int q = 0; q < NC; q++) { xa[q] = a2[j][t].l[lo + q]; xb[q] = T.b->l[k - lo - q]; }
                    if (ND) { xa[NS - 1] = T.a->l[k / 2]; xb[NS - 1] = T.b->l[k / 2]; }
                    macn<NS>(acc[j], xa, xb);
                }
            }
        if constexpr (k < 9) {
            if constexpr (k > 0) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    int32_t xm[k], xM[k];
#pragma unroll
                    for (int i = 0; i < k; i++) { xm[i] = m[j][i]; xM[i] = Q::M[k - i]; }
                    macn_const<k>(acc[j], xm, xM);
                }
            }
#pragma unroll
            for (int j = 0; j < K; j++) m[j][k] = (int32_t)(((uint32_t)acc[j] * Q::PINV) & MASK);
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int32_t xm[1] = {m[j][k]}, xM[1] = {Q::M[0]};
                macn_const<1>(acc[j], xm, xM);
                acc[j] >>= 29;
            }
        } else {
            constexpr int NR = 17 - k;
#pragma unroll
            for (int j = 0; j < K; j++) {
                int32_t xm[NR], xM[NR];
#pragma unroll
                for (int i = k - 8; i < 9; i++) { xm[i - (k - 8)] = m[j][i]; xM[i - (k - 8)] = Q::M[k - i]; }
                macn_const<NR>(acc[j], xm, xM);
            }
#pragma unroll
            for (int j = 0; j < K; j++) {
                r[j].l[k - 9] = (int32_t)((uint32_t)acc[j] & MASK);
                acc[j] >>= 29;
            }
        }
    }
    template <int K, int... Ks>
    static ZK_HD void lockstep_columns(const Term (&tm)[K][2], const int (&np)[K], const Field9 (&a2)[K][2], int64_t (&acc)[K], int32_t (&m)[K][9], Field9 (&r)[K],
                                       std::integer_sequence<int, Ks...>) {
        (lockstep_column<K, Ks>(tm, np, a2, acc, m, r), ...);
    }
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
                if (t < np[j] && tm[j][t].sym) a2[j][t] = add(*tm[j][t].a, *tm[j][t].a);
        }
        lockstep_columns<K>(tm, np, a2, acc, m, r, std::make_integer_sequence<int, 17>());
#pragma unroll
        for (int j = 0; j < K; j++) r[j].l[8] = (int32_t)acc[j];
    }
    static ZK_HD Field9 mul(const Field9 &a, const Field9 &b) {
        const Term t[1][2] = {{{&a, &b, false}, {nullptr, nullptr, false}}};
        const int np[1] = {1};
        Field9 r[1];
        mont_lockstep<1>(t, np, r);
        return r[0];
    }
    static ZK_HD Field9 sqr(const Field9 &a) {
        const Term t[1][2] = {{{&a, &a, true}, {nullptr, nullptr, false}}};
        const int np[1] = {1};
        Field9 r[1];
        mont_lockstep<1>(t, np, r);
        return r[0];
    }
    // a*b - c*d with ONE reduction
    static ZK_HD Field9 mulsub(const Field9 &a, const Field9 &b, const Field9 &c, const Field9 &d) {
        const Field9 nc = neg(c);
        const Term t[1][2] = {{{&a, &b, false}, {&nc, &d, false}}};
        const int np[1] = {2};
        Field9 r[1];
        mont_lockstep<1>(t, np, r);
        return r[0];
    }
    // a*b + c*d with ONE reduction
    static ZK_HD Field9 muladd(const Field9 &a, const Field9 &b, const Field9 &c, const Field9 &d) {
        const Term t[1][2] = {{{&a, &b, false}, {&c, &d, false}}};
        const int np[1] = {2};
        Field9 r[1];
        mont_lockstep<1>(t, np, r);
        return r[0];
    }
    // two / three independent products in lockstep
    static ZK_HD void mul2(Field9 &r0, const Field9 &a0, const Field9 &b0, Field9 &r1, const Field9 &a1, const Field9 &b1) {
        const Term t[2][2] = {{{&a0, &b0, false}, {nullptr, nullptr, false}}, {{&a1, &b1, false}, {nullptr, nullptr, false}}};
        const int np[2] = {1, 1};
        Field9 r[2];
        mont_lockstep<2>(t, np, r);
        r0 = r[0]; r1 = r[1];
    }
    static ZK_HD void sqr2(Field9 &r0, const Field9 &a0, Field9 &r1, const Field9 &a1) {
        const Term t[2][2] = {{{&a0, &a0, true}, {nullptr, nullptr, false}}, {{&a1, &a1, true}, {nullptr, nullptr, false}}};
        const int np[2] = {1, 1};
        Field9 r[2];
        mont_lockstep<2>(t, np, r);
        r0 = r[0]; r1 = r[1];
    }
    static ZK_HD void mul3(Field9 &r0, const Field9 &a0, const Field9 &b0, Field9 &r1, const Field9 &a1, const Field9 &b1, Field9 &r2, const Field9 &a2,
                           const Field9 &b2) {
        const Term t[3][2] = {{{&a0, &b0, false}, {nullptr, nullptr, false}}, {{&a1, &b1, false}, {nullptr, nullptr, false}}, {{&a2, &b2, false}, {nullptr, nullptr, false}}};
        const int np[3] = {1, 1, 1};
        Field9 r[3];
        mont_lockstep<3>(t, np, r);
        r0 = r[0]; r1 = r[1]; r2 = r[2];
    }
    // r0 = a*b - c*d (one reduction), r1 = e*f, in lockstep
    static ZK_HD void mulsub_mul(Field9 &r0, const Field9 &a, const Field9 &b, const Field9 &c, const Field9 &d, Field9 &r1, const Field9 &e, const Field9 &f) {
        const Field9 nc = neg(c);
        const Term t[2][2] = {{{&a, &b, false}, {&nc, &d, false}}, {{&e, &f, false}, {nullptr, nullptr, false}}};
        const int np[2] = {2, 1};
        Field9 r[2];
        mont_lockstep<2>(t, np, r);
        r0 = r[0]; r1 = r[1];
    }
    // rr - ppp - 2q, normalised (the X3 of the addition formulas: three strict operands, a 4-lazy difference)
    static ZK_HD Field9 x3(const Field9 &rr, const Field9 &ppp, const Field9 &q) {
        Field9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = rr.l[i] - ppp.l[i] - 2 * q.l[i];
        return norm(r);
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
        // (branch-free on purpose: an early exit on the lowest limb made the bucket walk 8 % SLOWER -- 79 vs 73 ms -- the branch splits
        // the block the scheduler overlaps with the next products)
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

// Fp2 = Fp[u] / (u^2 + 1) over the 9-limb Fp: same static interface as Field9, so that XYZZ9 (curve9.cuh) serves G2 as well.
// A product is FOUR limb products and TWO reductions (c0 = a0 b0 - a1 b1 and c1 = a0 b1 + a1 b0 each share one; the two
// reductions run in lockstep) = 486 multiply-adds, against three full 8 x 32 products + five modular additions for Karatsuba
// on the saturated field; a square is 369.  Every reduction here sums TWO products, so operands may be 1-lazy at most
// (strict - strict), never 2-lazy; results are strict.
struct Fp29 {
    using Words = Fp2;
    using T = Fp9::Term;
    Fp9 c0, c1;
    static ZK_HD Fp29 zero() { return Fp29{Fp9::zero(), Fp9::zero()}; }
    static ZK_HD Fp29 one() { return Fp29{Fp9::one(), Fp9::zero()}; }
    ZK_HD bool is_zero_exact() const { return c0.is_zero_exact() && c1.is_zero_exact(); }
    ZK_HD bool is_zero_mod() const { return c0.is_zero_mod() && c1.is_zero_mod(); }
    static ZK_HD Fp29 add(const Fp29 &a, const Fp29 &b) { return Fp29{Fp9::add(a.c0, b.c0), Fp9::add(a.c1, b.c1)}; }
    static ZK_HD Fp29 sub(const Fp29 &a, const Fp29 &b) { return Fp29{Fp9::sub(a.c0, b.c0), Fp9::sub(a.c1, b.c1)}; }
    static ZK_HD Fp29 neg(const Fp29 &a) { return Fp29{Fp9::neg(a.c0), Fp9::neg(a.c1)}; }
    static ZK_HD Fp29 cneg(const Fp29 &a, int32_t s) { return Fp29{Fp9::cneg(a.c0, s), Fp9::cneg(a.c1, s)}; }
    static ZK_HD Fp29 norm(const Fp29 &a) { return Fp29{Fp9::norm(a.c0), Fp9::norm(a.c1)}; }
    static ZK_HD Fp29 x3(const Fp29 &rr, const Fp29 &ppp, const Fp29 &q) { return Fp29{Fp9::x3(rr.c0, ppp.c0, q.c0), Fp9::x3(rr.c1, ppp.c1, q.c1)}; }
    static ZK_HD Fp29 from_words(const Fp2 &x) { return Fp29{Fp9::from_words(x.a0), Fp9::from_words(x.a1)}; }
    static ZK_HD Fp2 to_words(const Fp29 &x) { return Fp2{Fp9::to_words(x.c0), Fp9::to_words(x.c1)}; }
    static ZK_HD Fp29 from_field(const Fp2 &x) { return Fp29{Fp9::from_field(x.a0), Fp9::from_field(x.a1)}; }
    static ZK_HD Fp2 to_field(const Fp29 &x) { return Fp2{Fp9::to_field(x.c0), Fp9::to_field(x.c1)}; }
    static ZK_HD Fp29 mul(const Fp29 &a, const Fp29 &b) {
        const Fp9 na1 = Fp9::neg(a.c1);
        const T t[2][2] = {{{&a.c0, &b.c0, false}, {&na1, &b.c1, false}}, {{&a.c0, &b.c1, false}, {&a.c1, &b.c0, false}}};
        const int np[2] = {2, 2};
        Fp9 r[2];
        Fp9::mont_lockstep<2>(t, np, r);
        return Fp29{r[0], r[1]};
    }
    // (a0^2 - a1^2) + (2 a0 a1) u: two symmetric terms (45 products each) and one product against 2 a0 (2-lazy x 1-lazy)
    static ZK_HD Fp29 sqr(const Fp29 &a) {
        const Fp9 na1 = Fp9::neg(a.c1), d0 = Fp9::add(a.c0, a.c0);
        const T t[2][2] = {{{&a.c0, &a.c0, true}, {&na1, &a.c1, true}}, {{&d0, &a.c1, false}, {nullptr, nullptr, false}}};
        const int np[2] = {2, 1};
        Fp9 r[2];
        Fp9::mont_lockstep<2>(t, np, r);
        return Fp29{r[0], r[1]};
    }
    // the tower cannot fuse a*b - c*d (four products per coordinate would overflow a column): two products, a carry propagation
    static ZK_HD Fp29 mulsub(const Fp29 &a, const Fp29 &b, const Fp29 &c, const Fp29 &d) { return norm(sub(mul(a, b), mul(c, d))); }
    static ZK_HD void mul2(Fp29 &r0, const Fp29 &a0, const Fp29 &b0, Fp29 &r1, const Fp29 &a1, const Fp29 &b1) {
        r0 = mul(a0, b0);
        r1 = mul(a1, b1);
    }
    static ZK_HD void sqr2(Fp29 &r0, const Fp29 &a0, Fp29 &r1, const Fp29 &a1) {
        r0 = sqr(a0);
        r1 = sqr(a1);
    }
    static ZK_HD void mul3(Fp29 &r0, const Fp29 &a0, const Fp29 &b0, Fp29 &r1, const Fp29 &a1, const Fp29 &b1, Fp29 &r2, const Fp29 &a2, const Fp29 &b2) {
        r0 = mul(a0, b0);
        r1 = mul(a1, b1);
        r2 = mul(a2, b2);
    }
    static ZK_HD void mulsub_mul(Fp29 &r0, const Fp29 &a, const Fp29 &b, const Fp29 &c, const Fp29 &d, Fp29 &r1, const Fp29 &e, const Fp29 &f) {
        const Fp29 t = mulsub(a, b, c, d);      // r0 may alias c (the accumulator's Y): read everything first
        r1 = mul(e, f);
        r0 = t;
    }
};

}  // namespace zk
