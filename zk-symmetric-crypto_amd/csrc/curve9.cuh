// XYZZ accumulators over the 9 x 29-bit field (field9.cuh): the mixed / general additions of the bucket accumulation and reduction.
// Same formulas, special cases and results (mod M) as XYZZ<F> in curve.cuh -- madd-2008-s / add-2008-s / dbl-2008-s-1, replacing
// gnark-crypto v0.19.2 g1JacExtended.addMixed / add / double --, coordinates in the 2^261-Montgomery form.
// Per mixed addition: 6 products, 2 squarings and ONE fused pair (Y3 = R*(Q - X3) - Y1*PPP shares a reduction) = 1 467 multiply-adds,
// three carry-free subtractions and one carry propagation (X3, which becomes state); independent products run in lockstep.  The
// 8 x 32 form needs 1 280 multiply-adds + 1 280 carry instructions + seven ~28-instruction modular additions.
#pragma once
#include "curve.cuh"
#include "field9.cuh"

namespace zk {

template <class F9>
struct alignas(16) XYZZ9 {         // 144 bytes: bucket sums travel between the accumulation and the reduction as the raw strict limbs
    using W = typename F9::Words;
    F9 X, Y, ZZ, ZZZ;      // strict between operations; infinity = all limbs of ZZ zero

    // A finite point's ZZ is a product of values that are non-zero mod M, so its strict integer is neither 0 nor M: only the
    // all-zero limbs written by inf() mean infinity, and the test is exact.
    ZK_HD bool is_inf() const { return ZZ.is_zero_exact(); }
    static ZK_HD XYZZ9 inf() { return XYZZ9{F9::zero(), F9::zero(), F9::zero(), F9::zero()}; }

    // 2 * (x, y), affine, strict inputs (mdbl-2008-s-1).  Rare path: every lazy intermediate is normalised before it is multiplied.
    static __host__ __device__ __noinline__ XYZZ9 dbl_affine(F9 x, F9 y) {
        const F9 U = F9::norm(F9::add(y, y)), V = F9::mul(U, U), Wv = F9::mul(U, V), S = F9::mul(x, V);
        const F9 t = F9::mul(x, x);
        const F9 Mm = F9::norm(F9::add(F9::add(t, t), t));
        XYZZ9 r;
        r.X = F9::norm(F9::sub(F9::sub(F9::mul(Mm, Mm), S), S));
        r.Y = F9::mulsub(Mm, F9::norm(F9::sub(S, r.X)), Wv, y);
        r.ZZ = V;
        r.ZZZ = Wv;
        return r;
    }
    // The out-of-line helpers of the rare paths take and return VALUES: a member function called through `this` would pin the
    // caller's accumulator in scratch memory for the whole walk.
    // 2 * p (dbl-2008-s-1), rare path
    static __host__ __device__ __noinline__ XYZZ9 dbl_of(XYZZ9 p) {
        if (p.is_inf()) return p;
        const F9 U = F9::norm(F9::add(p.Y, p.Y)), V = F9::mul(U, U), Wv = F9::mul(U, V), S = F9::mul(p.X, V);
        const F9 t = F9::mul(p.X, p.X);
        const F9 Mm = F9::norm(F9::add(F9::add(t, t), t));
        XYZZ9 r;
        r.X = F9::norm(F9::sub(F9::sub(F9::mul(Mm, Mm), S), S));
        r.Y = F9::mulsub(Mm, F9::norm(F9::sub(S, r.X)), Wv, p.Y);
        r.ZZ = F9::mul(V, p.ZZ);
        r.ZZZ = F9::mul(Wv, p.ZZZ);
        return r;
    }
    ZK_HD XYZZ9 dbl() const { return dbl_of(*this); }
    // 2^m * p for a RUN of doublings (the Horner chain of the comb fold: 8 between two additions, 254 dependent ones per proof).
    // Through Jacobian coordinates: (X, Y, ZZ, ZZZ) = (X ZZ, Y ZZZ, ZZ) since ZZZ^2 = ZZ^3, m times dbl-2009-l with the factors of
    // two moved INSIDE the products (D = X * 4B, 8C = 4B * 2B, so X3 = E^2 - 2D and Y3 = E (D - X3) - 8C come out of their
    // reductions in the ranges field9.cuh states for X and Y), back with ZZ = Z^2, ZZZ = Z^3.  Per doubling 6 reductions over 7
    // products (dbl-2008-s-1 above: 8 over 9) plus 4 products per run.  The same point, another representative of it: callers
    // compare affine coordinates.  Bounds: A, B, D, F, C8 in (-0.14 M, 1.3 M); E = 3A, B4 = 4B strict after norm, |.| < 4.6 M;
    // X3 in (-2.8 M, 1.2 M); T = D - X3 1-lazy; Y3 = E T - B4 B2 fused, in (-0.1 M, 1.1 M); Z stays a plain product.
    static __host__ __device__ __noinline__ XYZZ9 dbl_many(XYZZ9 p, int m) {
        if (p.is_inf() || m <= 0) return p;
        F9 X = F9::mul(p.X, p.ZZ), Y = F9::mul(p.Y, p.ZZZ), Z = p.ZZ;
        for (int i = 0; i < m; i++) {
            const F9 A = F9::sqr(X), Bv = F9::sqr(Y);
            const F9 B2 = F9::norm(F9::add(Bv, Bv)), B4 = F9::norm(F9::add(B2, B2));
            const F9 D = F9::mul(X, B4);
            const F9 E = F9::norm(F9::add(F9::add(A, A), A));
            const F9 Fv = F9::sqr(E);
            const F9 X3 = F9::norm(F9::sub(F9::sub(Fv, D), D));
            const F9 Y3 = F9::mulsub(E, F9::norm(F9::sub(D, X3)), B4, B2);
            Z = F9::mul(F9::norm(F9::add(Y, Y)), Z);
            X = X3;
            Y = Y3;
        }
        XYZZ9 r;
        r.X = X;
        r.Y = Y;
        r.ZZ = F9::sqr(Z);
        r.ZZZ = F9::mul(r.ZZ, Z);
        return r;
    }
    // P == 0 mod M in a mixed addition: doubling (R == 0) or cancellation
    static __host__ __device__ __noinline__ XYZZ9 madd_special(F9 x2, F9 y2, F9 R) {
        if (F9::sqr(R).is_zero_mod()) return dbl_affine(x2, y2);
        return inf();
    }
    // this += (x2, y2): affine point that is NOT infinity; coordinates strict, or y2 the limb-wise negation of a strict value (a
    // negated base point: it only enters the product y2 * ZZZ, whose result is strict again).  Limb bounds: Pv, R = strict - strict
    // (1-lazy); X3 before its carry propagation is 4-lazy (it becomes state and an operand, hence normalised); T = Q - X3 is 1-lazy;
    // the fused Y3 takes (1-lazy x 1-lazy) + (strict x strict).
    // (Deciding the special cases only AFTER the arithmetic -- an empty accumulator computes zeros, P^2 == 0 flags it -- keeps the
    // products in one block but holds old and new state live together: spills at 128 VGPRs (84 ms), 73.8 ms at 3 wavefronts per
    // SIMD against 72.0 for this form.)
    ZK_HD void madd(const F9 &x2, const F9 &y2) {
        if (is_inf()) {
            X = x2; Y = F9::norm(y2); ZZ = F9::one(); ZZZ = F9::one();
            return;
        }
        F9 U2, S2, PP, RR, PPP, Q, ZZ3;
        F9::mul2(U2, x2, ZZ, S2, y2, ZZZ);
        const F9 Pv = F9::sub(U2, X), R = F9::sub(S2, Y);
        F9::sqr2(PP, Pv, RR, R);
        if (PP.is_zero_mod()) {          // M is prime: P^2 == 0 <=> P == 0
            *this = madd_special(x2, F9::norm(y2), R);
            return;
        }
        F9::mul3(PPP, Pv, PP, Q, X, PP, ZZ3, ZZ, PP);
        const F9 X3 = F9::x3(RR, PPP, Q);
        F9::mulsub_mul(Y, R, F9::sub(Q, X3), Y, PPP, ZZZ, ZZZ, PPP);
        X = X3;
        ZZ = ZZ3;
    }
    // this += q (add-2008-s); both in the 2^261 form, strict
    ZK_HD void add(const XYZZ9 &q) {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }
        F9 U1, U2, S1, S2, PP, RR, PPP, Q, Za, Zb;
        F9::mul2(U1, X, q.ZZ, U2, q.X, ZZ);
        F9::mul3(S1, Y, q.ZZZ, S2, q.Y, ZZZ, Za, ZZ, q.ZZ);
        const F9 Pv = F9::sub(U2, U1), R = F9::sub(S2, S1);
        F9::sqr2(PP, Pv, RR, R);
        if (PP.is_zero_mod()) {
            const XYZZ9 self = *this;
            if (RR.is_zero_mod()) *this = dbl_of(self);
            else *this = inf();
            return;
        }
        F9::mul3(PPP, Pv, PP, Q, U1, PP, Zb, ZZZ, q.ZZZ);
        const F9 X3 = F9::x3(RR, PPP, Q);
        F9::mulsub_mul(Y, R, F9::sub(Q, X3), S1, PPP, ZZ, Za, PP);
        X = X3;
        ZZZ = F9::mul(Zb, PPP);
    }
    // canonical 2^256-Montgomery words (what the rest of the pipeline reads); infinity -> all zero
    static __host__ __device__ __noinline__ XYZZ<W> to_xyzz_of(XYZZ9 p) {
        if (p.is_inf()) return XYZZ<W>::inf();
        return XYZZ<W>{F9::to_field(p.X), F9::to_field(p.Y), F9::to_field(p.ZZ), F9::to_field(p.ZZZ)};
    }
    ZK_HD XYZZ<W> to_xyzz() const { return to_xyzz_of(*this); }
    // canonical words of the 2^261 form (bucket sums that the 9-limb reduction reads back)
    ZK_HD XYZZ<W> to_words() const {
        if (is_inf()) return XYZZ<W>::inf();
        // (to_words reduces into [0, M): a finite point keeps ZZ != 0, so it cannot read back as infinity)
        return XYZZ<W>{F9::to_words(X), F9::to_words(Y), F9::to_words(ZZ), F9::to_words(ZZZ)};
    }
    static ZK_HD XYZZ9 from_words(const XYZZ<W> &p) {
        return XYZZ9{F9::from_words(p.X), F9::from_words(p.Y), F9::from_words(p.ZZ), F9::from_words(p.ZZZ)};
    }
};
using G1X9 = XYZZ9<Fp9>;
using G2X9 = XYZZ9<Fp29>;

// 9-limb counterparts of the coordinate fields
template <class F>
struct Nine;
template <>
struct Nine<Fp> { using F9 = Fp9; using X9 = G1X9; };
template <>
struct Nine<Fp2> { using F9 = Fp29; using X9 = G2X9; };

}  // namespace zk
