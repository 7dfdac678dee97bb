// BN254 G1 (over Fp) / G2 (over Fp2) group arithmetic for gfx950, extended-Jacobian ("XYZZ")
// accumulators with affine addends: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.
//
// Device replacement for the bucket arithmetic of gnark-crypto v0.19.2 ecc/bn254 (g1JacExtended /
// g2JacExtended, used by MultiExp) reached from groth16.Prove, provers.go:179,269.
// Infinity: affine (0,0) (gnark-crypto convention); XYZZ with ZZ == 0.
#pragma once
#include "field.cuh"

namespace zk {

// multiplication used inside the mixed addition; INL expands the Fp multiplier block in place (innermost loop of the bucket
// accumulation: saves the 24 register moves of a call per multiplication), Fp2 always keeps the calls
template <bool INL, class F>
ZK_HD F madd_mul(const F &a, const F &b) {
    if constexpr (INL && sizeof(F) == 32) return F::mul_inline(a, b);
    else return a * b;
}

template <class F>
struct Affine {
    F x, y;
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static ZK_HD Affine inf() { return Affine{F::zero(), F::zero()}; }
    ZK_HD Affine neg() const { return Affine{x, y.neg()}; }
};

template <class F>
struct XYZZ {
    F X, Y, ZZ, ZZZ;
    ZK_HD bool is_inf() const { return ZZ.is_zero(); }
    static ZK_HD XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
    static ZK_HD XYZZ from_affine(const Affine<F> &p) {
        if (p.is_inf()) return inf();
        return XYZZ{p.x, p.y, F::one(), F::one()};
    }
    ZK_HD XYZZ neg() const { return XYZZ{X, Y.neg(), ZZ, ZZZ}; }

    // 2*p for affine p (mdbl-2008-s-1)
    static ZK_HD XYZZ dbl_affine(const Affine<F> &p) {
        F U = p.y.dbl(), V = U.sqr(), W = U * V, S = p.x * V;
        F t = p.x.sqr();
        F M = t.dbl() + t;
        XYZZ r;
        r.X = M.sqr() - S.dbl();
        r.Y = M * (S - r.X) - W * p.y;
        r.ZZ = V;
        r.ZZZ = W;
        return r;
    }
    // dbl-2008-s-1
    ZK_HD XYZZ dbl() const {
        if (is_inf()) return *this;
        F U = Y.dbl(), V = U.sqr(), W = U * V, S = X * V;
        F t = X.sqr();
        F M = t.dbl() + t;
        XYZZ r;
        r.X = M.sqr() - S.dbl();
        r.Y = M * (S - r.X) - W * Y;
        r.ZZ = V * ZZ;
        r.ZZZ = W * ZZZ;
        return r;
    }
    // this += q (affine), madd-2008-s with the special cases
    template <bool INL = false>
    ZK_HD void madd(const Affine<F> &q) {
        if (q.is_inf()) return;
        if (is_inf()) {
            X = q.x; Y = q.y; ZZ = F::one(); ZZZ = F::one();
            return;
        }
        F U2 = madd_mul<INL>(q.x, ZZ), S2 = madd_mul<INL>(q.y, ZZZ);
        F Pv = U2 - X, R = S2 - Y;
        if (Pv.is_zero()) {
            if (R.is_zero()) *this = dbl_affine(q);
            else *this = inf();
            return;
        }
        F PP = madd_mul<INL>(Pv, Pv), PPP = madd_mul<INL>(Pv, PP), Q = madd_mul<INL>(X, PP);
        F X3 = madd_mul<INL>(R, R) - PPP - Q.dbl();
        Y = madd_mul<INL>(R, Q - X3) - madd_mul<INL>(Y, PPP);
        X = X3;
        ZZ = madd_mul<INL>(ZZ, PP);
        ZZZ = madd_mul<INL>(ZZZ, PPP);
    }
    // this += q with the accumulator kept in WEAK form (coordinates in [0, 2M), field.cuh).  Same formulas and special cases as
    // madd / add; q may be canonical or weak; the caller stores canon().  INL expands the multiplier block in place (innermost
    // loop of the G1 bucket accumulation), otherwise the shared out-of-line weak multiplier is called.
    // (Fp2 offers the same static interface with its coordinates weak; INL is ignored there.)
    template <bool INL>
    static ZK_HD F wmul(const F &a, const F &b) {
        if constexpr (INL) return F::mul_w(a, b);
        else return F::mul_wc(a, b);
    }
    template <bool INL = true>
    ZK_HD void madd_w(const Affine<F> &q) {
        if (q.is_inf()) return;
        if (is_inf()) {                      // infinity is the exact all-zero encoding; a finite accumulator never has ZZ = 0 mod M
            X = q.x; Y = q.y; ZZ = F::one(); ZZZ = F::one();
            return;
        }
        const F U2 = wmul<INL>(q.x, ZZ), S2 = wmul<INL>(q.y, ZZZ);
        const F Pv = F::sub_w(U2, X), R = F::sub_w(S2, Y);
        if (Pv.is_zero_w()) {
            if (R.is_zero_w()) *this = dbl_affine(q);
            else *this = inf();
            return;
        }
        const F PP = wmul<INL>(Pv, Pv), PPP = wmul<INL>(Pv, PP), Q = wmul<INL>(X, PP);
        const F X3 = F::sub_w(F::sub_w(wmul<INL>(R, R), PPP), F::add_w(Q, Q));
        Y = F::sub_w(wmul<INL>(R, F::sub_w(Q, X3)), wmul<INL>(Y, PPP));
        X = X3;
        ZZ = wmul<INL>(ZZ, PP);
        ZZZ = wmul<INL>(ZZZ, PPP);
    }
    // this += q (XYZZ, canonical or weak), accumulator weak
    ZK_HD void add_w(const XYZZ &q) {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }
        const F U1 = F::mul_wc(X, q.ZZ), U2 = F::mul_wc(q.X, ZZ), S1 = F::mul_wc(Y, q.ZZZ), S2 = F::mul_wc(q.Y, ZZZ);
        const F Pv = F::sub_w(U2, U1), R = F::sub_w(S2, S1);
        if (Pv.is_zero_w()) {
            if (R.is_zero_w()) *this = canon().dbl();
            else *this = inf();
            return;
        }
        const F PP = F::mul_wc(Pv, Pv), PPP = F::mul_wc(Pv, PP), Q = F::mul_wc(U1, PP);
        const F X3 = F::sub_w(F::sub_w(F::mul_wc(R, R), PPP), F::add_w(Q, Q));
        Y = F::sub_w(F::mul_wc(R, F::sub_w(Q, X3)), F::mul_wc(S1, PPP));
        X = X3;
        ZZ = F::mul_wc(F::mul_wc(ZZ, q.ZZ), PP);
        ZZZ = F::mul_wc(F::mul_wc(ZZZ, q.ZZZ), PPP);
    }
    ZK_HD XYZZ canon() const { return XYZZ{X.canon(), Y.canon(), ZZ.canon(), ZZZ.canon()}; }
    // this += q (XYZZ), add-2008-s
    ZK_HD void add(const XYZZ &q) {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }
        F U1 = X * q.ZZ, U2 = q.X * ZZ, S1 = Y * q.ZZZ, S2 = q.Y * ZZZ;
        F Pv = U2 - U1, R = S2 - S1;
        if (Pv.is_zero()) {
            if (R.is_zero()) *this = dbl();
            else *this = inf();
            return;
        }
        F PP = Pv.sqr(), PPP = Pv * PP, Q = U1 * PP;
        F X3 = R.sqr() - PPP - Q.dbl();
        Y = R * (Q - X3) - S1 * PPP;
        X = X3;
        ZZ = ZZ * q.ZZ * PP;
        ZZZ = ZZZ * q.ZZZ * PPP;
    }
    ZK_HD Affine<F> to_affine() const {
        if (is_inf()) return Affine<F>::inf();
        F zi3 = ZZZ.inv();
        F zi2 = zi3.sqr() * ZZ.sqr();   // 1/ZZ = ZZ^2 / ZZZ^2
        return Affine<F>{X * zi2, Y * zi3};
    }
};

// k * p, k = plain 256-bit integer (8 LE limbs); MSB-first double-and-add
template <class F>
ZK_HD XYZZ<F> scalar_mul(const Affine<F> &p, const uint32_t *k, int nbits = 254) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = acc.dbl();
        if ((k[i >> 5] >> (i & 31)) & 1) acc.madd(p);
    }
    return acc;
}

// same with a projective base point (key folding at InitAlgorithm, keyfold.hip)
template <class F>
ZK_HD XYZZ<F> scalar_mul_x(const XYZZ<F> &p, const uint32_t *k, int nbits = 254) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = acc.dbl();
        if ((k[i >> 5] >> (i & 31)) & 1) acc.add(p);
    }
    return acc;
}

using G1Affine = Affine<Fp>;
using G2Affine = Affine<Fp2>;
using G1XYZZ = XYZZ<Fp>;
using G2XYZZ = XYZZ<Fp2>;

}  // namespace zk
