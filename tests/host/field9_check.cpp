// Host-side check of the 9 x 29-bit field and its XYZZ arithmetic (csrc/field9.cuh, csrc/curve9.cuh) against the 8 x 32 field
// (csrc/field.cuh, itself pinned to the oracle by the GPU parity tests): the host build of both headers runs the same C++ with
// plain int64 arithmetic in place of v_mad_i64_i32.  Prints "ok" or the first mismatches; tests/test_field9_host.py runs it.
#include "curve9.cuh"
#include <stdio.h>
#include <random>
#include <vector>
using namespace zk;

static Fp rnd(std::mt19937_64 &rng, int kind) {
    Fp r;
    for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)rng();
    if (kind == 1) for (int i = 0; i < 8; i++) r.l[i] = 0xffffffffu;
    if (kind == 2) for (int i = 0; i < 8; i++) r.l[i] = 0;
    if (kind == 3) { for (int i = 0; i < 8; i++) r.l[i] = FpParams::M[i]; r.l[0] -= 1; }
    if (kind == 4) { for (int i = 1; i < 8; i++) r.l[i] = 0; r.l[0] = 1; }
    r.l[7] &= 0x3fffffffu;
    while (Fp::geq_mod(r.l)) Fp::reduce_once(r.l);
    return r;
}
int main() {
    std::mt19937_64 rng(7);
    int bad = 0;
    // field: products, squares, fused pairs with lazy operands, conversions
    for (int it = 0; it < 20000; it++) {
        const Fp a = rnd(rng, it % 7), b = rnd(rng, (it / 7) % 7), c = rnd(rng, 0), d = rnd(rng, (it / 49) % 7);
        const Fp9 A = Fp9::from_field(a), B = Fp9::from_field(b), C = Fp9::from_field(c), D = Fp9::from_field(d);
        const Fp9 u = Fp9::sub(A, C), v = Fp9::sub(B, D);       // 1-lazy, possibly negative
        bool ok = Fp9::to_field(A) == a;
        ok &= Fp9::to_field(Fp9::mul(u, v)) == (a - c) * (b - d);
        ok &= Fp9::to_field(Fp9::sqr(u)) == (a - c) * (a - c);
        ok &= Fp9::to_field(Fp9::mulsub(u, B, C, v)) == (a - c) * b - c * (b - d);
        ok &= Fp9::to_field(Fp9::muladd(u, v, C, D)) == (a - c) * (b - d) + c * d;
        Fp9 r0, r1, r2;
        Fp9::mul3(r0, A, B, r1, u, D, r2, C, v);
        ok &= Fp9::to_field(r0) == a * b && Fp9::to_field(r1) == (a - c) * d && Fp9::to_field(r2) == c * (b - d);
        Fp9::sqr2(r0, v, r1, A);
        ok &= Fp9::to_field(r0) == (b - d) * (b - d) && Fp9::to_field(r1) == a * a;
        Fp9::mulsub_mul(r0, A, v, u, D, r1, C, C);
        ok &= Fp9::to_field(r0) == a * (b - d) - (a - c) * d && Fp9::to_field(r1) == c * c;
        // to_words over its whole input range (-4M, 4M): sums / differences of up to 3 strict values
        const Fp9 w3 = Fp9::norm(Fp9::add(Fp9::add(A, B), C)), wn = Fp9::norm(Fp9::sub(Fp9::sub(Fp9::neg(A), B), C));
        ok &= Fp9::to_field(w3) == a + b + c && Fp9::to_field(wn) == (a + b + c).neg();
        ok &= Fp9::mul(u, Fp9::zero()).is_zero_mod() && (a == c) == Fp9::sqr(u).is_zero_mod();
        if (!ok && bad++ < 5) printf("field mismatch at iteration %d\n", it);
    }
    // the extremes of the value model (field9.cuh header): the state X of an accumulator at its lower edge (-3.5 M), U2 at its
    // upper edge (9M/8): P = U2 - X = 4.625 M, P^2 = 21.4 M^2 -- the largest sum of products any call site forms -- must still reduce
    // exactly, and to_words must take X itself (its input range is (-4M, 4M))
    for (int it = 0; it < 2000; it++) {
        Fp small = rnd(rng, 0), small2 = rnd(rng, 0);
        small.l[7] = 0; small2.l[7] = 0;                                                  // < 2^224: "a little above the edge"
        Fp half, eighth;
        for (int i = 0; i < 8; i++) {
            half.l[i] = (FpParams::M[i] >> 1) | (i < 7 ? FpParams::M[i + 1] << 31 : 0);       // (M - 1) / 2
            eighth.l[i] = (FpParams::M[i] >> 3) | (i < 7 ? FpParams::M[i + 1] << 29 : 0);     // floor(M / 8)
        }
        const Fp9 S9 = Fp9::from_words(small), T9 = Fp9::from_words(small2);
        const Fp9 Xe = Fp9::norm(Fp9::add(Fp9::sub(S9, Fp9::constant(Fp9Params::M4)), Fp9::from_words(half)));            // small - 4M + (M-1)/2  ~ -3.5 M
        const Fp9 Ue = Fp9::norm(Fp9::sub(Fp9::add(Fp9::constant(Fp9Params::M), Fp9::from_words(eighth)), T9));            // M + M/8 - small2  ~ 9M/8
        // the same residues on the 8 x 32 field (values in the 2^261 form on both sides: compare through to_words)
        const Fp xw = small + half, uw = eighth - small2;
        bool ok = Fp9::to_words(Xe) == xw && Fp9::to_words(Ue) == uw;
        const Fp9 Pv = Fp9::sub(Ue, Xe);                                                                                    // 1-lazy, ~ 4.625 M
        // products of the 2^261 form: compare with the 2^256 form through the conversions (x9 = X * 2^5 as integers mod M)
        const Fp xf = Fp9::to_field(Xe), uf = Fp9::to_field(Ue);
        ok &= Fp9::to_field(Fp9::sqr(Pv)) == (uf - xf) * (uf - xf);
        ok &= Fp9::to_field(Fp9::mul(Pv, Xe)) == (uf - xf) * xf;
        ok &= Fp9::to_field(Fp9::mulsub(Pv, Pv, Xe, Ue)) == (uf - xf) * (uf - xf) - xf * uf;
        Fp9 r0, r1, r2;
        Fp9::mul3(r0, Pv, Pv, r1, Xe, Pv, r2, Ue, Pv);
        ok &= Fp9::to_field(r0) == (uf - xf) * (uf - xf) && Fp9::to_field(r1) == xf * (uf - xf) && Fp9::to_field(r2) == uf * (uf - xf);
        ok &= Fp9::to_field(Fp9::x3(r0, r1, r2)) == (uf - xf) * (uf - xf) - xf * (uf - xf) - (uf * (uf - xf)).dbl();
        if (!ok && bad++ < 5) printf("extreme-value mismatch at iteration %d\n", it);
    }
    // curve: chains of mixed additions with negated points, repeated points (doubling), cancellations, then general additions
    G1Affine g{Fp::one(), Fp::one() + Fp::one()};
    std::vector<G1Affine> pts;
    for (int i = 0; i < 48; i++) { uint32_t k[8] = {(uint32_t)rng(), (uint32_t)rng(), 0, 0, 0, 0, 0, 0}; pts.push_back(scalar_mul(g, k, 64).to_affine()); }
    Fp k32;
    for (int j = 0; j < 8; j++) k32.l[j] = Fp9Params::SCALE32[j];
    auto chain = [&](int trial, int len, G1XYZZ &a, G1X9 &b) {
        a = G1XYZZ::inf(); b = G1X9::inf();
        for (int s = 0; s < len; s++) {
            G1Affine p = pts[rng() % 48];
            bool neg = rng() & 1;
            if (trial % 5 == 3 && s == 2) p = pts[0];
            if (trial % 5 == 3 && s == 3) { p = pts[0]; neg = (trial / 5) & 1; }        // twice the same point in a row: doubling or cancellation
            G1Affine q = p;
            if (neg) q.y = q.y.neg();
            a.madd(q);
            b.madd(Fp9::from_words(p.x * k32), Fp9::cneg(Fp9::from_words(p.y * k32), neg ? -1 : 0));
        }
    };
    auto same = [](const G1XYZZ &a, const G1X9 &b) {
        const G1XYZZ r = b.to_xyzz();
        return r.X == a.X && r.Y == a.Y && r.ZZ == a.ZZ && r.ZZZ == a.ZZZ;
    };
    for (int trial = 0; trial < 300; trial++) {
        G1XYZZ a1, a2; G1X9 b1, b2;
        chain(trial, 1 + rng() % 24, a1, b1);
        chain(trial + 1, trial % 11 == 0 ? 0 : 1 + rng() % 24, a2, b2);
        bool ok = same(a1, b1) && same(a2, b2);
        // record format of the bucket sums: raw limbs, and the canonical 2^261 words
        ok &= same(a1, G1X9::from_words(b1.to_words()));
        G1XYZZ s = a1; s.add(a2);
        G1X9 t = b1; t.add(b2);
        ok &= same(s, t);
        G1XYZZ d = a1; d.add(a1);                  // P + P through the general addition
        G1X9 e = b1; e.add(b1);
        ok &= same(d, e);
        G1XYZZ z = a1; z.add(a1.neg());            // P - P
        G1X9 y = b1; G1X9 nb = b1; nb.Y = Fp9::norm(Fp9::neg(b1.Y)); y.add(nb);
        ok &= z.is_inf() && y.is_inf();
        ok &= same(a1.dbl(), b1.dbl());
        {   // a run of doublings through Jacobian coordinates: another representative of the same point, and one that the record formats take
            const int m = 1 + trial % 9;
            G1XYZZ dd = a1;
            for (int i = 0; i < m; i++) dd = dd.dbl();
            const G1X9 dm = G1X9::dbl_many(b1, m);
            const G1Affine want = dd.to_affine(), got = dm.to_xyzz().to_affine(), got2 = G1X9::from_words(dm.to_words()).to_xyzz().to_affine();
            ok &= dd.is_inf() == dm.is_inf() && (dd.is_inf() || (want.x == got.x && want.y == got.y && want.x == got2.x && want.y == got2.y));
            G1X9 fm = dm; fm.add(b2);              // ... and one that the addition formulas take as state
            G1XYZZ fd = dd; fd.add(a2);
            if (!fd.is_inf() && !dd.is_inf()) { const G1Affine w2 = fd.to_affine(), g2 = fm.to_xyzz().to_affine(); ok &= w2.x == g2.x && w2.y == g2.y; }
        }
        if (!ok && bad++ < 10) printf("curve mismatch in trial %d\n", trial);
    }
    // the quadratic extension and G2: the formulas are identities in the coordinates, so arbitrary (x, y) pairs exercise them
    for (int it = 0; it < 4000; it++) {
        const Fp2 a{rnd(rng, it % 7), rnd(rng, (it / 7) % 7)}, b{rnd(rng, 0), rnd(rng, (it / 49) % 7)}, c{rnd(rng, 0), rnd(rng, 0)};
        const Fp29 A = Fp29::from_field(a), B = Fp29::from_field(b), C = Fp29::from_field(c);
        const Fp29 u = Fp29::sub(A, C);                         // 1-lazy
        bool ok = Fp29::to_field(A) == a;
        ok &= Fp29::to_field(Fp29::mul(u, B)) == (a - c) * b;
        ok &= Fp29::to_field(Fp29::mul(u, Fp29::sub(B, C))) == (a - c) * (b - c);
        ok &= Fp29::to_field(Fp29::sqr(u)) == (a - c).sqr();
        ok &= Fp29::to_field(Fp29::mulsub(u, B, C, Fp29::sub(B, A))) == (a - c) * b - c * (b - a);
        ok &= Fp29::to_field(Fp29::x3(A, B, C)) == a - b - c.dbl();
        if (!ok && bad++ < 5) printf("Fp2 mismatch at iteration %d\n", it);
    }
    {
        std::vector<G2Affine> q2;
        for (int i = 0; i < 24; i++) q2.push_back(G2Affine{Fp2{rnd(rng, 0), rnd(rng, 0)}, Fp2{rnd(rng, 0), rnd(rng, 0)}});
        const Fp2 k2{k32, Fp::zero()};
        auto chain2 = [&](int trial, int len, G2XYZZ &a, G2X9 &b) {
            a = G2XYZZ::inf(); b = G2X9::inf();
            for (int s = 0; s < len; s++) {
                G2Affine p = q2[rng() % 24];
                bool neg = rng() & 1;
                if (trial % 5 == 3 && s == 2) p = q2[0];
                if (trial % 5 == 3 && s == 3) { p = q2[0]; neg = (trial / 5) & 1; }
                G2Affine q = p;
                if (neg) q.y = q.y.neg();
                a.madd(q);
                b.madd(Fp29::from_words(p.x * k2), Fp29::cneg(Fp29::from_words(p.y * k2), neg ? -1 : 0));
            }
        };
        auto same2 = [](const G2XYZZ &a, const G2X9 &b) {
            const G2XYZZ r = b.to_xyzz();
            return r.X == a.X && r.Y == a.Y && r.ZZ == a.ZZ && r.ZZZ == a.ZZZ;
        };
        for (int trial = 0; trial < 120; trial++) {
            G2XYZZ a1, a2; G2X9 b1, b2;
            chain2(trial, 1 + rng() % 16, a1, b1);
            chain2(trial + 1, trial % 11 == 0 ? 0 : 1 + rng() % 16, a2, b2);
            bool ok = same2(a1, b1) && same2(a2, b2) && same2(a1, G2X9::from_words(b1.to_words()));
            G2XYZZ s2 = a1; s2.add(a2);
            G2X9 t2 = b1; t2.add(b2);
            ok &= same2(s2, t2);
            G2XYZZ d = a1; d.add(a1);
            G2X9 e = b1; e.add(b1);
            ok &= same2(d, e) && same2(a1.dbl(), b1.dbl());
            {   // (arbitrary (x, y) pairs are not curve points: compare the cross-multiplied coordinates, which are identities of the formulas)
                const int m = 1 + trial % 9;
                G2XYZZ dd = a1;
                for (int i = 0; i < m; i++) dd = dd.dbl();
                const G2XYZZ dm = G2X9::dbl_many(b1, m).to_xyzz();
                ok &= dd.is_inf() == dm.is_inf() && dd.X * dm.ZZ == dm.X * dd.ZZ && dd.Y * dm.ZZZ == dm.Y * dd.ZZZ;
            }
            if (!ok && bad++ < 10) printf("G2 mismatch in trial %d\n", trial);
        }
    }
    printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad != 0;
}
