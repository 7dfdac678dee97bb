// Membership of a point of the BN254 twist E'(Fp2) in G2, the subgroup of order r -- gnark-crypto v0.19.2 G2Jac.IsInSubGroup, applied by its point
// decoder to every G2 point of a proving key, a verifying key and a proof (ProvingKey.ReadFrom at gnark/libraries/prover/impl/library.go:104-108;
// verifier/impl: proof.ReadFrom).  The twist has cofactor 2p - r, so "on the curve" is not enough.
//   [r]P = O   <=>   [x+1]P + psi([x]P) + psi^2([x]P) = psi^3([2x]P),      x = the curve seed (63 bits), psi = untwist-Frobenius-twist
// One 63-bit scalar multiplication and three applications of psi instead of a 254-bit scalar multiplication (round 3; the direct [r]P
// test before: 4x the doublings).  Checked against [r]P = O on subgroup points, a random twist point and its cofactor-cleared multiple.
#pragma once
#include "curve.cuh"
#include "frobenius_consts.inc"

namespace zk {

// psi on extended-Jacobian coordinates: conjugation is a field automorphism, so it goes through ZZ / ZZZ unchanged in form
ZK_HD G2XYZZ g2_psi(const G2XYZZ &p) {
    if (p.is_inf()) return p;
    const Fp2 g12 = ZK_FROB_G12, g13 = ZK_FROB_G13;
    auto conj = [](const Fp2 &a) { return Fp2{a.a0, a.a1.neg()}; };
    return G2XYZZ{conj(p.X) * g12, conj(p.Y) * g13, conj(p.ZZ), conj(p.ZZZ)};
}
ZK_HD bool g2_in_subgroup(const G2Affine &p) {
    if (p.is_inf()) return true;
    const uint32_t x[8] = {ZK_BN_X_WORDS_0, ZK_BN_X_WORDS_1, 0, 0, 0, 0, 0, 0};
    G2XYZZ a = scalar_mul(p, x, ZK_BN_X_BITS);      // [x]P
    const G2XYZZ b = g2_psi(a);                      // psi([x]P)
    a.madd(p);                                       // [x+1]P
    G2XYZZ res = g2_psi(b);                          // psi^2([x]P)
    G2XYZZ c = res;
    c.add(b);
    c.add(a);
    res = g2_psi(res).dbl();                         // psi^3([2x]P)
    res.add(c.neg());
    return res.is_inf();
}

}  // namespace zk
