"""ORACLE (test infrastructure only) -- independent pure-Python BN254 pairing and the
Groth16 verification equation.

This is the reference's own end-to-end oracle restated: `verifier.Verify` must accept
what `prover.Prove` produced (gnark/libraries/test_helpers.go:332-339, 403-410;
verifier/impl/verifiers.go:73-89, 142-158 -> groth16.Verify of gnark v0.14.0,
not in the mount).  Deliberately written with a different algorithm from the C code
(python big ints; plain ate pairing with loop length t-1, affine twist arithmetic,
final exponentiation by one big exponent) so that it shares no code or tricks with
either the C oracle or the HIP product.

Any non-degenerate bilinear pairing validates the product equation
    e(Ar, Bs) = e(alpha, beta) * e(sum_pub K_i w_i + commitments, gamma) * e(Krs, delta)
so using the (un-optimised) ate pairing instead of gnark's optimal-ate changes nothing.
"""
from __future__ import annotations

from .gnark_format import P, R, fp2_mul, fp2_sqr, fp2_inv, g1_on_curve, g2_on_curve

XI = (9, 1)
BN_X = 4965661367192848881
ATE_LOOP = 6 * BN_X * BN_X            # t - 1
FINAL_EXP = (P ** 12 - 1) // R


def fp2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def fp2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def fp2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


# Fp12 = Fp2[w]/(w^6 - xi): list of 6 Fp2 coefficients
F12_ONE = [(1, 0)] + [(0, 0)] * 5


def f12_mul(a, b):
    t = [(0, 0)] * 11
    for i in range(6):
        ai = a[i]
        if ai == (0, 0):
            continue
        for j in range(6):
            bj = b[j]
            if bj == (0, 0):
                continue
            t[i + j] = fp2_add(t[i + j], fp2_mul(ai, bj))
    out = list(t[:6])
    for k in range(6, 11):
        out[k - 6] = fp2_add(out[k - 6], fp2_mul(t[k], XI))
    return out


def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_mul(a, a)
        e >>= 1
    return r


def _line(T, lam, Pt):
    """line through twist point T with twist slope lam, evaluated at P in E(Fp):
    y_P - lam*x_P*w + (lam*x_T - y_T)*w^3   (untwist: (x', y') -> (x' w^2, y' w^3))"""
    xp, yp = Pt
    c0 = (yp % P, 0)
    c1 = fp2_scalar(lam, -xp % P)
    c3 = fp2_sub(fp2_mul(lam, T[0]), T[1])
    return [c0, c1, (0, 0), c3, (0, 0), (0, 0)]


def miller_loop(Pt, Q):
    """ate Miller function f_{t-1,Q}(P); P in G1 affine (ints), Q in G2 affine (Fp2 pairs)."""
    if (Pt[0] == 0 and Pt[1] == 0) or (Q[0] == (0, 0) and Q[1] == (0, 0)):
        return F12_ONE
    f = F12_ONE
    T = Q
    bits = bin(ATE_LOOP)[3:]
    for bit in bits:
        lam = fp2_mul(fp2_scalar(fp2_sqr(T[0]), 3), fp2_inv(fp2_scalar(T[1], 2)))
        f = f12_mul(f12_mul(f, f), _line(T, lam, Pt))
        x3 = fp2_sub(fp2_sqr(lam), fp2_scalar(T[0], 2))
        y3 = fp2_sub(fp2_mul(lam, fp2_sub(T[0], x3)), T[1])
        T = (x3, y3)
        if bit == "1":
            if T[0] == Q[0]:
                # T == -Q cannot happen before the last step for prime-order points
                raise ArithmeticError("degenerate addition in miller loop")
            lam = fp2_mul(fp2_sub(T[1], Q[1]), fp2_inv(fp2_sub(T[0], Q[0])))
            f = f12_mul(f, _line(T, lam, Pt))
            x3 = fp2_sub(fp2_sub(fp2_sqr(lam), T[0]), Q[0])
            y3 = fp2_sub(fp2_mul(lam, fp2_sub(T[0], x3)), T[1])
            T = (x3, y3)
    return f


def pairing_product_is_one(pairs) -> bool:
    f = F12_ONE
    for Pt, Q in pairs:
        f = f12_mul(f, miller_loop(Pt, Q))
    return f12_pow(f, FINAL_EXP) == F12_ONE


# ------------------------------------------------------------- plain-python G1 ops
def g1_add(a, b):
    if a == (0, 0):
        return b
    if b == (0, 0):
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return (0, 0)
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def g1_neg(a):
    return (a[0], -a[1] % P) if a != (0, 0) else a


def g1_mul(a, k):
    k %= R
    r = (0, 0)
    while k:
        if k & 1:
            r = g1_add(r, a)
        a = g1_add(a, a)
        k >>= 1
    return r


def expand_message_xmd_py(msg: bytes, dst: bytes, n: int) -> bytes:
    """RFC 9380 section 5.3.1 expand_message_xmd with SHA-256 (pinned to the RFC's appendix K.1 vectors by tests/test_pins.py)."""
    import hashlib
    ell = (n + 31) // 32
    assert ell <= 255 and n <= 65535 and len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    out, bi = b"", bytes(32)
    for i in range(1, ell + 1):
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:n]


def hash_to_fr_py(msg: bytes, dst: bytes) -> int:
    """gnark-crypto fr.Hash(msg, dst, 1)[0]: 48 uniform bytes, big-endian, mod r.  [gnark-mem]"""
    return int.from_bytes(expand_message_xmd_py(msg, dst, 48), "big") % R


def groth16_verify(vk: dict, proof: dict, public_inputs) -> bool:
    """vk: gnark_format.vk_parse layout; proof: gnark_format.proof_parse layout;
    public_inputs: ints, WITHOUT the constant-one wire (gnark public witness)."""
    pts1 = [proof["Ar"], proof["Krs"], proof["pok"]] + list(proof["commitments"])
    if not all(g1_on_curve(p) for p in pts1) or not g2_on_curve(proof["Bs"]):
        return False
    pub = [int(v) % R for v in public_inputs]
    ncom = len(proof["commitments"])
    if ncom != len(vk["commitment_keys"]) or ncom != len(vk["public_and_commitment_committed"]):
        return False
    if len(vk["K"]) != 1 + len(pub) + ncom:
        return False
    # commitment challenges become extra public inputs (gnark groth16 verify.go)
    for i, cm in enumerate(proof["commitments"]):
        ser = cm[0].to_bytes(32, "big") + cm[1].to_bytes(32, "big")
        for j in vk["public_and_commitment_committed"][i]:
            ser += pub[j - 1].to_bytes(32, "big")
        pub.append(hash_to_fr_py(ser, b"bsb22-commitment"))
    if ncom:
        # pedersen PoK: e(C, GSigmaNeg) * e(pok, G) == 1   (single commitment: fold is trivial)
        g, gsn = vk["commitment_keys"][0]
        if ncom != 1:
            raise NotImplementedError("batched pedersen verification")
        if not pairing_product_is_one([(proof["commitments"][0], gsn), (proof["pok"], g)]):
            return False
    ksum = vk["K"][0]
    for k, w in zip(vk["K"][1:], pub):
        ksum = g1_add(ksum, g1_mul(k, w))
    for cm in proof["commitments"]:
        ksum = g1_add(ksum, cm)
    return pairing_product_is_one([
        (proof["Ar"], proof["Bs"]),
        (g1_neg(vk["alpha1"]), vk["beta2"]),
        (g1_neg(ksum), vk["gamma2"]),
        (g1_neg(proof["Krs"]), vk["delta2"]),
    ])
