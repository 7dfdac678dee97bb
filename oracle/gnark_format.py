"""ORACLE (test infrastructure only).

gnark / gnark-crypto binary encodings on the proving path (third-party modules
gnark v0.14.0 + gnark-crypto v0.19.2, gnark/go.mod:6-7, not in the mount):
  * bn254 point (de)compression   -- gnark-crypto ecc/bn254/marshal.go
  * groth16 Proof.WriteTo         -- call sites impl/provers.go:183-188, 273-279
  * groth16 VerifyingKey          -- DECODED against the reference's own
    gnark/libraries/verifier/impl/generated/vk.* (tests/test_oracle_formats.py)
  * groth16 ProvingKey.WriteTo / WriteRawTo -- keygen/keygen.go:99-118 writes it,
    impl/library.go:104-109 reads it.  The reference's pk.* are absent from the
    mount, so this layout is restated from the published gnark source
    ("[gnark-mem]" in SURVEY.md App. C) and is NOT pinned by any artefact.
"""
from __future__ import annotations

import struct

import numpy as np

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
HALF_P = (P - 1) // 2

M_MASK = 0b11 << 6
M_UNCOMPRESSED = 0b00 << 6
M_COMP_INF = 0b01 << 6
M_COMP_SMALLEST = 0b10 << 6
M_COMP_LARGEST = 0b11 << 6


# ----------------------------------------------------------------- Fp / Fp2 helpers
def fp_sqrt(a):
    r = pow(a, (P + 1) // 4, P)
    return r if r * r % P == a % P else None


def fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def fp2_sqr(a):
    return fp2_mul(a, a)


def fp2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, -a[1] * n % P)


def fp2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = fp2_mul(r, a)
        a = fp2_mul(a, a)
        e >>= 1
    return r


def fp2_sqrt(a):
    """sqrt in Fp2 = Fp[u]/(u^2+1), p = 3 mod 4 (complex method)."""
    if a == (0, 0):
        return (0, 0)
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = fp_sqrt(-a0 % P)
        return (0, s) if s is not None else None
    n = fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    inv2 = pow(2, -1, P)
    for nn in (n, -n % P):
        t = (a0 + nn) * inv2 % P
        x0 = fp_sqrt(t)
        if x0 is None or x0 == 0:
            continue
        x1 = a1 * pow(2 * x0, -1, P) % P
        if fp2_sqr((x0, x1)) == (a0 % P, a1 % P):
            return (x0, x1)
    return None


B2 = fp2_mul((3, 0), fp2_inv((9, 1)))     # twist coefficient 3/(9+u)


def fp_lex_largest(y):
    return y > HALF_P


def fp2_lex_largest(y):
    return fp_lex_largest(y[0]) if y[1] == 0 else fp_lex_largest(y[1])


# ----------------------------------------------------------------------- points
def g1_compress(pt) -> bytes:
    x, y = pt
    if x == 0 and y == 0:
        return bytes([M_COMP_INF]) + bytes(31)
    b = bytearray(x.to_bytes(32, "big"))
    b[0] |= M_COMP_LARGEST if fp_lex_largest(y) else M_COMP_SMALLEST
    return bytes(b)


def g1_raw(pt) -> bytes:
    x, y = pt
    return x.to_bytes(32, "big") + y.to_bytes(32, "big")


def g1_decode(buf, pos=0):
    """-> ((x,y), newpos); accepts compressed or uncompressed (flag auto-detect)."""
    flag = buf[pos] & M_MASK
    if flag == M_UNCOMPRESSED:
        x = int.from_bytes(buf[pos:pos + 32], "big")
        y = int.from_bytes(buf[pos + 32:pos + 64], "big")
        return (x, y), pos + 64
    if flag == M_COMP_INF:
        return (0, 0), pos + 32
    b = bytearray(buf[pos:pos + 32])
    b[0] &= ~M_MASK & 0xFF
    x = int.from_bytes(b, "big")
    y = fp_sqrt((x * x * x + 3) % P)
    if y is None:
        raise ValueError("invalid compressed G1 point")
    if fp_lex_largest(y) != (flag == M_COMP_LARGEST):
        y = P - y
    return (x, y), pos + 32


def g2_compress(pt) -> bytes:
    (x0, x1), (y0, y1) = pt
    if x0 == 0 and x1 == 0 and y0 == 0 and y1 == 0:
        return bytes([M_COMP_INF]) + bytes(63)
    b = bytearray(x1.to_bytes(32, "big") + x0.to_bytes(32, "big"))
    b[0] |= M_COMP_LARGEST if fp2_lex_largest((y0, y1)) else M_COMP_SMALLEST
    return bytes(b)


def g2_raw(pt) -> bytes:
    (x0, x1), (y0, y1) = pt
    return x1.to_bytes(32, "big") + x0.to_bytes(32, "big") + y1.to_bytes(32, "big") + y0.to_bytes(32, "big")


def g2_decode(buf, pos=0):
    flag = buf[pos] & M_MASK
    if flag == M_UNCOMPRESSED:
        v = [int.from_bytes(buf[pos + 32 * i:pos + 32 * i + 32], "big") for i in range(4)]
        return ((v[1], v[0]), (v[3], v[2])), pos + 128
    if flag == M_COMP_INF:
        return ((0, 0), (0, 0)), pos + 64
    b = bytearray(buf[pos:pos + 64])
    b[0] &= ~M_MASK & 0xFF
    x1 = int.from_bytes(b[:32], "big")
    x0 = int.from_bytes(b[32:], "big")
    x = (x0, x1)
    x3 = fp2_mul(fp2_sqr(x), x)
    y = fp2_sqrt(((x3[0] + B2[0]) % P, (x3[1] + B2[1]) % P))
    if y is None:
        raise ValueError("invalid compressed G2 point")
    if fp2_lex_largest(y) != (flag == M_COMP_LARGEST):
        y = (-y[0] % P, -y[1] % P)
    return (x, y), pos + 64


def g1_on_curve(pt):
    x, y = pt
    return (x == 0 and y == 0) or (y * y - x * x * x - 3) % P == 0


def g2_on_curve(pt):
    x, y = pt
    if x == (0, 0) and y == (0, 0):
        return True
    x3 = fp2_mul(fp2_sqr(x), x)
    y2 = fp2_sqr(y)
    return ((y2[0] - x3[0] - B2[0]) % P, (y2[1] - x3[1] - B2[1]) % P) == (0, 0)


# ------------------------------------------------------------------------ proof
def proof_bytes(proof: dict, has_commitment: bool) -> bytes:
    """groth16 Proof.WriteTo: Ar | Bs | Krs | u32 n | commitments | pok (all compressed)."""
    out = g1_compress(proof["Ar"]) + g2_compress(proof["Bs"]) + g1_compress(proof["Krs"])
    if has_commitment:
        out += struct.pack(">I", 1) + g1_compress(proof["commitment"])
    else:
        out += struct.pack(">I", 0)
    out += g1_compress(proof["pok"])
    return out


def proof_parse(buf: bytes) -> dict:
    ar, pos = g1_decode(buf, 0)
    bs, pos = g2_decode(buf, pos)
    krs, pos = g1_decode(buf, pos)
    n = struct.unpack_from(">I", buf, pos)[0]
    pos += 4
    com = []
    for _ in range(n):
        c, pos = g1_decode(buf, pos)
        com.append(c)
    pok, pos = g1_decode(buf, pos)
    assert pos == len(buf), (pos, len(buf))
    return dict(Ar=ar, Bs=bs, Krs=krs, commitments=com, pok=pok)


# ---------------------------------------------------------------- verifying key
def vk_parse(buf: bytes) -> dict:
    pos = 0
    alpha1, pos = g1_decode(buf, pos)
    beta1, pos = g1_decode(buf, pos)
    beta2, pos = g2_decode(buf, pos)
    gamma2, pos = g2_decode(buf, pos)
    delta1, pos = g1_decode(buf, pos)
    delta2, pos = g2_decode(buf, pos)
    nk = struct.unpack_from(">I", buf, pos)[0]
    pos += 4
    K = []
    for _ in range(nk):
        k, pos = g1_decode(buf, pos)
        K.append(k)
    n_outer = struct.unpack_from(">I", buf, pos)[0]
    pos += 4
    pacc = []
    for _ in range(n_outer):
        n_in = struct.unpack_from(">I", buf, pos)[0]
        pos += 4
        pacc.append(list(struct.unpack_from(">%dQ" % n_in, buf, pos)))
        pos += 8 * n_in
    nkeys = struct.unpack_from(">I", buf, pos)[0]
    pos += 4
    keys = []
    for _ in range(nkeys):
        g, pos = g2_decode(buf, pos)
        gs, pos = g2_decode(buf, pos)
        keys.append((g, gs))
    assert pos == len(buf), (pos, len(buf))
    return dict(alpha1=alpha1, beta1=beta1, beta2=beta2, gamma2=gamma2, delta1=delta1, delta2=delta2, K=K,
                public_and_commitment_committed=pacc, commitment_keys=keys)


def vk_bytes(vk: dict, raw=False) -> bytes:
    e1 = g1_raw if raw else g1_compress
    e2 = g2_raw if raw else g2_compress
    out = e1(vk["alpha1"]) + e1(vk["beta1"]) + e2(vk["beta2"]) + e2(vk["gamma2"]) + e1(vk["delta1"]) + e2(vk["delta2"])
    out += struct.pack(">I", len(vk["K"])) + b"".join(e1(k) for k in vk["K"])
    out += struct.pack(">I", len(vk["public_and_commitment_committed"]))
    for lst in vk["public_and_commitment_committed"]:
        out += struct.pack(">I", len(lst)) + b"".join(struct.pack(">Q", v) for v in lst)
    out += struct.pack(">I", len(vk["commitment_keys"]))
    for g, gs in vk["commitment_keys"]:
        out += e2(g) + e2(gs)
    return out


# ------------------------------------------------------------------ proving key
def _le_pts_to_int(arr: np.ndarray):
    """(n, 32k) uint8 LE coordinates -> list of k-tuples of ints"""
    n, w = arr.shape
    k = w // 32
    out = []
    for i in range(n):
        row = arr[i].tobytes()
        out.append(tuple(int.from_bytes(row[32 * j:32 * j + 32], "little") for j in range(k)))
    return out


def _g1_raw_block(arr: np.ndarray) -> bytes:
    """vectorised uncompressed encoding of (n,64) LE (x|y) -> n*64 BE bytes"""
    a = arr.reshape(-1, 2, 32)[:, :, ::-1]
    return np.ascontiguousarray(a).tobytes()


def _g2_raw_block(arr: np.ndarray) -> bytes:
    """(n,128) LE (x0|x1|y0|y1) -> BE (x1|x0|y1|y0)"""
    a = arr.reshape(-1, 4, 32)[:, [1, 0, 3, 2], ::-1]
    return np.ascontiguousarray(a).tobytes()


def _bools(flags) -> bytes:
    """gnark-crypto encoder []bool: u32 BE length, then bits packed 8 per byte,
    LSB-first within a byte (buf[i/8] |= 1 << (i%8)).  [gnark-mem, unpinned]"""
    f = np.asarray(flags, dtype=np.uint8)
    return struct.pack(">I", len(f)) + np.packbits(f, bitorder="little").tobytes()


def pk_bytes(pk, raw=True) -> bytes:
    """Serialise an oracle.ref.ProvingKey in gnark's ProvingKey.WriteRawTo / WriteTo layout."""
    N = pk.domain
    r1 = pk.circuit.r1cs
    out = bytearray()
    # fft.Domain.WriteTo: cardinality u64, then CardinalityInv, Generator, GeneratorInv, FrMultiplicativeGen, FrMultiplicativeGenInv (32B BE), withPrecompute bool
    logn = N.bit_length() - 1
    gen = pow(5, (R - 1) >> logn, R)
    out += struct.pack(">Q", N)
    for v in (pow(N, -1, R), gen, pow(gen, -1, R), 5, pow(5, -1, R)):
        out += v.to_bytes(32, "big")
    out += b"\x00"
    g1m = pk.g1(7)
    misc1 = _le_pts_to_int(g1m)
    g2m = _le_pts_to_int(pk.g2_misc())
    to2 = lambda t: ((t[0], t[1]), (t[2], t[3]))
    e1 = g1_raw if raw else g1_compress
    e2 = g2_raw if raw else g2_compress

    def g1_slice(which):
        arr = pk.g1(which)
        hdr = struct.pack(">I", arr.shape[0])
        if raw:
            return hdr + _g1_raw_block(arr)
        return hdr + b"".join(g1_compress(p) for p in _le_pts_to_int(arr))

    out += e1(misc1[0]) + e1(misc1[1]) + e1(misc1[2])
    out += g1_slice(0) + g1_slice(1) + g1_slice(2) + g1_slice(3)
    out += e2(to2(g2m[0])) + e2(to2(g2m[1]))
    b2 = pk.g2_B()
    out += struct.pack(">I", b2.shape[0])
    out += _g2_raw_block(b2) if raw else b"".join(g2_compress(to2(p)) for p in _le_pts_to_int(b2))
    infa, infb = pk.infinity_flags()
    out += struct.pack(">Q", r1.nb_wires) + struct.pack(">Q", int(infa.sum())) + struct.pack(">Q", int(infb.sum()))
    out += _bools(infa) + _bools(infb)
    nped = pk.count(5)
    out += struct.pack(">I", 1 if nped else 0)
    if nped:   # pedersen.ProvingKey.WriteTo: Basis slice, BasisExpSigma slice
        out += g1_slice(5) + g1_slice(6)
    return bytes(out)


def vk_from_pk(pk) -> dict:
    misc1 = _le_pts_to_int(pk.g1(7))
    g2m = _le_pts_to_int(pk.g2_misc())
    to2 = lambda t: ((t[0], t[1]), (t[2], t[3]))
    nped = pk.count(5)
    return dict(alpha1=misc1[0], beta1=misc1[1], delta1=misc1[2], beta2=to2(g2m[0]), delta2=to2(g2m[1]), gamma2=to2(g2m[2]),
                K=_le_pts_to_int(pk.g1(4)),
                public_and_commitment_committed=[[]] if nped else [],
                commitment_keys=[(to2(g2m[3]), to2(g2m[4]))] if nped else [])
