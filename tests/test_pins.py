"""Pins of the restated gnark conventions against data that gnark (or a public standard) made -- CPU side.

The reference holds no golden proof, so proof bytes cannot be compared with gnark's.  What CAN be pinned
without gnark, and is pinned here:

* the BSB22 challenge hash: RFC 9380 expand_message_xmd(SHA-256) against the RFC's published vectors (appendix K.1,
  tests/golden/rfc9380_k1_xmd_sha256.json) -- for the oracle's C and Python expanders and for the product's host expander (the
  verifier's path); the product's DEVICE expander (the prover's k_bsb22_challenge) is in tests/test_gpu_pins.py;
* pairing, G1 / G2 generators, Fp2 coordinate order and the verifying-key field order: the reference's own
  verifier/impl/generated/vk.* (made by gnark's setup) satisfy e(beta1, G2) = e(G1, beta2) and e(delta1, G2) = e(G1, delta2) -- under
  the oracle's pairing and under the product's host pairing, each reading the key with its own decoder.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import gnark_format as gf
from oracle import pairing

import helpers as H

# BN254 G2 generator (gnark-crypto ecc/bn254 g2Gen; EIP-197's P2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))
G1_GEN = (1, 2)


def rfc_vectors():
    d = json.load(open(os.path.join(H.ROOT, "tests", "golden", "rfc9380_k1_xmd_sha256.json")))
    out = []
    for v in d["vectors"]:
        msg = v["msg"].encode() + (v["repeat"][0] * v["repeat"][1]).encode() if "repeat" in v else v["msg"].encode()
        out.append((msg, d["DST"].encode(), v["len_in_bytes"], bytes.fromhex(v["uniform_bytes"])))
    return out


def _xmd_by_the_rfc(msg, dst, n):
    """RFC 9380 section 5.3.1, steps 1-12, written out with hashlib (the yardstick that checks the vector file itself)"""
    ell = -(-n // 32)
    dst_prime = dst + bytes([len(dst)])
    msg_prime = bytes(64) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime
    b = [hashlib.sha256(msg_prime).digest()]
    b.append(hashlib.sha256(b[0] + b"\x01" + dst_prime).digest())
    for i in range(2, ell + 1):
        b.append(hashlib.sha256(bytes(x ^ y for x, y in zip(b[0], b[i - 1])) + bytes([i]) + dst_prime).digest())
    return b"".join(b[1:])[:n]


def xmd_through(lib_fn, msg, dst, n, *extra):
    out = np.zeros(n, np.uint8)
    m = np.frombuffer(msg, np.uint8).copy() if msg else np.zeros(1, np.uint8)
    d = np.frombuffer(dst, np.uint8).copy()
    rc = lib_fn(m.ctypes.data, len(msg), d.ctypes.data, len(dst), out.ctypes.data, n, *extra)
    assert rc == 0
    return out.tobytes()


def test_expand_message_xmd_rfc9380_vectors(zk, oracle):
    olib = oracle.lib()
    olib.oracle_expand_message_xmd.restype = C.c_int
    olib.oracle_expand_message_xmd.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    vecs = rfc_vectors()
    assert len(vecs) == 6
    for msg, dst, n, want in vecs:
        assert _xmd_by_the_rfc(msg, dst, n) == want
        assert pairing.expand_message_xmd_py(msg, dst, n) == want                                   # oracle, Python (verification side)
        assert xmd_through(olib.oracle_expand_message_xmd, msg, dst, n) == want                      # oracle, C (its prover's BSB22 challenge)
        assert xmd_through(zk.testlib.zkp_expand_message_xmd, msg, dst, n, 0) == want                # product, host (hostutil.cpp: verifier's challenge)
    # the shape the prover uses: 64-byte message, 16-byte tag, 48 bytes out -- all implementations agree there too
    rng = np.random.default_rng(9380)
    for _ in range(8):
        msg = rng.integers(0, 256, 64, dtype=np.uint8).tobytes()
        want = _xmd_by_the_rfc(msg, b"bsb22-commitment", 48)
        assert pairing.expand_message_xmd_py(msg, b"bsb22-commitment", 48) == want
        assert xmd_through(olib.oracle_expand_message_xmd, msg, b"bsb22-commitment", 48) == want
        assert xmd_through(zk.testlib.zkp_expand_message_xmd, msg, b"bsb22-commitment", 48, 0) == want
        assert pairing.hash_to_fr_py(msg, b"bsb22-commitment") == int.from_bytes(want, "big") % H.R
        got = np.zeros(32, np.uint8)
        m = np.frombuffer(msg, np.uint8).copy()
        assert zk.testlib.zkp_bsb22_challenge_host(m.ctypes.data, 1, got.ctypes.data) == 0
        assert int.from_bytes(got.tobytes(), "little") == int.from_bytes(want, "big") % H.R
        o32 = np.zeros(32, np.uint8)
        olib.oracle_hash_to_fr(m.ctypes.data, C.c_long(64), b"bsb22-commitment", o32.ctypes.data)
        assert int.from_bytes(o32.tobytes(), "little") == int.from_bytes(want, "big") % H.R


def vk_identity_pairs(vk):
    return ([(vk["beta1"], G2_GEN), (pairing.g1_neg(G1_GEN), vk["beta2"])],
            [(vk["delta1"], G2_GEN), (pairing.g1_neg(G1_GEN), vk["delta2"])])


@pytest.mark.parametrize("name", ["chacha20", "aes128", "aes256"])
def test_reference_vk_satisfies_the_setup_identities(zk, name):
    """gnark's setup made these files: beta1 = beta G1, beta2 = beta G2, delta likewise, so e(beta1, G2) = e(G1, beta2) and
    e(delta1, G2) = e(G1, delta2).  Holds only if the pairing, both generators, the Fp2 coordinate order (A1 | A0 on the wire) and the
    field order of VerifyingKey.WriteTo are gnark's -- for the oracle (gnark_format.vk_parse + pairing.py) and for the product
    (verifier.cpp: vk_parse + host ate pairing), independently."""
    buf = open(os.path.join(H.ROOT, "tests", "golden", "vk." + name), "rb").read()
    vk = gf.vk_parse(buf)
    for pairs in vk_identity_pairs(vk):
        assert pairing.pairing_product_is_one(pairs)
    # a swapped pair of fields must NOT pass (the identities are not vacuous)
    assert not pairing.pairing_product_is_one([(vk["beta1"], G2_GEN), (pairing.g1_neg(G1_GEN), vk["delta2"])])
    b = np.frombuffer(buf, np.uint8).copy()
    assert zk.testlib.zkp_vk_pairing_identities(b.ctypes.data, len(buf), 0) == 3
    # the same points through the generic product check, with the generators supplied from here (raw gnark-crypto encoding)
    for pairs, expect in ((vk_identity_pairs(vk)[0], 1), ([(vk["beta1"], G2_GEN), (pairing.g1_neg(G1_GEN), vk["delta2"])], 0)):
        g1 = np.frombuffer(b"".join(gf.g1_raw(p) for p, _ in pairs), np.uint8).copy()
        g2 = np.frombuffer(b"".join(gf.g2_raw(q) for _, q in pairs), np.uint8).copy()
        assert zk.testlib.zkp_pairing_product_check(g1.ctypes.data, g2.ctypes.data, len(pairs), 0) == expect


def test_witness_binary_layout(zk):
    """the layout ZkpProveWitness takes (gnark witness.MarshalBinary): header and element encoding"""
    w = zk.witness_binary(2, [1, 2, 3])
    assert w[:12] == bytes([0, 0, 0, 2, 0, 0, 0, 1, 0, 0, 0, 3]) and len(w) == 12 + 96
    assert w[12:44] == (1).to_bytes(32, "big") and w[-32:] == (3).to_bytes(32, "big")
    # not initialised -> the error groth16.Prove's caller would see, not a crash
    with pytest.raises(RuntimeError, match="not initialized"):
        zk.prove_witness(0, w)
