"""GPU side of tests/test_pins.py: the DEVICE code of the prover / batch verifier against published or gnark-made data, and the
witness-level inner boundary ZkpProveWitness (groth16.Prove(r1cs, pk, wtns), gnark/libraries/prover/impl/provers.go:175-188, 265-279)."""
import base64
import json
import os

import numpy as np
import pytest

from oracle import gnark_format as gf
from oracle import pairing
from oracle import r1cs_format as rf

import helpers as H
from test_pins import G1_GEN, G2_GEN, rfc_vectors, vk_identity_pairs, xmd_through, _xmd_by_the_rfc

pytestmark = pytest.mark.gpu


def test_device_expander_against_rfc9380_vectors(zk):
    """expand_message_xmd_dev (csrc/xmd.cuh) is the function k_bsb22_challenge calls: RFC 9380 K.1 vectors, then the prover's shape"""
    for msg, dst, n, want in rfc_vectors():
        assert xmd_through(zk.testlib.zkp_expand_message_xmd, msg, dst, n, 1) == want
    for msg, dst, n in ((b"", b"x", 1), (b"a" * 55, b"d" * 255, 33), (b"b" * 119, b"bsb22-commitment", 255 * 32)):      # padding edges, longest tag / output
        assert xmd_through(zk.testlib.zkp_expand_message_xmd, msg, dst, n, 1) == _xmd_by_the_rfc(msg, dst, n)


def test_bsb22_challenge_kernel_matches_hash_to_field(zk, oracle):
    """k_bsb22_challenge on commitments: = fr.Hash(X || Y, "bsb22-commitment", 1)[0] as the oracle's Python hash computes it and as
    the product's verifier (host) computes it; includes the all-zero (infinity) commitment"""
    rng = np.random.default_rng(22)
    n = 130
    xs = [int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "big") % gf.P for _ in range(n)]
    ys = [int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "big") % gf.P for _ in range(n)]
    xs[0] = ys[0] = 0
    le = np.frombuffer(b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") for x, y in zip(xs, ys)), np.uint8).copy()
    be = np.frombuffer(b"".join(x.to_bytes(32, "big") + y.to_bytes(32, "big") for x, y in zip(xs, ys)), np.uint8).copy()
    got = np.zeros((n, 32), np.uint8)
    host = np.zeros((n, 32), np.uint8)
    assert zk.testlib.zkp_bsb22_challenge(le.ctypes.data, n, got.ctypes.data) == 0, zk.last_error()
    assert zk.testlib.zkp_bsb22_challenge_host(be.ctypes.data, n, host.ctypes.data) == 0
    want = [pairing.hash_to_fr_py(x.to_bytes(32, "big") + y.to_bytes(32, "big"), b"bsb22-commitment") for x, y in zip(xs, ys)]
    assert [int.from_bytes(g.tobytes(), "little") for g in got] == want
    assert np.array_equal(got, host)


@pytest.mark.parametrize("name", ["chacha20", "aes128", "aes256"])
def test_reference_vk_identities_through_the_device_miller_kernel(zk, name):
    """e(beta1, G2) = e(G1, beta2), e(delta1, G2) = e(G1, delta2) on the reference's own vk.* with the Miller loops in k_vb_miller
    (the batch verifier's kernel); a mismatched pair is rejected"""
    buf = open(os.path.join(H.ROOT, "tests", "golden", "vk." + name), "rb").read()
    b = np.frombuffer(buf, np.uint8).copy()
    assert zk.testlib.zkp_vk_pairing_identities(b.ctypes.data, len(buf), 1) == 3, zk.last_error()
    vk = gf.vk_parse(buf)
    for pairs, expect in ((vk_identity_pairs(vk)[1], 1), ([(vk["delta1"], G2_GEN), (pairing.g1_neg(G1_GEN), vk["beta2"])], 0)):
        g1 = np.frombuffer(b"".join(gf.g1_raw(p) for p, _ in pairs), np.uint8).copy()
        g2 = np.frombuffer(b"".join(gf.g2_raw(q) for _, q in pairs), np.uint8).copy()
        assert zk.testlib.zkp_pairing_product_check(g1.ctypes.data, g2.ctypes.data, len(pairs), 1) == expect


def test_product_setup_keys_satisfy_the_same_identities(zk, oracle):
    """ZkpSetup's verifying key under the oracle's pairing with the generators of test_pins.py: the product's generators are gnark's"""
    _, vkb = zk.setup(zk.synth.r1cs_bytes(0), oracle.toxic_from_seed(5), raw=False)
    vk = gf.vk_parse(vkb)
    for pairs in vk_identity_pairs(vk):
        assert pairing.pairing_product_is_one(pairs)


# ------------------------------------------------------------------------------------------------ inner boundary
@pytest.fixture(scope="module")
def algs(zk):
    zk.lib.ZkpAllowUnpinnedKeys(1)

    def get(alg):
        assert zk.init_algorithm(alg, H.pk_file_bytes(alg), rf.load_r1cs_bytes(H.NAMES[alg])), "InitAlgorithm(%d) failed" % alg
        return alg
    return get


@pytest.mark.parametrize("alg", [0, 1, 2])
def test_prove_witness_equals_prove_and_the_oracle(zk, oracle, algs, alg):
    """ZkpProveWitness(id, wtns.MarshalBinary()) -> Proof.WriteTo bytes: byte-identical to what Prove(json) returns for the request that
    assignment came from (same r, s, randomize) and to the oracle's proof"""
    algs(alg)
    inst = H.make_instance(alg, H.SplitMix64(0xB0DA + alg))
    wit, out = H.oracle_witness(alg, inst)
    npub = H.circuit(alg).r1cs.nb_public - 1
    wb = zk.witness_binary(npub, wit)
    got = zk.prove_witness(alg, wb, inst["r"], inst["s"], inst["randomize"])
    res = json.loads(zk.prove_deterministic_raw(H.request_json(alg, inst), inst["r"], inst["s"], inst["randomize"]))
    assert got == base64.b64decode(res["proof"])
    exp, *_ = H.oracle_proof_bytes(alg, inst)
    assert got == exp and len(got) == (164 if alg == 0 else 196)
    # random blinding: differs, verifies
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    p2 = zk.prove_witness(alg, wb)
    assert p2 != got and pairing.groth16_verify(vk, gf.proof_parse(p2), wit[:npub])


def test_prove_witness_errors_are_returned_not_thrown(zk, algs):
    alg = algs(0)
    inst = H.make_instance(alg, H.SplitMix64(77))
    wit, _ = H.oracle_witness(alg, inst)
    npub = H.circuit(alg).r1cs.nb_public - 1
    good = zk.witness_binary(npub, wit)
    with pytest.raises(RuntimeError, match="invalid witness size"):
        zk.prove_witness(alg, zk.witness_binary(npub - 1, wit))
    with pytest.raises(RuntimeError, match="bytes for"):
        zk.prove_witness(alg, good[:-1])
    with pytest.raises(RuntimeError, match="short header"):
        zk.prove_witness(alg, b"\x00\x00")
    with pytest.raises(RuntimeError, match="canonical"):
        zk.prove_witness(alg, good[:12] + H.R.to_bytes(32, "big") + good[44:])
    bad = list(wit); bad[1500] ^= 1                       # one bit of Out flipped: the solver reports the violated constraint
    with pytest.raises(RuntimeError, match=r"constraint #\d+ is not satisfied"):
        zk.prove_witness(alg, zk.witness_binary(npub, bad))
    wide = list(wit); wide[3] = H.R - 5                    # an element wider than the request layer's u32 packing: a bit wire that is not boolean
    with pytest.raises(RuntimeError, match="not satisfied"):
        zk.prove_witness(alg, zk.witness_binary(npub, wide))
    with pytest.raises(RuntimeError, match="unknown algorithm|not initialized"):
        zk.prove_witness(5, good)
    assert len(zk.prove_witness(alg, good)) == 164         # and the library is fine afterwards


def test_prove_witness_takes_full_width_elements(zk, oracle, algs):
    """an AES counter is a field element as far as the witness format goes: a wide value reaches the solver exactly (here it violates
    the circuit's range check, as in gnark), and a legal value just below 2^32 proves"""
    alg = algs(1)
    inst = H.make_instance(alg, H.SplitMix64(0xC0))
    inst["counters"] = [0xFFFFFFFB + i for i in range(5)]
    wit, out = H.oracle_witness(alg, inst)
    npub = H.circuit(alg).r1cs.nb_public - 1
    exp, *_ = H.oracle_proof_bytes(alg, inst)
    assert zk.prove_witness(alg, zk.witness_binary(npub, wit), inst["r"], inst["s"], inst["randomize"]) == exp
    wide = list(wit); wide[60] += 1 << 32                  # Counter[0] + 2^32
    with pytest.raises(RuntimeError, match="not satisfied|does not solve"):
        zk.prove_witness(alg, zk.witness_binary(npub, wide), 1, 1, 0)
