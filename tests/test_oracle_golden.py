"""CPU tests: pin the oracle against every golden vector the reference holds for this path
 and against its own independent second implementations."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import gnark_format as gf
from oracle import pairing
from oracle import r1cs_format as rf
from oracle import witness as W

import helpers as H

R = rf.R_MOD


# ---------------------------------------------------------------- cipher known answers (reference tests)
def test_chacha_quarter_round_rfc8439_2_1_1():
    # gnark/circuits/chachaV3/chacha_test.go:21-29
    s = [0x11111111, 0x01020304, 0x9B8D6F43, 0x01234567]
    W._qr(s, 0, 1, 2, 3)
    assert s == [0xEA2A92F4, 0xCB1CF8CE, 0x4581472E, 0x5881C4BB]


def test_chacha_block_rfc8439_2_3_2():
    # gnark/circuits/chachaV3/chacha_test.go:94-104
    key = bytes(range(32))
    nonce = bytes.fromhex("000000090000004a00000000")
    out = W.chacha20_block(key, 1, nonce)
    assert out.hex() == ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                         "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_chacha_cipher_vector_chacha_test_go_123_152():
    key = bytes(range(32))
    nonce = bytes.fromhex("000000000000004a00000000")
    pt = bytes.fromhex(
        "4c616469657320616e642047656e746c656d656e206f662074686520636c617373206f66202739393a204966204920636f"
        "756c64206f6666657220796f75206f6e6c79206f6e652074697020666f7220746865206675747572652c2073756e736372"
        "65656e20776f756c642062652069" "4c616469657320616e642047656e746c")
    ct = bytes.fromhex(
        "6e2e359a2568f98041ba0728dd0d6981e97e7aec1d4360c20a27afccfd9fae0bf91b65c5524733ab8f593dabcd62b357"
        "1639d624e65152ab8f530c359f0861d807ca0dbf500d6a6156a38e088a22b65e52bc514d16ccf806818ce91ab7793736"
        "5af90bbf74a35be6b40b8eedf2785e42" "bf02106a16533ac0e69f9caf5cffb081")
    wit, out = W.chacha_witness(key, [nonce, nonce], [1, 2], pt)
    assert out == ct
    assert len(wit) == 2304 + 256


def test_aes_rfc3686_and_fips197():
    # gnark/circuits/aesV2/aes128_test.go:20-24 (RFC 3686 #2), aes256_test.go:20-24 key shape, FIPS-197 C.3
    k = bytes.fromhex("7E24067817FAE0D743D6CE1F32539163")
    n = bytes.fromhex("006CB6DBC0543B59DA48D90B")
    pt = bytes(range(32))
    ct = b"".join(bytes(a ^ b for a, b in zip(pt[16 * i:16 * i + 16], W.aes_encrypt_block(k, n + (1 + i).to_bytes(4, "big")))) for i in range(2))
    assert ct.hex().upper() == "5104A106168A72D9790D41EE8EDAD388EB2E1EFC46DA57C8FCE630DF9141BE28"
    assert W.aes_encrypt_block(bytes(range(32)), bytes.fromhex("00112233445566778899aabbccddeeff")).hex() == "8ea2b7ca516745bfeafc49904b496089"
    assert W.aes_encrypt_block(bytes(range(16)), bytes.fromhex("00112233445566778899aabbccddeeff")).hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"


def test_aux_test_go_request_ciphertext_pin():
    # gnark/libraries/aux_test.go:150-185
    j = json.load(open(os.path.join(H.ROOT, "tests", "golden", "aux_test_request.json")))
    key = base64.b64decode(j["key"])
    wit, out = W.aes_witness(key, [base64.b64decode(b["nonce"]) for b in j["blocks"]], [b["counter"] for b in j["blocks"]], base64.b64decode(j["input"]))
    assert out == base64.b64decode(j["ciphertext"])
    assert len(wit) == 225 + 16


# ---------------------------------------------------------------- R1CS artefacts
@pytest.mark.parametrize("name,stats", [
    ("chacha20", (46912, 2305, 256, 43680, 47584, 900224, 163, 40)),
    ("aes128", (85955, 226, 16, 84576, 90312, 1480956, 432, 4981)),
    ("aes256", (116501, 226, 32, 114397, 122298, 1994931, 457, 4981)),
])
def test_r1cs_hash_pin_and_decoded_shape(name, stats):
    data = rf.load_r1cs_bytes(name)
    assert hashlib.sha256(data).hexdigest() == rf.CIRCUIT_SHA256[name]      # impl/library.go:38,43,48
    r = rf.parse_r1cs(data, name)
    got = (r.nb_constraints, r.nb_public, r.nb_secret, r.nb_internal, r.nb_instructions, len(r.calldata), len(r.levels), len(r.coeffs))
    assert got == stats
    assert sorted(i for l in r.levels for i in l) == list(range(r.nb_instructions))
    assert r.coeffs[:5] == [0, 1, 2, R - 1, R - 2]
    assert r.wire_offset[-1] <= r.nb_wires


def _rfc_instance(alg):
    if alg == 0:
        return dict(key=bytes(range(32)), nonces=[bytes.fromhex("000000000000004a00000000")] * 2, counters=[1, 2], data=bytes(range(128)))
    if alg == 1:
        return dict(key=bytes.fromhex("7E24067817FAE0D743D6CE1F32539163"), nonces=[bytes.fromhex("006CB6DBC0543B59DA48D90B")] * 5, counters=[1, 2, 3, 4, 5], data=bytes(range(80)))
    return dict(key=bytes.fromhex("F6D66D6BD52D59BB0796365879EFF886C66DD51A5B6A99744B50590C87A23884"), nonces=[bytes.fromhex("00FAAC24C1585EF15A43D875")] * 5,
                counters=[1, 2, 3, 4, 5], data=bytes(range(80)))


@pytest.mark.parametrize("alg", [0, 1, 2])
def test_solver_satisfies_every_constraint_and_detects_tampering(oracle, alg):
    inst = _rfc_instance(alg)
    wit, _ = H.oracle_witness(alg, inst)
    c = H.circuit(alg)
    wires, a, b, cc = c.solve(wit, randomize=12345, challenge_override=6789)
    A, B, C = oracle.le32_to_ints(a), oracle.le32_to_ints(b), oracle.le32_to_ints(cc)
    assert all(x * y % R == z for x, y, z in zip(A, B, C))
    # independent re-evaluation of <L,w>, <R,w>, <O,w> from the decoded calldata in pure python
    r, cd, wv = c.r1cs, c.r1cs.calldata, oracle.le32_to_ints(wires)
    step = 37 if alg else 11
    for i in range(0, r.nb_instructions, step):
        if r.blueprints[r.blueprint_id[i]][0] != "r1c":
            continue
        s = r.start_calldata[i]
        p = s + 4
        vals = []
        for n in (cd[s + 1], cd[s + 2], cd[s + 3]):
            vals.append(sum(r.coeffs[cd[p + 2 * k]] * wv[cd[p + 2 * k + 1]] for k in range(n)) % R)
            p += 2 * n
        j = r.constraint_offset[i]
        assert (vals[0], vals[1], vals[2]) == (A[j], B[j], C[j])
    bad = list(wit)
    bad[len(bad) // 2] ^= 1
    with pytest.raises(ValueError, match="not satisfied"):
        c.solve(bad, randomize=1, challenge_override=2)


def test_witness_shape_statistics_match_survey_app_d(oracle):
    inst = _rfc_instance(0)
    wit, _ = H.oracle_witness(0, inst)
    wires, *_ = H.circuit(0).solve(wit)
    v = oracle.le32_to_ints(wires)
    assert (v.count(0), v.count(1), sum(1 for x in v if x >= 2 ** 64)) == (23561, 17331, 5349)


# ---------------------------------------------------------------- arithmetic cross-checks
def test_field_and_hash_against_python(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        a = int.from_bytes(rng.bytes(32), "little") % R
        b = int.from_bytes(rng.bytes(32), "little") % R
        out = np.zeros(32, np.uint8)
        xa, xb = oracle.ints_to_le32([a]), oracle.ints_to_le32([b])
        oracle.lib().oracle_fr_mul(xa.ctypes.data, xb.ctypes.data, out.ctypes.data)
        assert int.from_bytes(out.tobytes(), "little") == a * b % R
        oracle.lib().oracle_fp_mul(xa.ctypes.data, xb.ctypes.data, out.ctypes.data)
        assert int.from_bytes(out.tobytes(), "little") == a * b % gf.P
    for msg in (b"", b"abc", bytes(range(64)), bytes(200)):
        assert oracle.hash_to_fr(msg, b"bsb22-commitment") == pairing.hash_to_fr_py(msg, b"bsb22-commitment")
    out = np.zeros(32, np.uint8)
    m = np.frombuffer(b"abc", np.uint8).copy()
    oracle.lib().oracle_sha256(m.ctypes.data, 3, out.ctypes.data)
    assert out.tobytes() == hashlib.sha256(b"abc").digest()


def test_ntt_against_naive_dft(oracle):
    n = 16
    w = pow(5, (R - 1) // n, R)
    rng = np.random.default_rng(2)
    x = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    fwd = oracle.le32_to_ints(oracle.ntt(oracle.ints_to_le32(x)))
    assert fwd == [sum(x[j] * pow(w, j * k, R) for j in range(n)) % R for k in range(n)]
    cos = oracle.le32_to_ints(oracle.ntt(oracle.ints_to_le32(x), coset=True))
    assert cos == [sum(x[j] * pow(5 * pow(w, k, R), j, R) for j in range(n)) % R for k in range(n)]
    assert oracle.le32_to_ints(oracle.ntt(oracle.ntt(oracle.ints_to_le32(x)), inverse=True)) == x
    assert oracle.le32_to_ints(oracle.ntt(oracle.ntt(oracle.ints_to_le32(x), coset=True), inverse=True, coset=True)) == x


def test_compute_h_is_the_quotient(oracle):
    n, ncons = 64, 50
    rng = np.random.default_rng(3)
    a = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(ncons)]
    b = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(ncons)]
    c = [x * y % R for x, y in zip(a, b)]
    h = oracle.le32_to_ints(oracle.compute_h(oracle.ints_to_le32(a), oracle.ints_to_le32(b), oracle.ints_to_le32(c), n))
    assert h[n - 1] == 0
    # check A(z)B(z) - C(z) = H(z) (z^n - 1) at a random point through Lagrange interpolation
    z = 123456789
    w = pow(5, (R - 1) // n, R)
    def interp(vals):
        vals = vals + [0] * (n - len(vals))
        zn = (pow(z, n, R) - 1) * pow(n, -1, R) % R
        return sum(v * zn % R * pow(w, i, R) % R * pow(z - pow(w, i, R), -1, R) for i, v in enumerate(vals)) % R
    lhs = (interp(a) * interp(b) - interp(c)) % R
    rhs = sum(hk * pow(z, k, R) for k, hk in enumerate(h)) % R * (pow(z, n, R) - 1) % R
    assert lhs == rhs


def test_msm_bucket_method_against_naive_and_python(oracle):
    rng = np.random.default_rng(4)
    n = 200
    ks = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    pts = oracle.g1_fixed_base(oracle.ints_to_le32(ks))
    sc = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    sc[:6] = [0, 1, R - 1, 2, 1 << 200, R - 2]
    s8 = oracle.ints_to_le32(sc)
    got = oracle.msm_g1(pts, s8)
    assert np.array_equal(got, oracle.msm_g1(pts, s8, naive=True))
    tot = sum(s * k for s, k in zip(sc, ks)) % R
    assert tuple(oracle.le32_to_ints(got)) == pairing.g1_mul((1, 2), tot)
    pts2 = oracle.g2_fixed_base(oracle.ints_to_le32(ks[:40]))
    assert np.array_equal(oracle.msm_g2(pts2, s8[:40]), oracle.msm_g2(pts2, s8[:40], naive=True))


def test_pairing_bilinearity(oracle):
    g2 = np.zeros(128, np.uint8)
    oracle.lib().oracle_g2_gen(g2.ctypes.data)
    v = oracle.le32_to_ints(g2)
    G2 = ((v[0], v[1]), (v[2], v[3]))
    assert gf.g2_on_curve(G2)
    a, b = 0x1234567, 0x89ABCDEF0123
    v = oracle.le32_to_ints(oracle.g2_fixed_base(oracle.ints_to_le32([b]))[0])
    bG2 = ((v[0], v[1]), (v[2], v[3]))
    aG = pairing.g1_mul((1, 2), a)
    assert pairing.pairing_product_is_one([(aG, bG2), (pairing.g1_neg(pairing.g1_mul((1, 2), a * b)), G2)])
    assert not pairing.pairing_product_is_one([(aG, bG2), (pairing.g1_neg(pairing.g1_mul((1, 2), a * b + 1)), G2)])


# ---------------------------------------------------------------- formats decoded from the reference's artefacts
@pytest.mark.parametrize("name,nk,nkeys,size", [("chacha20", 2305, 0, 74060), ("aes128", 227, 1, 7696), ("aes256", 227, 1, 7696)])
def test_reference_verifying_keys_decode(name, nk, nkeys, size):
    # gnark/libraries/verifier/impl/generated/vk.* (committed as data fixtures)
    buf = open(os.path.join(H.ROOT, "tests", "golden", "vk." + name), "rb").read()
    assert len(buf) == size
    vk = gf.vk_parse(buf)
    assert len(vk["K"]) == nk and len(vk["commitment_keys"]) == nkeys
    assert all(gf.g1_on_curve(p) for p in [vk["alpha1"], vk["beta1"], vk["delta1"]] + vk["K"])
    assert all(gf.g2_on_curve(p) for p in [vk["beta2"], vk["gamma2"], vk["delta2"]] + [q for k in vk["commitment_keys"] for q in k])
    assert gf.vk_bytes(vk) == buf          # our encoder reproduces the reference file byte for byte


# ---------------------------------------------------------------- end to end: the reference's own oracle (prove -> verify)
@pytest.mark.parametrize("alg", [0, 1])
def test_oracle_prove_then_verify_and_tamper(oracle, alg):
    inst = H.make_instance(alg, H.SplitMix64(2024 + alg))
    pb, out, prf, wit = H.oracle_proof_bytes(alg, inst)
    assert len(pb) == (164 if alg == 0 else 196)
    pk = H.oracle_pk(alg)
    assert [pk.count(i) for i in (0, 1, 2, 3)] == ([43681, 24737, 65535, 43936] if alg == 0 else [68212, 57310, 131071, 66626])
    vk = gf.vk_from_pk(pk)
    assert len(gf.vk_bytes(vk)) == (74060 if alg == 0 else 7696)      # same size as the reference's vk files
    npub = H.circuit(alg).r1cs.nb_public - 1
    pp = gf.proof_parse(pb)
    assert pairing.groth16_verify(vk, pp, wit[:npub])
    bad = list(wit[:npub]); bad[0] ^= 1
    assert not pairing.groth16_verify(vk, pp, bad)
    pp2 = dict(pp); pp2["Krs"] = pairing.g1_add(pp["Krs"], (1, 2))
    assert not pairing.groth16_verify(vk, pp2, wit[:npub])
    # blinding changes the proof, not its validity
    inst2 = dict(inst, r=inst["r"] + 1)
    pb2, *_ = H.oracle_proof_bytes(alg, inst2)
    assert pb2 != pb and pairing.groth16_verify(vk, gf.proof_parse(pb2), wit[:npub])


def test_proving_key_serialisation_roundtrip_of_points(oracle):
    pk = H.oracle_pk(0)
    raw = gf.pk_bytes(pk, raw=True)
    assert int.from_bytes(raw[:8], "big") == 65536
    pos = 8 + 5 * 32 + 1
    a, pos = gf.g1_decode(raw, pos)
    assert gf.g1_on_curve(a) and a == gf._le_pts_to_int(pk.g1(7))[0]
    for pt in gf._le_pts_to_int(pk.g1(0)[:50]):
        assert gf.g1_decode(gf.g1_compress(pt))[0] == pt
    for t in gf._le_pts_to_int(pk.g2_B()[:20]):
        pt = ((t[0], t[1]), (t[2], t[3]))
        assert gf.g2_decode(gf.g2_compress(pt))[0] == pt and gf.g2_on_curve(pt)
