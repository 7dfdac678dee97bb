"""CPU tests of the product's verifier (libverify ABI; host-only code, no GPU needed): it must agree
with the oracle's independent pairing on oracle-made proofs -- the reference's own check `verifier.Verify(prover.Prove(x))`
(gnark/libraries/test_helpers.go:332-339, 403-410) and its tamper tests (js/src/tests/lib.test.ts:124-158)."""
import base64
import json
import os

import pytest

from oracle import gnark_format as gf
from oracle import pairing

import helpers as H


def _req(alg, inst, proof, out):
    b = lambda x: base64.b64encode(x).decode()
    return {"cipher": H.CIPHER[alg], "proof": b(proof),
            "publicSignals": {"ciphertext": b(out), "blocks": [{"nonce": b(n), "counter": c, "boundary": None} for n, c in zip(inst["nonces"], inst["counters"])],
                              "input": b(inst["data"])}}


@pytest.mark.parametrize("alg", [0, 1])
def test_verify_accepts_oracle_proofs_and_rejects_tampering(zk, oracle, alg):
    inst = H.make_instance(alg, H.SplitMix64(777 + alg))
    proof, out, prf, wit = H.oracle_proof_bytes(alg, inst)
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    assert zk.init_verifier(alg, gf.vk_bytes(vk))
    good = _req(alg, inst, proof, out)
    assert zk.verify_raw(json.dumps(good).encode()) is True
    npub = H.circuit(alg).r1cs.nb_public - 1
    assert pairing.groth16_verify(vk, gf.proof_parse(proof), wit[:npub])        # same verdict as the oracle
    # tampered ciphertext / input / counter / proof -> false (lib.test.ts:124-158 "invalid proof")
    for field in ("ciphertext", "input"):
        bad = json.loads(json.dumps(good))
        raw = bytearray(base64.b64decode(bad["publicSignals"][field])); raw[5] ^= 1
        bad["publicSignals"][field] = base64.b64encode(bytes(raw)).decode()
        assert zk.verify_raw(json.dumps(bad).encode()) is False
    bad = json.loads(json.dumps(good)); bad["publicSignals"]["blocks"][0]["counter"] += 1
    assert zk.verify_raw(json.dumps(bad).encode()) is False
    other, *_ = H.oracle_proof_bytes(alg, H.make_instance(alg, H.SplitMix64(1)))
    bad = json.loads(json.dumps(good)); bad["proof"] = base64.b64encode(other).decode()
    assert zk.verify_raw(json.dumps(bad).encode()) is False
    raw = bytearray(proof); raw[40] ^= 0x10
    bad = json.loads(json.dumps(good)); bad["proof"] = base64.b64encode(bytes(raw)).decode()
    assert zk.verify_raw(json.dumps(bad).encode()) is False


def test_verifier_rejects_g2_points_outside_the_subgroup(zk, oracle):
    """gnark-crypto's G2Affine.SetBytes (proof.ReadFrom, the embedded vk) refuses points of the twist that are not in the
    order-r subgroup (cofactor 2p - r); so must Verify (Bs of the proof) and ZkpInitVerifier (beta2 / gamma2 / delta2)."""
    alg = 0
    inst = H.make_instance(alg, H.SplitMix64(4242))
    proof, out, *_ = H.oracle_proof_bytes(alg, inst)
    vkb = gf.vk_bytes(gf.vk_from_pk(H.oracle_pk(alg)))
    assert zk.init_verifier(alg, vkb)
    assert zk.verify_raw(json.dumps(_req(alg, inst, proof, out)).encode()) is True
    xx = 1
    while True:                                    # a random twist point is (almost surely) outside G2
        comp = bytearray((0).to_bytes(32, "big") + xx.to_bytes(32, "big")); comp[0] |= 0x80
        try:
            pt, _ = gf.g2_decode(bytes(comp))
            break
        except ValueError:
            xx += 1
    assert gf.g2_on_curve(pt)
    bad = bytearray(proof); bad[32:96] = gf.g2_compress(pt)                    # Proof.WriteTo: Ar(32) Bs(64) Krs(32) ...
    assert zk.verify_raw(json.dumps(_req(alg, inst, bytes(bad), out)).encode()) is False
    for off in (64, 128, 224):                                                 # vk: alpha1(32) beta1(32) beta2(64) gamma2(64) delta1(32) delta2(64)
        badvk = bytearray(vkb); badvk[off:off + 64] = gf.g2_compress(pt)
        assert not zk.init_verifier(alg, bytes(badvk))
    assert zk.init_verifier(alg, vkb)


def test_verify_malformed_requests_return_false(zk):
    for bad in (b"", b"{", b"[]", b'{"cipher":"rot13"}', b'{"cipher":"chacha20","proof":"AAAA","publicSignals":{}}',
                b'{"cipher":"chacha20","proof":"!!","publicSignals":{"ciphertext":"","blocks":[],"input":""}}'):
        assert zk.verify_raw(bad) is False


def test_reference_verifying_keys_load(zk):
    """the reference's own vk files (gnark/libraries/verifier/impl/generated/vk.*) decode: every point decompresses onto its curve"""
    for alg, name in ((0, "chacha20"), (1, "aes128")):          # (aes256: tests/test_oracle_golden.py; its slot stays free for the test below)
        buf = open(os.path.join(H.ROOT, "tests", "golden", "vk." + name), "rb").read()
        assert zk.init_verifier(alg, buf)
        assert not zk.init_verifier(alg, buf[:-3])


def test_vk_directory_fallback(zk, oracle, tmp_path, monkeypatch):
    """drop-in behaviour: with no ZkpInitVerifier call the key is read from $ZKPROVE_VK_DIR/vk.<ext> on first use"""
    alg = 2            # no other test initialises the aes-256 verifier in this process
    inst = H.make_instance(alg, H.SplitMix64(4))
    proof, out, *_ = H.oracle_proof_bytes(alg, inst)
    (tmp_path / "vk.aes256").write_bytes(gf.vk_bytes(gf.vk_from_pk(H.oracle_pk(alg))))
    req = json.dumps(_req(alg, inst, proof, out)).encode()
    monkeypatch.delenv("ZKPROVE_VK_DIR", raising=False)
    assert zk.verify_raw(req) is False
    monkeypatch.setenv("ZKPROVE_VK_DIR", str(tmp_path))
    assert zk.verify_raw(req) is True


def test_vk_files_beside_the_shared_object(zk, oracle, tmp_path):
    """The reference embeds generated/vk.* in libverify at build time (gnark/libraries/verifier/impl/library.go:92-108), so the unchanged
    JS `lib.verify` works with no set-up call.  Drop-in equivalent: with ZKPROVE_VK_DIR unset the library reads vk.<ext> from the
    directory IT was loaded from (`make dropin` copies them beside bin/gnark/linux-x86_64-libverify.so).  Own process: a copy of the
    library in a directory that holds the key, nothing initialised."""
    import shutil
    import subprocess
    import sys
    alg = 0
    inst = H.make_instance(alg, H.SplitMix64(6))
    proof, out, *_ = H.oracle_proof_bytes(alg, inst)
    shutil.copy(zk.LIB_PATH, tmp_path / "linux-x86_64-libverify.so")
    (tmp_path / "vk.chacha20").write_bytes(gf.vk_bytes(gf.vk_from_pk(H.oracle_pk(alg))))
    (tmp_path / "req.json").write_bytes(json.dumps(_req(alg, inst, proof, out)).encode())
    script = (
        "import ctypes as C, sys\n"
        "class GoSlice(C.Structure):\n"
        "    _fields_ = [('data', C.c_void_p), ('len', C.c_longlong), ('cap', C.c_longlong)]\n"
        "lib = C.CDLL(sys.argv[1] + '/linux-x86_64-libverify.so')\n"
        "lib.Verify.restype = C.c_ubyte; lib.Verify.argtypes = [GoSlice]\n"
        "b = open(sys.argv[1] + '/req.json', 'rb').read()\n"
        "buf = C.create_string_buffer(b, len(b))\n"
        "print('VERIFY', int(lib.Verify(GoSlice(C.cast(buf, C.c_void_p), len(b), len(b)))))\n")
    env = {k: v for k, v in os.environ.items() if k != "ZKPROVE_VK_DIR"}
    res = subprocess.run([sys.executable, "-c", script, str(tmp_path)], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "VERIFY 1" in res.stdout, res.stdout[-500:]
    os.remove(tmp_path / "vk.chacha20")                     # without the file: false, as before
    res = subprocess.run([sys.executable, "-c", script, str(tmp_path)], env=env, capture_output=True, text=True, timeout=300)
    assert "VERIFY 0" in res.stdout
