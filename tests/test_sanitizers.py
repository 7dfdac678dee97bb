"""Malformed-input corpus against the host-side readers under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only).

The reference gets memory safety from Go (gnark/libraries/prover/libprove.go:34-44 recovers every panic; aux_test.go:85-93); here a
hand-written JSON reader, base64 decoder, request layer and r1cs / vk / proof readers sit on a C ABI whose Verify takes untrusted
bytes.  `make -C zk-symmetric-crypto_amd asan` compiles those units (cabi.cpp, hostutil.cpp, r1cs_reader.cpp, verifier.cpp) with
-fsanitize=address,undefined on the host side and links them with the ordinary objects into libzkprove_asan.so; tests/host/fuzz_host.c
feeds it records.  Every input below must be ANSWERED (an error value, false, 0) -- any sanitizer report, crash or leak fails the test.
"""
import base64
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import gnark_format as gf
from oracle import r1cs_format as rf

import helpers as H

PKG = os.path.join(H.ROOT, "zk-symmetric-crypto_amd")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")

REQUEST_WITNESS, PROVE, PROVE_BATCH, VERIFY, VERIFY_BATCH, INIT_VERIFIER, R1CS_STATS, PROVE_WITNESS, INIT_ALGORITHM, PROOF_SERIALIZE = range(1, 11)


class Driver:
    """one fuzz_host process; send() returns the result byte of a record, close() the exit status and stderr"""

    def __init__(self):
        r, w = os.pipe()
        self.r = r
        env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
                   ZKPROVE_VK_DIR="/nonexistent", FUZZ_ACK_FD=str(w))
        self.p = subprocess.Popen([os.path.join(PKG, "fuzz_host")], stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, pass_fds=(w,), env=env)
        os.close(w)

    def send(self, kind, payload):
        self.p.stdin.write(bytes([kind]) + struct.pack("<I", len(payload)) + payload)
        self.p.stdin.flush()
        b = os.read(self.r, 1)
        if len(b) != 1:
            rc, err = self.close()
            raise AssertionError("fuzz_host died (exit %s) on kind %d, %d bytes: %r ...\n%s" % (rc, kind, len(payload), payload[:80], err[-4000:]))
        return b[0]

    def close(self):
        try:
            self.p.stdin.close()
        except BrokenPipeError:
            pass
        err = self.p.stderr.read().decode(errors="replace")
        rc = self.p.wait(timeout=600)
        os.close(self.r)
        return rc, err


@pytest.fixture(scope="module")
def driver():
    subprocess.run(["make", "-s", "-j4", "-C", PKG, "asan"], check=True, capture_output=True, timeout=3000)
    d = Driver()
    yield d
    rc, err = d.close()
    bad = [l for l in err.splitlines() if "AddressSanitizer" in l or "runtime error" in l or "LeakSanitizer" in l]
    assert rc == 0 and not bad, "exit %s\n%s" % (rc, err[-6000:])


def mutations(data: bytes, rng, n_trunc=24, n_flip=24):
    """truncations (dense near both ends), single-byte and 32-bit-field corruptions"""
    out = []
    cuts = sorted(set(list(range(0, min(len(data), 40))) + [int(x) for x in np.linspace(0, len(data), n_trunc)] + list(range(max(0, len(data) - 8), len(data)))))
    out += [data[:c] for c in cuts]
    for _ in range(n_flip):
        i = int(rng.integers(0, len(data)))
        b = bytearray(data)
        b[i] ^= 1 << int(rng.integers(0, 8))
        out.append(bytes(b))
    for _ in range(n_flip):
        i = int(rng.integers(0, max(1, len(data) - 4)))
        b = bytearray(data)
        b[i:i + 4] = struct.pack(">I", int(rng.choice([0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, len(data), len(data) * 8, 1 << 24])))
        out.append(bytes(b))
    out.append(data + b"\x00")
    out.append(data * 2)
    return out


def valid_request(alg=0, seed=1):
    return H.request_json(alg, H.make_instance(alg, H.SplitMix64(seed)))


def test_json_request_layer_survives_malformed_requests(driver):
    rng = np.random.default_rng(1)
    good = valid_request(0)
    assert driver.send(REQUEST_WITNESS, good) == 0                        # algorithm id 0: decoded, witness built
    assert driver.send(REQUEST_WITNESS, valid_request(2, 3)) == 2
    bombs = [b"[" * 100000, b"{\"a\":" * 50000, b"[" * 199 + b"]" * 199, b"[" * 201 + b"]" * 201, b"\"" + b"\\u" * 30000, b"\"\\ud800\"", b"\"\\u00", b"-", b"1e", b"1e999999",
             b"9" * 100000, b"{\"cipher\":\"chacha20\",\"key\":\"" + b"A" * 1000001 + b"\"}", b"\xff\xfe\x00", b"", b" ", b"nul", b"{\"cipher\":\"chacha20\",\"blocks\":[" + b"null," * 70000 + b"null]}",
             b"{\"cipher\":\"chacha20\",\"key\":[" + b"1," * 100000 + b"1]}", b"{\"cipher\":\"aes-128-ctr\",\"key\":\"====\"}", b"{\"cipher\":\"aes-128-ctr\",\"key\":\"A===\"}",
             b"{\"cipher\":\"chacha20\",\"blocks\":[{\"nonce\":\"AAAAAAAAAAAAAAAA\",\"counter\":4294967296}]}", b"{\"cipher\":\"chacha20\",\"blocks\":[{\"counter\":-1}]}",
             b"{\"CIPHER\":\"chacha20\",\"cipher\":5}", b"{\"cipher\":\"chacha20-toprf\"}"]
    for kind in (REQUEST_WITNESS, PROVE, PROVE_BATCH):
        for m in mutations(good, rng) + bombs:
            driver.send(kind, m)
        driver.send(kind, b"[" + good + b"," + good[:-5] + b"]")
        driver.send(kind, b"[" + b",".join([good] * 40) + b"]")
    # every prefix of the JS-shaped literal of aux_test.go:150-185
    lit = open(os.path.join(H.ROOT, "tests", "golden", "aux_test_request.json"), "rb").read()
    for c in range(0, len(lit), 3):
        driver.send(REQUEST_WITNESS, lit[:c])


def test_r1cs_reader_survives_corrupted_constraint_systems(driver):
    rng = np.random.default_rng(2)
    data = rf.load_r1cs_bytes("chacha20")
    assert driver.send(R1CS_STATS, data) == 1
    head = data[:4096]
    for m in mutations(data, rng, n_trunc=12, n_flip=10):
        driver.send(R1CS_STATS, m)
    for m in mutations(head, rng, n_trunc=40, n_flip=60):               # corrupt the header / first sections, keep the body
        driver.send(R1CS_STATS, m + data[len(head):])
    # InitAlgorithm: the hash pins see every byte first (impl/library.go:99-101,115-117); id range
    for alg in (0, 1, 2, 3, 6, 255):
        driver.send(INIT_ALGORITHM, bytes([alg]) + struct.pack("<I", 64) + b"\x00" * 64 + data[:1000])


def test_vk_and_proof_readers_survive_garbage(driver, oracle):
    rng = np.random.default_rng(3)
    for name in ("chacha20", "aes128"):
        vk = open(os.path.join(H.ROOT, "tests", "golden", "vk." + name), "rb").read()
        alg = 0 if name == "chacha20" else 1
        for m in mutations(vk, rng, n_trunc=30, n_flip=30):
            driver.send(INIT_VERIFIER, bytes([alg]) + m)
        for bad_alg in (3, 200):
            assert driver.send(INIT_VERIFIER, bytes([bad_alg]) + vk) == 0
        assert driver.send(INIT_VERIFIER, bytes([alg]) + vk) == 1
    # a key and a proof that belong together (oracle): Verify is true, every corruption of proof / public signals is answered false
    alg = 0
    inst = H.make_instance(alg, H.SplitMix64(99))
    proof, out, *_ = H.oracle_proof_bytes(alg, inst)
    assert driver.send(INIT_VERIFIER, bytes([alg]) + gf.vk_bytes(gf.vk_from_pk(H.oracle_pk(alg)))) == 1
    b64 = lambda x: base64.b64encode(x).decode()

    def req(p, ct=out, data=inst["data"], nonces=inst["nonces"]):
        return json.dumps({"cipher": "chacha20", "proof": b64(p), "publicSignals": {"ciphertext": b64(ct), "input": b64(data),
                                                                                    "blocks": [{"nonce": b64(n), "counter": c} for n, c in zip(nonces, inst["counters"])]}}).encode()
    assert driver.send(VERIFY, req(proof)) == 1
    flags = []
    for off in (0, 32, 96, 132):                                          # the flag bits of every point of the proof
        for top in (0x00, 0x40, 0x80, 0xC0, 0xFF):
            b = bytearray(proof)
            b[off] = (b[off] & 0x3F) | (top & 0xC0) if top != 0xFF else 0xFF
            flags.append(bytes(b))
    for p in mutations(proof, rng, n_trunc=30, n_flip=16)[:60] + flags:
        if p != proof:
            assert driver.send(VERIFY, req(p)) == 0
    assert driver.send(VERIFY, req(proof, ct=out[:-1])) == 0
    assert driver.send(VERIFY, req(proof, nonces=[b"", b"x" * 13])) == 0
    batch = b"[" + b",".join([req(proof), req(proof[:-1]), b"{}", b"7", req(proof)]) + b"]"
    driver.send(VERIFY_BATCH, batch)
    for m in mutations(batch, rng, n_trunc=20, n_flip=10)[:30]:
        driver.send(VERIFY_BATCH, m)
    # gnark binary witnesses for an algorithm that is not initialised here: the header is read, nothing else
    for w in (b"", b"\x00" * 11, b"\x00" * 12, struct.pack(">III", 0xFFFFFFFF, 1, 0), struct.pack(">III", 2304, 256, 2560) + b"\x00" * 64):
        for a in (0, 2, 5, 9):
            assert driver.send(PROVE_WITNESS, bytes([a]) + w) == 0
    for _ in range(8):
        driver.send(PROOF_SERIALIZE, rng.integers(0, 256, 384, dtype=np.uint8).tobytes())


json_leaf = st.one_of(st.none(), st.booleans(), st.integers(min_value=-2**70, max_value=2**70), st.floats(allow_nan=False, allow_infinity=False), st.text(max_size=40),
                      st.binary(max_size=48).map(lambda b: base64.b64encode(b).decode()))
json_value = st.recursive(json_leaf, lambda c: st.one_of(st.lists(c, max_size=6), st.dictionaries(
    st.sampled_from(["cipher", "key", "blocks", "nonce", "counter", "boundary", "input", "toprf", "proof", "publicSignals", "ciphertext", "Cipher", "x"]), c, max_size=6)), max_leaves=25)


@settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(v=json_value, kind=st.sampled_from([REQUEST_WITNESS, PROVE, PROVE_BATCH, VERIFY, VERIFY_BATCH]), cut=st.integers(min_value=0, max_value=400))
def test_hypothesis_json_shapes(driver, v, kind, cut):
    """well-formed JSON of every shape the request / verify decoders may meet (right keys with wrong types, nested arrays, huge
    numbers, arbitrary base64), whole and truncated"""
    body = json.dumps(v).encode()
    driver.send(kind, body)
    driver.send(kind, body[:cut])


@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(b=st.binary(max_size=300), kind=st.sampled_from([REQUEST_WITNESS, PROVE_BATCH, VERIFY, INIT_VERIFIER, R1CS_STATS, PROVE_WITNESS]))
def test_hypothesis_raw_bytes(driver, b, kind):
    driver.send(kind, b)


_ORACLE_UNDER_ASAN = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import helpers as H
from oracle import ref, pairing, gnark_format as gf
ref.lib(); ref.set_threads(2)
rng = np.random.default_rng(5)
# field / hash / transforms / MSM entry points on small inputs, then the whole pipeline: solve -> setup -> prove -> verify (AES-128:
# lookups, countHints, the BSB22 commitment) and a violated witness
v = np.frombuffer(b"".join((int.from_bytes(rng.bytes(32), "little") % H.R).to_bytes(32, "little") for _ in range(64)), np.uint8).reshape(64, 32).copy()
assert np.array_equal(ref.ntt(ref.ntt(v), inverse=True), v)
alg = 1
inst = H.make_instance(alg, H.SplitMix64(3))
proof, out, prf, wit = H.oracle_proof_bytes(alg, inst)
vk = gf.vk_from_pk(H.oracle_pk(alg))
npub = H.circuit(alg).r1cs.nb_public - 1
assert pairing.groth16_verify(vk, gf.proof_parse(proof), wit[:npub])
bad = list(wit); bad[100] ^= 1
try:
    H.circuit(alg).solve([int(x) for x in bad]); raise SystemExit("violated witness solved")
except ValueError:
    pass
print("oracle under asan ok")
"""


def test_oracle_restatement_under_sanitizers():
    """the CPU restatement (oracle/oracle.c) under AddressSanitizer + UBSan: transforms, the gnark solver with lookups /
    countHints / BSB22, setup, a whole AES-128 proof and its verification, a violated witness -- no report"""
    import sys
    odir = os.path.join(H.ROOT, "oracle")
    subprocess.run(["make", "-s", "-C", odir, "asan"], check=True, capture_output=True, timeout=600)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan):
        pytest.skip("gcc's libasan.so not found")
    env = dict(os.environ, ORACLE_LIB=os.path.join(odir, "liboracle_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([sys.executable, "-c", _ORACLE_UNDER_ASAN.format(root=H.ROOT)], capture_output=True, text=True, timeout=1500, env=env)
    bad = [l for l in res.stderr.splitlines() if "AddressSanitizer" in l or "runtime error" in l]
    assert res.returncode == 0 and "oracle under asan ok" in res.stdout and not bad, res.stdout[-2000:] + res.stderr[-4000:]
