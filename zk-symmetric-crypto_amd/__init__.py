"""zk-symmetric-crypto_amd -- MI355X-native Groth16 prover for the gnark ChaCha20 / AES-CTR circuits.

Python host mirror of the reference's gnark operator boundary
(js/src/gnark/utils.ts:47-136 koffi bindings; gnark/libraries/prover/libprove.go:18-48):
the product is the C-ABI shared library `libzkprove.so` (HIP, gfx950); this module only loads it
with ctypes and exposes the same calls.  There is NO CPU fallback: if the HIP library is missing
the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKPROVE_LIB: another build of the same library (tests load libzkprove_exp.so, the -DZK_EXPERIMENTS build, through it; that build
# exports the kernel-level test entries too, so it then also stands in for the test library)
LIB_PATH = os.environ.get("ZKPROVE_LIB") or os.path.join(_HERE, "libzkprove.so")
# kernel-level entry points of the parity tests and micro-benchmarks (include/zkprove_test.h): a library of their own, linked from the
# same objects as the product + csrc/test_api.hip.  The drop-in exports include/zkprove.h and nothing else.
TESTLIB_PATH = os.environ.get("ZKPROVE_LIB") or os.environ.get("ZKPROVE_TESTLIB") or os.path.join(_HERE, "libzkprove_test.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libzkprove.so (HIP extension) not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback in the product path)" % LIB_PATH)

lib = C.CDLL(LIB_PATH)


class GoSlice(C.Structure):  # js/src/gnark/utils.ts:47-51
    _fields_ = [("data", C.c_void_p), ("len", C.c_longlong), ("cap", C.c_longlong)]


class ProveReturn(C.Structure):  # js/src/gnark/utils.ts:53-56
    _fields_ = [("r0", C.c_void_p), ("r1", C.c_longlong)]


def _sig(name, restype, argtypes, of=None):
    fn = getattr(of if of is not None else lib, name, None)
    if fn is not None:
        fn.restype = restype
        fn.argtypes = argtypes
    return fn


_vp, _i64, _i32 = C.c_void_p, C.c_long, C.c_int
_sig("zkp_last_error", C.c_char_p, [])
_sig("zkp_device_count", _i32, [])

_testlib = None


def _load_testlib():
    global _testlib
    if _testlib is None:
        if not os.path.exists(TESTLIB_PATH):
            raise ImportError("libzkprove_test.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'`" % TESTLIB_PATH)
        t = lib if os.path.abspath(TESTLIB_PATH) == os.path.abspath(LIB_PATH) else C.CDLL(TESTLIB_PATH)
        for name, res, args in (
                ("zkp_last_error", C.c_char_p, []),
                ("zkp_ntt", _i32, [_i32, _i64, _vp, _i32]),
                ("zkp_compute_h", _i32, [_i32, _i64, _i64, _vp, _vp, _vp, _vp]),
                ("zkp_msm_g1", _i32, [_i64, _i64, _vp, _vp, _vp, _i32]),
                ("zkp_msm_g2", _i32, [_i64, _i64, _vp, _vp, _vp, _i32]),
                ("zkp_msm_g1_dense", _i32, [_i64, _i64, _vp, _vp, _vp, _i32]),
                ("zkp_msm_comb", _i32, [_i64, _i64, _vp, _vp, _vp, _i32, _i32]),
                ("zkp_msm_wire_comb", _i32, [_i32, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _vp]),
                ("zkp_bench_comb", _i32, [_i32, _i64, _i32, _i32, _vp]),
                ("zkp_fold_quotient_bases", _i32, [_i32, _i64, _vp, _vp, _vp]),
                ("zkp_fold_output_matrix", _i64, [_i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
                ("zkp_microbench", C.c_double, [_i32, _i64, _i32]),
                ("zkp_field_inv", _i32, [_i32, _i64, _vp, _vp]),
                ("zkp_bench_to_coset", C.c_double, [_i32, _i64, _i32]),
                ("zkp_debug_tree", _i32, [_i32, _i32, _i32, _i32]),
                ("zkp_debug_fetch", _i32, [_i32, _vp, _i64, _vp, _vp]),
                ("zkp_debug_leftovers", _i32, [_i32, _vp]),
                ("zkp_debug_solver_plan", _i32, [_i32, _vp]),
                ("zkp_pairing_product_check", _i32, [_vp, _vp, _i64, _i32]),
                ("zkp_vk_pairing_identities", _i32, [_vp, _i64, _i32]),
                ("zkp_expand_message_xmd", _i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32]),
                ("zkp_bsb22_challenge", _i32, [_vp, _i64, _vp]),
                ("zkp_bsb22_challenge_host", _i32, [_vp, _i64, _vp]),
                ("zkp_request_decode_both", _i32, [_vp, _i64]),
                ("zkp_host_stage_bench", _i32, [_vp, _i64, _i32, _vp])):
            _sig(name, res, args, of=t)
        if t is not lib:
            _bind_product(t)          # the test library exports include/zkprove.h too (same objects): a resident batch may live in it
        _testlib = t
    return _testlib


def __getattr__(name):          # zk.testlib: loaded on first use (the product path never needs it)
    if name == "testlib":
        return _load_testlib()
    raise AttributeError(name)


def last_error() -> str:
    """last error text of this thread, from whichever of the two libraries reported one"""
    msg = lib.zkp_last_error().decode()
    if not msg and _testlib is not None and _testlib is not lib:
        msg = _testlib.zkp_last_error().decode()
    return msg


def device_count() -> int:
    return lib.zkp_device_count()

# --------------------------------------------------------------------------- reference operator API
import json as _json

import numpy as _np

ALGORITHM_IDS = {"chacha20": 0, "aes-128-ctr": 1, "aes-256-ctr": 2}      # impl/library.go:17-24, js/src/config.ts
FILE_EXT = {"chacha20": "chacha20", "aes-128-ctr": "aes128", "aes-256-ctr": "aes256"}   # js/src/gnark/operator.ts:6-12

_u8p = C.c_void_p


def _bind_product(l):
    """ctypes signatures of include/zkprove.h on library handle l (the drop-in; the test library exports the same entries)"""
    _sig("enforce_binding", None, [], of=l)
    _sig("InitAlgorithm", C.c_ubyte, [C.c_ubyte, GoSlice, GoSlice], of=l)
    _sig("Prove", ProveReturn, [GoSlice], of=l)
    _sig("ProveBatch", ProveReturn, [GoSlice], of=l)
    _sig("Free", None, [C.c_void_p], of=l)
    _sig("GenerateOPRFRequestData", ProveReturn, [GoSlice], of=l)
    _sig("TOPRFFinalize", ProveReturn, [GoSlice], of=l)
    _sig("ZkpAllowUnpinnedKeys", None, [_i32], of=l)
    _sig("ZkpSetDevice", _i32, [_i32], of=l)
    _sig("ZkpReset", None, [], of=l)
    _sig("VerifyBatch", ProveReturn, [GoSlice], of=l)
    _sig("zkp_verify_batch_timed", ProveReturn, [GoSlice, C.POINTER(C.c_double)], of=l)
    _sig("VFree", None, [C.c_void_p], of=l)
    _sig("ZkpProveDeterministic", ProveReturn, [GoSlice, _u8p, _u8p, _u8p], of=l)
    _sig("ZkpProveBatchDeterministic", ProveReturn, [GoSlice, _i64, _u8p, _u8p, _u8p], of=l)
    _sig("ZkpProveWitness", ProveReturn, [C.c_ubyte, GoSlice], of=l)
    _sig("ZkpProveWitnessDeterministic", ProveReturn, [C.c_ubyte, GoSlice, _u8p, _u8p, _u8p], of=l)
    _sig("ZkpSetHardwareQueues", _i32, [_i32], of=l)
    _sig("zkp_n_inputs", _i64, [_i32], of=l)
    _sig("zkp_bench_upload", _i32, [_i32, _i64, _vp, _vp, _vp, _vp], of=l)
    _sig("zkp_bench_step", _i32, [_i32, _vp], of=l)
    _sig("zkp_bench_fetch", _i32, [_i32, _vp, _vp], of=l)
    _sig("zkp_bench_steps", _i32, [_i32, _i32, _vp], of=l)
    _sig("zkp_quotient_msm_info", _i32, [_i32, _vp], of=l)
    _sig("ZkpSetSplit", _i32, [_i32, _i32], of=l)
    _sig("zkp_split_partials", _i32, [_i32, _vp], of=l)
    _sig("zkp_split_finish", _i32, [_i32, _i32, _vp], of=l)
    _sig("ZkpDeviceReplicas", _i32, [_i32], of=l)
    _sig("zkp_request_witness", _i32, [GoSlice, _vp, _i64, _vp, _i64, C.POINTER(C.c_long), C.POINTER(C.c_long)], of=l)
    _sig("zkp_last_panic", C.c_char_p, [], of=l)
    _sig("zkp_proof_serialize", _i64, [_vp, _i32, _vp, _i64], of=l)
    _sig("zkp_r1cs_stats", _i32, [GoSlice, _vp], of=l)


_bind_product(lib)


def _slice(b: bytes):
    buf = C.create_string_buffer(b, len(b))
    return GoSlice(C.cast(buf, C.c_void_p), len(b), len(b)), buf


def _take(ret: ProveReturn) -> bytes:
    out = C.string_at(ret.r0, ret.r1)
    lib.Free(ret.r0)                      # js/src/gnark/utils.ts:223
    return out


def init_algorithm(algorithm_id: int, proving_key: bytes, r1cs: bytes, lib=None) -> bool:
    """lib.initAlgorithm(id, pk, r1cs)  (js/src/gnark/utils.ts:111-136); lib: another handle of the library (zk.testlib), default the drop-in"""
    s1, k1 = _slice(proving_key)
    s2, k2 = _slice(r1cs)
    return bool((lib or globals()["lib"]).InitAlgorithm(algorithm_id, s1, s2))


def prove_raw(request_json: bytes) -> bytes:
    s, keep = _slice(request_json)
    return _take(lib.Prove(s))


def prove(request: dict) -> dict:
    """ZKOperator.groth16Prove for the gnark engine (js/src/gnark/operator.ts:22-32): returns the decoded
    response; raises if the library answered with a panic value instead of a proof."""
    res = _json.loads(prove_raw(_json.dumps(request).encode()))
    if not isinstance(res, dict) or "proof" not in res:
        raise RuntimeError("prove failed: %r" % (res,))
    return res


def prove_deterministic_raw(request_json: bytes, r: int, s: int, randomize: int = 0) -> bytes:
    sl, keep = _slice(request_json)
    bufs = [C.create_string_buffer(int(v).to_bytes(32, "little"), 32) for v in (r, s, randomize)]
    return _take(lib.ZkpProveDeterministic(sl, *[C.cast(b, C.c_void_p) for b in bufs]))


def witness_binary(n_public: int, values) -> bytes:
    """gnark witness.MarshalBinary layout: u32 nbPublic | u32 nbSecret | u32 n | n x 32-byte big-endian elements (public first, no
    constant-one wire) -- what the Go shim passes for groth16.Prove(r1cs, pk, wtns) (impl/provers.go:175-179)"""
    vals = [int(v) for v in values]
    n = len(vals)
    return (n_public.to_bytes(4, "big") + (n - n_public).to_bytes(4, "big") + n.to_bytes(4, "big") + b"".join(v.to_bytes(32, "big") for v in vals))


def prove_witness(algorithm_id: int, witness: bytes, r=None, s=None, randomize=None) -> bytes:
    """ZkpProveWitness (the inner boundary groth16.Prove(r1cs, pk, wtns)): gnark binary witness -> Proof.WriteTo bytes; raises
    RuntimeError(error text) where groth16.Prove returns an error.  r / s / randomize given: the deterministic variant."""
    sl, keep = _slice(witness)
    if r is None and s is None and randomize is None:
        ret = lib.ZkpProveWitness(algorithm_id, sl)
    else:
        bufs = [C.create_string_buffer(int(v or 0).to_bytes(32, "little"), 32) for v in (r, s, randomize)]
        ret = lib.ZkpProveWitnessDeterministic(algorithm_id, sl, *[C.cast(b, C.c_void_p) for b in bufs])
    n = ret.r1
    out = C.string_at(ret.r0, abs(n))
    lib.Free(ret.r0)
    if n < 0:
        raise RuntimeError(out.decode())
    return out


def prove_batch_raw(requests_json: bytes) -> bytes:
    s, keep = _slice(requests_json)
    return _take(lib.ProveBatch(s))


def prove_batch_deterministic_raw(requests_json: bytes, r, s, randomize) -> bytes:
    """ZkpProveBatchDeterministic: ProveBatch with one (r, s, randomize) triple per request of the array"""
    sl, keep = _slice(requests_json)
    le = lambda vals: _np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), _np.uint8).copy()
    rr, ss, nn = le(r), le(s), le(randomize)
    assert len(rr) == len(ss) == len(nn)
    return _take(lib.ZkpProveBatchDeterministic(sl, len(rr) // 32, rr.ctypes.data, ss.ctypes.data, nn.ctypes.data))


def request_witness(request_json: bytes):
    """-> (algorithm id, witness assignment uint32[n], output bytes); raises ValueError(panic json)"""
    s, keep = _slice(request_json)
    inp = _np.zeros(4096, _np.uint32)
    out = _np.zeros(256, _np.uint8)
    ni, no = C.c_long(0), C.c_long(0)
    rc = lib.zkp_request_witness(s, inp.ctypes.data, inp.size, out.ctypes.data, out.size, C.byref(ni), C.byref(no))
    if rc < 0:
        raise ValueError(lib.zkp_last_panic().decode())
    return rc, inp[:ni.value].copy(), out[:no.value].tobytes()


def r1cs_stats(r1cs: bytes):
    s, keep = _slice(r1cs)
    st = _np.zeros(8, _np.uint64)
    if lib.zkp_r1cs_stats(s, st.ctypes.data) != 0:
        raise ValueError("r1cs parse failed")
    return [int(v) for v in st]


class ResidentBatch:
    """Device-resident batch of one algorithm (bench.py / parity tests)."""

    def __init__(self, algorithm_id: int, inputs, r, s, randomize, lib=None):
        """lib: the library that holds the batch -- default the drop-in; zk.testlib (same objects + the kernel-level / debug entries,
        its own algorithm registry: InitAlgorithm there first) for the tests that read solved wires back"""
        lib = self.lib = lib or globals()["lib"]
        self.alg = algorithm_id
        self.inputs = _np.ascontiguousarray(inputs, _np.uint32)
        self.batch = self.inputs.shape[0]
        le = lambda vals: _np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), _np.uint8).copy()
        self._r, self._s, self._rnd = le(r), le(s), le(randomize)
        rc = lib.zkp_bench_upload(self.alg, self.batch, self.inputs.ctypes.data, self._r.ctypes.data, self._s.ctypes.data, self._rnd.ctypes.data)
        if rc != 0:
            raise RuntimeError("zkp_bench_upload failed")

    def step(self):
        t = _np.zeros(8, _np.float32)
        if self.lib.zkp_bench_step(self.alg, t.ctypes.data) != 0:
            raise RuntimeError("zkp_bench_step failed")
        return dict(zip(("solve", "commit", "ntt", "msm_g1", "msm_g2", "finalize", "total", "z_segments"), [float(x) for x in t]))

    def steps(self, n: int):
        """n passes back to back, solve(k+1) overlapped with NTT+MSM(k); returns phase times of the last pass
        ('total' = whole call / n)"""
        t = _np.zeros(8, _np.float32)
        if self.lib.zkp_bench_steps(self.alg, int(n), t.ctypes.data) != 0:
            raise RuntimeError("zkp_bench_steps failed")
        return dict(zip(("solve", "commit", "ntt", "msm_g1", "msm_g2", "finalize", "total", "z_segments"), [float(x) for x in t]))

    def fetch(self):
        proofs = _np.zeros((self.batch, 384), _np.uint8)
        errs = _np.zeros(self.batch, _np.uint32)
        if self.lib.zkp_bench_fetch(self.alg, proofs.ctypes.data, errs.ctypes.data) != 0:
            raise RuntimeError("zkp_bench_fetch failed")
        return proofs, errs

    def split_partials(self):
        """split mode (ZkpSetSplit): this rank's partial sums, [batch][384] uint8"""
        out = _np.zeros((self.batch, 384), _np.uint8)
        if self.lib.zkp_split_partials(self.alg, out.ctypes.data) != 0:
            raise RuntimeError("zkp_split_partials failed")
        return out

    def split_finish(self, gathered):
        g = _np.ascontiguousarray(gathered, _np.uint8)
        assert g.ndim == 3 and g.shape[1:] == (self.batch, 384)
        if self.lib.zkp_split_finish(self.alg, int(g.shape[0]), g.ctypes.data) != 0:
            raise RuntimeError("zkp_split_finish failed")

    def debug(self, n_wires, domain, instances=None, want_h=True):
        """solved wires [n][n_wires][32] and quotient-stage products [n][domain][32] of the listed instances (default: all) after a
        step -- only on a batch that lives in the test library (lib=zk.testlib): the drop-in does not export the debug entries"""
        if not hasattr(self.lib, "zkp_debug_fetch"):
            raise RuntimeError("ResidentBatch.debug needs lib=zk.testlib (the drop-in library exports no debug entries)")
        inst = None if instances is None else _np.ascontiguousarray(instances, _np.uint32)
        n = self.batch if inst is None else len(inst)
        wires = _np.zeros((n, n_wires, 32), _np.uint8)
        h = _np.zeros((n, domain, 32), _np.uint8) if want_h else None
        if self.lib.zkp_debug_fetch(self.alg, None if inst is None else inst.ctypes.data, n, wires.ctypes.data, None if h is None else h.ctypes.data) != 0:
            raise RuntimeError("zkp_debug_fetch failed")
        return wires, h


def proof_serialize(proof384, has_commitment: bool) -> bytes:
    p = _np.ascontiguousarray(proof384, _np.uint8)
    out = _np.zeros(256, _np.uint8)
    n = lib.zkp_proof_serialize(p.ctypes.data, int(has_commitment), out.ctypes.data, out.size)
    return out[:n].tobytes()


def shard_range(total: int, rank: int, world: int):
    """Contiguous slice of a batch owned by `rank` (independent proofs: no data-path collective)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_sig("ZkpSetup", _i32, [GoSlice, _vp, _i32, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong), C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)])


def setup(r1cs: bytes, toxic, raw: bool = True):
    """Groth16 key generation on the GPU (gnark/keygen/keygen.go:94-132) -> (pk bytes, vk bytes) in gnark's formats."""
    s, keep = _slice(r1cs)
    tox = _np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in toxic), _np.uint8).copy()
    pk, vk = C.c_void_p(), C.c_void_p()
    npk, nvk = C.c_longlong(0), C.c_longlong(0)
    if lib.ZkpSetup(s, tox.ctypes.data, int(raw), C.byref(pk), C.byref(npk), C.byref(vk), C.byref(nvk)) != 0:
        raise RuntimeError("ZkpSetup failed")
    out = (C.string_at(pk.value, npk.value), C.string_at(vk.value, nvk.value))
    lib.Free(pk)
    lib.Free(vk)
    return out


from . import synth  # noqa: E402  (synthetic workload generator used by bench.py / tests)


_sig("Verify", C.c_ubyte, [GoSlice])
_sig("VFree", None, [C.c_void_p])
_sig("ZkpInitVerifier", C.c_ubyte, [C.c_ubyte, GoSlice])


def init_verifier(algorithm_id: int, vk: bytes) -> bool:
    s, keep = _slice(vk)
    return bool(lib.ZkpInitVerifier(algorithm_id, s))


def verify_raw(request_json: bytes) -> bool:
    """lib.verify(json) (js/src/gnark/operator.ts:33-45 -> libverify.go:19)"""
    s, keep = _slice(request_json)
    return bool(lib.Verify(s))


def verify_request(cipher: str, proof: bytes, ciphertext: bytes, blocks, plaintext_input: bytes) -> dict:
    """the JSON object lib.verify takes (js/src/gnark/operator.ts:33-45)"""
    import base64 as _b64
    b = lambda x: _b64.b64encode(x).decode()
    return {"cipher": cipher, "proof": b(proof),
            "publicSignals": {"ciphertext": b(ciphertext), "blocks": [{"nonce": b(n), "counter": c} for n, c in blocks], "input": b(plaintext_input)}}


def verify_batch_raw(requests_json: bytes, timed: bool = False):
    """VerifyBatch: JSON array of Verify requests -> list of booleans (GPU batch check per circuit, host fallback per proof);
    timed=True also returns the device milliseconds of the batch check"""
    s, keep = _slice(requests_json)
    ms = C.c_double(0.0)
    ret = lib.zkp_verify_batch_timed(s, C.byref(ms))
    out = C.string_at(ret.r0, ret.r1)
    lib.VFree(ret.r0)
    res = _json.loads(out)
    return (res, ms.value) if timed else res


def verify(cipher: str, proof: bytes, ciphertext: bytes, blocks, plaintext_input: bytes) -> bool:
    import base64 as _b64
    b = lambda x: _b64.b64encode(x).decode()
    req = {"cipher": cipher, "proof": b(proof),
           "publicSignals": {"ciphertext": b(ciphertext), "blocks": [{"nonce": b(n), "counter": c} for n, c in blocks], "input": b(plaintext_input)}}
    return verify_raw(_json.dumps(req).encode())
