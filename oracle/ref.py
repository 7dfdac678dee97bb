"""ORACLE (test infrastructure only) -- ctypes front-end of oracle/liboracle.so.

The product path never imports this module; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do.  See oracle/oracle.c for the reference
citations and the parity status ("parity unpinned" for proof bytes).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import r1cs_format as rf

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_MOD = rf.R_MOD


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _LIB
    if _LIB is None:
        # ORACLE_LIB: another build of the same file (tests/test_sanitizers.py loads liboracle_asan.so, `make -C oracle asan`)
        so = os.environ.get("ORACLE_LIB") or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.oracle_fields_init()
        vp, i64, i32, cp = C.c_void_p, C.c_long, C.c_int, C.c_char_p
        L.oracle_circuit_new.restype = vp
        L.oracle_circuit_new.argtypes = [i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp]
        L.oracle_circuit_set_lookup.argtypes = [vp, i32, i64, vp]
        L.oracle_circuit_set_commitment.argtypes = [vp, i64, i64, vp]
        L.oracle_circuit_free.argtypes = [vp]
        L.oracle_setup.restype = vp
        L.oracle_setup.argtypes = [vp, vp]
        L.oracle_pk_free.argtypes = [vp]
        L.oracle_pk_count.restype = i64
        L.oracle_pk_count.argtypes = [vp, i32]
        L.oracle_pk_export_g1.argtypes = [vp, i32, vp]
        L.oracle_pk_export_g2.argtypes = [vp, i32, vp]
        L.oracle_pk_export_inf.argtypes = [vp, vp, vp]
        L.oracle_pk_domain.restype = i64
        L.oracle_pk_domain.argtypes = [vp]
        L.oracle_solve.restype = i32
        L.oracle_solve.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64]
        L.oracle_prove.restype = i32
        L.oracle_prove.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64]
        L.oracle_ntt_canonical.argtypes = [i64, vp, i32, i32]
        L.oracle_compute_h_canonical.argtypes = [i64, i64, vp, vp, vp, vp]
        for nm in ("oracle_msm_g1", "oracle_msm_g2", "oracle_msm_g1_naive", "oracle_msm_g2_naive"):
            getattr(L, nm).argtypes = [i64, vp, vp, vp]
        L.oracle_g1_fixed_base.argtypes = [i64, vp, vp]
        L.oracle_g2_fixed_base.argtypes = [i64, vp, vp]
        L.oracle_set_threads.argtypes = [i32]
        L.oracle_get_threads.restype = i32
        L.oracle_hash_to_fr.argtypes = [vp, i64, cp, vp]
        L.oracle_sha256.argtypes = [vp, i64, vp]
        for nm in ("oracle_fr_mul", "oracle_fp_mul"):
            getattr(L, nm).argtypes = [vp, vp, vp]
        L.oracle_fr_inv.argtypes = [vp, vp]
        L.oracle_g1_gen.argtypes = [vp]
        L.oracle_g2_gen.argtypes = [vp]
        _LIB = L
    return _LIB


def set_threads(n: int):
    lib().oracle_set_threads(int(n))


# ---------------------------------------------------------------- byte helpers
def ints_to_le32(vals) -> np.ndarray:
    """list of python ints -> (n,32) uint8 little-endian"""
    out = np.zeros((len(vals), 32), dtype=np.uint8)
    for i, v in enumerate(vals):
        out[i] = np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)
    return out


def le32_to_ints(arr) -> list:
    a = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, 32)
    return [int.from_bytes(a[i].tobytes(), "little") for i in range(a.shape[0])]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---------------------------------------------------------------- circuit / pk
class Circuit:
    def __init__(self, r1cs: rf.R1CS):
        L = lib()
        self.r1cs = r1cs
        kind = {"hint": 0, "r1c": 1, "lookup": 2}
        self._keep = dict(
            bp=np.asarray(r1cs.blueprint_id, dtype=np.uint32),
            co=np.asarray(r1cs.constraint_offset, dtype=np.uint32),
            wo=np.asarray(r1cs.wire_offset, dtype=np.uint32),
            sc=np.asarray(r1cs.start_calldata, dtype=np.uint64),
            cd=np.asarray(r1cs.calldata, dtype=np.uint32),
            cf=np.frombuffer(r1cs.coeffs_mont_raw, dtype=np.uint8).copy(),
            bk=np.asarray([kind[b[0]] for b in r1cs.blueprints], dtype=np.int32),
        )
        k = self._keep
        self.h = L.oracle_circuit_new(
            r1cs.nb_instructions, _ptr(k["bp"]), _ptr(k["co"]), _ptr(k["wo"]), _ptr(k["sc"]),
            len(r1cs.calldata), _ptr(k["cd"]), len(r1cs.coeffs), _ptr(k["cf"]),
            r1cs.nb_wires, r1cs.nb_public, r1cs.nb_secret, r1cs.nb_constraints,
            len(r1cs.blueprints), _ptr(k["bk"]))
        for i, (kd, payload) in enumerate(r1cs.blueprints):
            if kd == "lookup":
                e = np.asarray(payload, dtype=np.uint32)
                L.oracle_circuit_set_lookup(self.h, i, len(e), _ptr(e))
        if r1cs.commitment_info:
            assert len(r1cs.commitment_info) == 1
            ci = r1cs.commitment_info[0]
            assert ci["NbPublicCommitted"] == 0 and not ci["PublicAndCommitmentCommitted"]
            pc = np.asarray(ci["PrivateCommitted"], dtype=np.uint32)
            L.oracle_circuit_set_commitment(self.h, ci["CommitmentIndex"], len(pc), _ptr(pc))

    @property
    def n_inputs(self):
        return self.r1cs.nb_public - 1 + self.r1cs.nb_secret

    def solve(self, witness_ints, randomize=0, challenge_override=None, pk=None):
        """-> (wires, a, b, c) as (n,32) uint8 canonical LE arrays"""
        L = lib()
        r = self.r1cs
        w = ints_to_le32(witness_ints)
        assert w.shape[0] == self.n_inputs
        wires = np.zeros((r.nb_wires, 32), np.uint8)
        a = np.zeros((r.nb_constraints, 32), np.uint8)
        b = np.zeros_like(a)
        c = np.zeros_like(a)
        rnd = ints_to_le32([randomize])
        co = ints_to_le32([challenge_override]) if challenge_override is not None else None
        err = C.create_string_buffer(256)
        rc = L.oracle_solve(self.h, pk.h if pk else None, _ptr(w), _ptr(rnd), _ptr(co), _ptr(wires), _ptr(a), _ptr(b), _ptr(c), err, 256)
        if rc:
            raise ValueError(err.value.decode())
        return wires, a, b, c

    def __del__(self):
        try:
            lib().oracle_circuit_free(self.h)
        except Exception:
            pass


def toxic_from_seed(seed: int):
    """Deterministic toxic waste (alpha,beta,gamma,delta,tau,sigma) from SplitMix64."""
    import hashlib
    out = []
    for i in range(6):
        d = hashlib.sha256(b"zkprove-mi355x-setup|%d|%d" % (seed, i)).digest() + hashlib.sha256(b"x|%d|%d" % (seed, i)).digest()
        v = int.from_bytes(d, "big") % (R_MOD - 2) + 2
        out.append(v)
    return out


class ProvingKey:
    """Synthetic single-party Groth16 setup over a Circuit (the reference's pk.* are absent from the mount)."""

    def __init__(self, circuit: Circuit, seed: int = 1):
        self.circuit = circuit
        self.toxic = toxic_from_seed(seed)
        tw = ints_to_le32(self.toxic)
        self.h = lib().oracle_setup(circuit.h, _ptr(tw))

    def count(self, which):
        return lib().oracle_pk_count(self.h, which)

    def g1(self, which) -> np.ndarray:
        n = self.count(which)
        out = np.zeros((n, 64), np.uint8)
        lib().oracle_pk_export_g1(self.h, which, _ptr(out))
        return out

    def g2_B(self) -> np.ndarray:
        n = self.count(8)
        out = np.zeros((n, 128), np.uint8)
        lib().oracle_pk_export_g2(self.h, 8, _ptr(out))
        return out

    def g2_misc(self) -> np.ndarray:
        out = np.zeros((5, 128), np.uint8)   # beta2, delta2, gamma2, ped_g, ped_gsigneg
        lib().oracle_pk_export_g2(self.h, 9, _ptr(out))
        return out

    def infinity_flags(self):
        n = self.circuit.r1cs.nb_wires
        a = np.zeros(n, np.uint8)
        b = np.zeros(n, np.uint8)
        lib().oracle_pk_export_inf(self.h, _ptr(a), _ptr(b))
        return a, b

    @property
    def domain(self):
        return lib().oracle_pk_domain(self.h)

    def prove(self, witness_ints, r, s, randomize=0, want_aux=False):
        """-> dict(Ar=(x,y), Bs=((x0,x1),(y0,y1)), Krs, commitment, pok, timings[, wires, h])"""
        L = lib()
        w = ints_to_le32(witness_ints)
        assert w.shape[0] == self.circuit.n_inputs
        proof = np.zeros(384, np.uint8)
        tim = np.zeros(5, np.float64)
        rr, ss, rnd = ints_to_le32([r]), ints_to_le32([s]), ints_to_le32([randomize])
        wires = np.zeros((self.circuit.r1cs.nb_wires, 32), np.uint8) if want_aux else None
        h = np.zeros((self.domain, 32), np.uint8) if want_aux else None
        err = C.create_string_buffer(256)
        rc = L.oracle_prove(self.circuit.h, self.h, _ptr(w), _ptr(rr), _ptr(ss), _ptr(rnd), _ptr(proof), _ptr(tim), _ptr(wires), _ptr(h), err, 256)
        if rc:
            raise ValueError(err.value.decode())
        v = le32_to_ints(proof)
        out = dict(Ar=(v[0], v[1]), Bs=((v[2], v[3]), (v[4], v[5])), Krs=(v[6], v[7]),
                   commitment=(v[8], v[9]), pok=(v[10], v[11]), timings=tim.tolist())
        if want_aux:
            out["wires"] = wires
            out["h"] = h
        return out

    def __del__(self):
        try:
            lib().oracle_pk_free(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------- kernels
def ntt(vals_le32: np.ndarray, inverse=False, coset=False) -> np.ndarray:
    a = np.ascontiguousarray(vals_le32, dtype=np.uint8).copy()
    lib().oracle_ntt_canonical(a.shape[0], _ptr(a), int(inverse), int(coset))
    return a


def compute_h(a, b, c, N) -> np.ndarray:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    c = np.ascontiguousarray(c, np.uint8)
    out = np.zeros((N, 32), np.uint8)
    lib().oracle_compute_h_canonical(a.shape[0], N, _ptr(a), _ptr(b), _ptr(c), _ptr(out))
    return out


def msm_g1(points: np.ndarray, scalars: np.ndarray, naive=False) -> np.ndarray:
    points = np.ascontiguousarray(points, np.uint8)
    scalars = np.ascontiguousarray(scalars, np.uint8)
    out = np.zeros(64, np.uint8)
    (lib().oracle_msm_g1_naive if naive else lib().oracle_msm_g1)(points.shape[0], _ptr(points), _ptr(scalars), _ptr(out))
    return out


def msm_g2(points: np.ndarray, scalars: np.ndarray, naive=False) -> np.ndarray:
    points = np.ascontiguousarray(points, np.uint8)
    scalars = np.ascontiguousarray(scalars, np.uint8)
    out = np.zeros(128, np.uint8)
    (lib().oracle_msm_g2_naive if naive else lib().oracle_msm_g2)(points.shape[0], _ptr(points), _ptr(scalars), _ptr(out))
    return out


def g1_fixed_base(scalars: np.ndarray) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, np.uint8)
    out = np.zeros((scalars.shape[0], 64), np.uint8)
    lib().oracle_g1_fixed_base(scalars.shape[0], _ptr(scalars), _ptr(out))
    return out


def g2_fixed_base(scalars: np.ndarray) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, np.uint8)
    out = np.zeros((scalars.shape[0], 128), np.uint8)
    lib().oracle_g2_fixed_base(scalars.shape[0], _ptr(scalars), _ptr(out))
    return out


def hash_to_fr(msg: bytes, dst: bytes) -> int:
    out = np.zeros(32, np.uint8)
    m = np.frombuffer(msg, np.uint8).copy() if msg else np.zeros(1, np.uint8)
    lib().oracle_hash_to_fr(_ptr(m), len(msg), dst, _ptr(out))
    return int.from_bytes(out.tobytes(), "little")
