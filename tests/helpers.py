"""Shared test helpers: synthetic requests (SplitMix64) and synthetic proving keys."""
import base64
import json
import os

import numpy as np

from oracle import gnark_format as gf
from oracle import r1cs_format as rf
from oracle import ref
from oracle import witness as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "tests", "_cache")
R = rf.R_MOD
NAMES = {0: "chacha20", 1: "aes128", 2: "aes256"}
CIPHER = {0: "chacha20", 1: "aes-128-ctr", 2: "aes-256-ctr"}


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & (2 ** 64 - 1)

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        return z ^ (z >> 31)

    def bytes(self, n):
        out = b""
        while len(out) < n:
            out += self.next().to_bytes(8, "little")
        return out[:n]

    def fr(self):
        return int.from_bytes(self.bytes(40), "little") % R


def make_instance(alg: int, rng: SplitMix64):
    """key, nonce, counter, plaintext, r, s, randomize -- consumed in this order"""
    key = rng.bytes(32 if alg != 1 else 16)
    nonce = rng.bytes(12)
    ctr = rng.next() & 0xFFFFFFFF
    nblk = 2 if alg == 0 else 5
    ctr = min(ctr, 0xFFFFFFFF - nblk)
    data = rng.bytes(128 if alg == 0 else 80)
    r, s, rnd = rng.fr(), rng.fr(), rng.fr()
    counters = [ctr + i for i in range(nblk)]
    return dict(key=key, nonces=[nonce] * nblk, counters=counters, data=data, r=r, s=s, randomize=rnd)


def request_json(alg: int, inst) -> bytes:
    b64 = lambda b: base64.b64encode(b).decode()
    return json.dumps({
        "cipher": CIPHER[alg], "key": b64(inst["key"]),
        "blocks": [{"nonce": b64(n), "counter": c, "boundary": None} for n, c in zip(inst["nonces"], inst["counters"])],
        "input": b64(inst["data"]), "toprf": {},
    }).encode()


def oracle_witness(alg: int, inst):
    f = W.chacha_witness if alg == 0 else W.aes_witness
    return f(inst["key"], inst["nonces"], inst["counters"], inst["data"])


_circuits = {}
_pks = {}


def circuit(alg: int):
    if alg not in _circuits:
        _circuits[alg] = ref.Circuit(rf.load_reference_r1cs(NAMES[alg]))
    return _circuits[alg]


def oracle_pk(alg: int, seed: int = 1):
    if (alg, seed) not in _pks:
        _pks[(alg, seed)] = ref.ProvingKey(circuit(alg), seed=seed)
    return _pks[(alg, seed)]


def pk_file_bytes(alg: int, seed: int = 1, raw: bool = True) -> bytes:
    """synthetic proving key in gnark's ProvingKey.WriteRawTo / WriteTo layout (cached on disk)"""
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "pk.%s.seed%d.%s" % (NAMES[alg], seed, "raw" if raw else "comp"))
    if os.path.exists(path):
        return open(path, "rb").read()
    data = gf.pk_bytes(oracle_pk(alg, seed), raw=raw)
    tmp = "%s.tmp.%d" % (path, os.getpid())      # ranks of a multi-GPU run may build the fixture concurrently
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)
    return data


def oracle_proof_bytes(alg: int, inst, seed: int = 1, want_aux=False):
    wit, out = oracle_witness(alg, inst)
    pk = oracle_pk(alg, seed)
    prf = pk.prove(wit, inst["r"], inst["s"], inst["randomize"], want_aux=want_aux)
    pb = gf.proof_bytes(prf, has_commitment=(alg != 0))
    return pb, out, prf, wit
