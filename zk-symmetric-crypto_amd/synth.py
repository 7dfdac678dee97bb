"""Synthetic workload generator of the product (bench.py, smoke, tests): SplitMix64 request streams
(SURVEY.md section 8d), deterministic toxic waste for ZkpSetup, and the loader of the R1CS artefacts
(`resources/gnark/r1cs.*`, shipped xz-compressed under tests/golden with their pinned SHA-256)."""
import base64
import hashlib
import json
import lzma
import os

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
NAMES = {0: "chacha20", 1: "aes128", 2: "aes256"}
CIPHER = {0: "chacha20", 1: "aes-128-ctr", 2: "aes-256-ctr"}
CIRCUIT_SHA256 = {  # gnark/libraries/prover/impl/library.go:38,43,48
    "chacha20": "517cdf7642c84dcb165a29f0accea413178fc11b21d8998461156a1acce6ebbd",
    "aes128": "8ef58f13706568eed44f9aa40b97da7c2f2a7f1bce47b7bfe52f7df860a049fb",
    "aes256": "7f589a6d1a24efa6d01f663c11b671e0abef196c15cadb93d2c6b226df7bf0dd",
}
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def r1cs_bytes(alg: int) -> bytes:
    name = NAMES[alg]
    data = lzma.open(os.path.join(_ROOT, "tests", "golden", "r1cs.%s.xz" % name)).read()
    assert hashlib.sha256(data).hexdigest() == CIRCUIT_SHA256[name]
    return data


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
        return int.from_bytes(self.bytes(40), "little") % R_MOD


def make_instance(alg: int, rng: SplitMix64):
    """key, nonce, counter, plaintext, r, s, randomize -- consumed in this order"""
    key = rng.bytes(32 if alg != 1 else 16)
    nonce = rng.bytes(12)
    nblk = 2 if alg == 0 else 5
    ctr = min(rng.next() & 0xFFFFFFFF, 0xFFFFFFFF - nblk)
    data = rng.bytes(128 if alg == 0 else 80)
    r, s, rnd = rng.fr(), rng.fr(), rng.fr()
    return dict(key=key, nonces=[nonce] * nblk, counters=[ctr + i for i in range(nblk)], data=data, r=r, s=s, randomize=rnd)


def request_json(alg: int, inst) -> bytes:
    b64 = lambda b: base64.b64encode(b).decode()
    return json.dumps({
        "cipher": CIPHER[alg], "key": b64(inst["key"]),
        "blocks": [{"nonce": b64(n), "counter": c, "boundary": None} for n, c in zip(inst["nonces"], inst["counters"])],
        "input": b64(inst["data"]), "toprf": {},
    }).encode()


def toxic_from_seed(seed: int):
    """Deterministic toxic waste (alpha, beta, gamma, delta, tau, sigma) for synthetic keys."""
    out = []
    for i in range(6):
        d = hashlib.sha256(b"zkprove-mi355x-setup|%d|%d" % (seed, i)).digest() + hashlib.sha256(b"x|%d|%d" % (seed, i)).digest()
        out.append(int.from_bytes(d, "big") % (R_MOD - 2) + 2)
    return out
