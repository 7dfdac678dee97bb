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


import importlib as _il

_synth = _il.import_module("zk-symmetric-crypto_amd.synth")      # product-side synthetic workload generator
SplitMix64, make_instance, request_json = _synth.SplitMix64, _synth.make_instance, _synth.request_json


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
