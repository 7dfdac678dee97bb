"""Batch-size boundaries of the size-switched code paths (VERDICT r4 task 1; ADVICE r4, msm.hip part2 sizing).

The prover switches kernels and scratch layouts by batch size: sparse split of the wire MSMs (<= 4), tiny-batch bucket reduction and
tree folds (<= 8), solver workgroup shapes (<= 8 / <= 16), 16-entry accumulation (<= 32), side streams / bucket pipeline (<= 64), lane
program and pipelined tail (<= 128), comb tables from 128, whole-wavefront paths above.  Every threshold is crossed here from both
sides, for the resident-batch entry AND the JSON entry (ProveBatch with caller-supplied blinding), against the CPU oracle:
  * proofs of instances 0, B/2, B-1 byte-identical to the oracle's, their solved wires identical to the oracle's;
  * all B proofs of the resident path byte-identical to the JSON path's (two different host paths, one device pipeline);
  * all B proofs verify through VerifyBatch, ciphertexts equal the oracle's.
Reference behaviour: random inputs of any count (gnark/libraries/test_helpers.go:292-411), 8 proofs per call (js/src/tests/benchmark.ts:14-34).
Capacities, pattern slots and partial-sum strides are sized by the largest batch a process has seen, so call SEQUENCES that cross the
paths in one process are part of the test (1 -> 129 -> 8 -> 1024 -> 5 -> 33; 10 -> 8 in a fresh library: the sequence ADVICE r4 names)."""
import base64
import json

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

SIZES = [2, 5, 8, 9, 16, 17, 32, 33, 64, 127, 128, 129, 255, 256]
# second pass, once the batch of 128 has built the comb tables: from 16 proofs on every batch now takes the comb path (quotient walk with
# (plane, proof) lanes, range levels of the fold, wire combs, transforms beside the wire MSMs), 15 and below keep the bucket pipeline
SIZES_WITH_TABLES = [15, 16, 17, 31, 33, 64, 65, 100, 127, 8]
POOL = 256


class Case:
    """one circuit: keys shared by the product library, the test library and the oracle; a pool of instances with cached oracle proofs"""

    def __init__(self, zk, alg, with_testlib=True):
        from oracle import ref
        self.zk, self.alg = zk, alg
        zk.lib.ZkpAllowUnpinnedKeys(1)
        r1cs = zk.synth.r1cs_bytes(alg)
        pk, vk = zk.setup(r1cs, ref.toxic_from_seed(1), raw=True)
        zk.lib.ZkpReset()
        assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)
        self.tl = None
        if with_testlib:
            self.tl = zk.testlib
            self.tl.ZkpAllowUnpinnedKeys(1)
            self.tl.ZkpReset()
            assert zk.init_algorithm(alg, pk, r1cs, lib=self.tl)
        rng = H.SplitMix64(0xB0DA0000 + alg)
        self.pool = [H.make_instance(alg, rng) for _ in range(POOL)]
        self.wit = {}
        self.orc = {}
        self.r1 = H.circuit(alg).r1cs

    def close(self):
        self.zk.lib.ZkpReset()
        if self.tl is not None and self.tl is not self.zk.lib:
            self.tl.ZkpReset()

    def witness(self, i):
        if i not in self.wit:
            _, w, out = self.zk.request_witness(H.request_json(self.alg, self.pool[i]))
            self.wit[i] = (w, out)
        return self.wit[i]

    def oracle(self, i):
        if i not in self.orc:
            pb, out, prf, _ = H.oracle_proof_bytes(self.alg, self.pool[i], want_aux=True)
            self.orc[i] = (pb, out, prf["wires"])
        return self.orc[i]

    def members(self, B, salt=0):
        off = (B * 7 + salt * 13) % POOL
        return [(off + k) % POOL for k in range(B)]

    def json_batch(self, idx):
        """ProveBatch with caller-supplied blinding through the DROP-IN library -> serialized proofs; ciphertexts checked"""
        zk, alg = self.zk, self.alg
        insts = [self.pool[i] for i in idx]
        body = b"[" + b",".join(H.request_json(alg, it) for it in insts) + b"]"
        res = json.loads(zk.prove_batch_deterministic_raw(body, [it["r"] for it in insts], [it["s"] for it in insts], [it["randomize"] for it in insts]))
        assert isinstance(res, list) and len(res) == len(idx), res
        proofs = []
        for k, r in enumerate(res):
            assert isinstance(r, dict) and "proof" in r, "request %d of %d answered %r" % (k, len(idx), r)
            assert base64.b64decode(r["ciphertext"]) == self.witness(idx[k])[1]
            proofs.append(base64.b64decode(r["proof"]))
        return proofs

    def verify_all(self, idx, proofs):
        zk, alg = self.zk, self.alg
        reqs = [zk.verify_request(H.CIPHER[alg], proofs[k], self.witness(i)[1], list(zip(self.pool[i]["nonces"], self.pool[i]["counters"])), self.pool[i]["data"])
                for k, i in enumerate(idx)]
        res = zk.verify_batch_raw(json.dumps(reqs).encode())
        assert res == [True] * len(idx), "VerifyBatch rejects %s of a batch of %d" % ([k for k, v in enumerate(res) if not v][:8], len(idx))

    def check_against_oracle(self, idx, proofs, picks):
        for k in picks:
            pb, out, _ = self.oracle(idx[k])
            assert out == self.witness(idx[k])[1]
            assert proofs[k] == pb, "proof %d of a batch of %d differs from the oracle's" % (k, len(idx))

    def resident_batch(self, idx, picks):
        """the resident-batch entry (test library: same objects) -> serialized proofs; solved wires of `picks` against the oracle"""
        zk, alg = self.zk, self.alg
        insts = [self.pool[i] for i in idx]
        rb = zk.ResidentBatch(alg, np.array([self.witness(i)[0] for i in idx], np.uint32), [it["r"] for it in insts], [it["s"] for it in insts],
                              [it["randomize"] for it in insts], lib=self.tl)
        rb.step()
        proofs, errs = rb.fetch()
        assert not errs.any(), errs
        wires, _ = rb.debug(self.r1.nb_wires, self.r1.domain_size(), instances=picks, want_h=False)
        for q, k in enumerate(picks):
            assert np.array_equal(wires[q], self.oracle(idx[k])[2]), "solved wires of instance %d of a batch of %d differ from the oracle's" % (k, len(idx))
        return [zk.proof_serialize(proofs[k], alg != 0) for k in range(len(idx))]


def _picks(B):
    return sorted({0, B // 2, B - 1})


@pytest.mark.parametrize("alg", [0, 2, 1])
def test_every_batch_size_threshold_from_both_sides(zk, oracle, alg):
    c = Case(zk, alg)
    try:
        for B in SIZES + SIZES_WITH_TABLES:
            idx, picks = c.members(B), _picks(B)
            via_json = c.json_batch(idx)
            c.check_against_oracle(idx, via_json, picks)
            c.verify_all(idx, via_json)
            via_resident = c.resident_batch(idx, picks)
            assert via_resident == via_json, "resident and JSON entries disagree at B = %d (first at %d)" % (
                B, next(k for k in range(B) if via_resident[k] != via_json[k]))
    finally:
        c.close()


@pytest.mark.parametrize("alg", [0, 2])
def test_call_sequence_that_crosses_the_paths_in_one_process(zk, oracle, alg):
    """capacities, pattern slots and partial-sum strides are those of the LARGEST batch seen so far: small batches after large ones,
    large after small, the comb tables appearing in the middle"""
    c = Case(zk, alg, with_testlib=False)
    try:
        for step, B in enumerate([1, 129, 8, 1024, 5, 33, 2, 256, 16]):
            idx, picks = c.members(B, salt=step), _picks(B)
            proofs = c.json_batch(idx)
            c.check_against_oracle(idx, proofs, picks)
            c.verify_all(idx, proofs)
    finally:
        c.close()


@pytest.mark.parametrize("first,then", [(10, 8), (9, 7), (5, 4), (12, 3), (15, 1)])
def test_a_tiny_batch_after_a_slightly_larger_first_call(zk, oracle, first, then):
    """ADVICE r4 (msm.hip part2): a FIRST call of 9..15 proofs sized the G2 partial-sum scratch for B * stride; a later batch of <= 8
    takes the finer bucket reduction (up to twice the slots per instance) and answered {} for every request.  AES-256: the G2 wire MSM
    with c = 13 is the one whose tiny stride exceeds the regular one."""
    c = Case(zk, 2, with_testlib=False)
    try:
        for step, B in enumerate((first, then)):
            idx = c.members(B, salt=step)
            proofs = c.json_batch(idx)
            c.check_against_oracle(idx, proofs, [0, B - 1])
            c.verify_all(idx, proofs)
    finally:
        c.close()
