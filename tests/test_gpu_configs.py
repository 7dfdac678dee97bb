"""BASELINE.json's configurations at their full sizes on the HIP path, through the C ABI (task 1):
config 1 (the fixed RFC 8439 vector of chacha_test.go:123-152 through Prove), config 3 (AES-256-CTR, resident batch of
1024), the per-GPU share of config 4 (8192 ChaCha20 requests through ProveBatch), and the in-library device sharding
(two replicas) that lets the unchanged callers use every GPU of a node."""
import base64
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import gnark_format as gf
from oracle import pairing
from oracle import r1cs_format as rf

import helpers as H

pytestmark = pytest.mark.gpu


def _fresh(zk, alg, seed=1):
    """keys from the product's own setup with the oracle fixture's toxic waste (both provers then share a key)"""
    from oracle import ref
    zk.lib.ZkpAllowUnpinnedKeys(1)
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, ref.toxic_from_seed(seed), raw=True)
    zk.lib.ZkpReset()
    assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)


def test_config1_rfc8439_vector_through_prove(zk, oracle):
    """BASELINE config 1 / SURVEY 8d-1: key 00..1f, nonce 00 00 00 00 00 00 00 4a 00 00 00 00, counters 1,2, the 128-byte bPt / bCt
    literals of gnark/circuits/chachaV3/chacha_test.go:123-152.  Prove's ciphertext is pinned by the reference; with fixed blinding
    the proof bytes equal the oracle's, and a second run with PRNG blinding verifies."""
    _fresh(zk, 0)
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
    inst = dict(key=key, nonces=[nonce, nonce], counters=[1, 2], data=pt, r=1, s=1, randomize=0)
    res = json.loads(zk.prove_deterministic_raw(H.request_json(0, inst), 1, 1, 0))
    assert base64.b64decode(res["ciphertext"]) == ct                           # the reference-held known answer
    exp, out, *_ = H.oracle_proof_bytes(0, inst)
    assert out == ct and base64.b64decode(res["proof"]) == exp                  # r = s = 1: bit-identical to the oracle
    blocks = [(nonce, 1), (nonce, 2)]
    assert zk.verify("chacha20", exp, ct, blocks, pt)
    rnd = zk.prove(json.loads(H.request_json(0, inst)))                         # urandom blinding
    assert rnd["proof"] != res["proof"] and base64.b64decode(rnd["ciphertext"]) == ct
    assert zk.verify("chacha20", base64.b64decode(rnd["proof"]), ct, blocks, pt)
    zk.lib.ZkpReset()


def test_config3_aes256_full_size_batch_every_proof_verifies(zk, oracle):
    """BASELINE config 3 at full size (AES-256-CTR, resident batch 1024, seven pipelined passes == one single pass): no solver error, 1024 distinct
    proofs, every one passes the product's verifier, a proof does not verify against its neighbour's public inputs, and
    instances 0 / 511 / 1023 are bit-identical to the oracle's proofs (BSB22 commitment and its proof of knowledge included)."""
    from concurrent.futures import ThreadPoolExecutor
    alg, B = 2, 1024
    _fresh(zk, alg)
    rng = H.SplitMix64(0x5EED0000 + 2)
    insts = [H.make_instance(alg, rng) for _ in range(B)]
    wits, outs = [], []
    for inst in insts:
        _, w, out = zk.request_witness(H.request_json(alg, inst))
        wits.append(w); outs.append(out)
    rb = zk.ResidentBatch(alg, np.array(wits, np.uint32), [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
    rb.step()                                   # unpipelined reference pass
    single, errs = rb.fetch()
    assert not errs.any()
    rb.steps(7)                                 # every working set and both ahead-prepared copies reused at the bench's size
    proofs, errs = rb.fetch()
    assert not errs.any()
    assert np.array_equal(proofs, single), "pipelined passes differ from the single pass in records %s" % np.nonzero((proofs != single).any(axis=1))[0][:10]
    info = np.zeros(8, np.int64)
    assert zk.lib.zkp_quotient_msm_info(alg, info.ctypes.data) == 0
    assert info[0] == 1 and info[5] == 1 and info[6] > 1000, "a 1024-batch of the AES circuit must run the quotient comb and the wire combs: %s" % info
    ser = [zk.proof_serialize(proofs[k], True) for k in range(B)]
    assert len(set(ser)) == B and all(len(s) == 196 for s in ser)

    def check(k):
        return zk.verify(H.CIPHER[alg], ser[k], outs[k], list(zip(insts[k]["nonces"], insts[k]["counters"])), insts[k]["data"])
    with ThreadPoolExecutor(max_workers=16) as ex:
        ok = list(ex.map(check, range(B)))
    assert all(ok), "instances failing verification: %s" % [k for k, v in enumerate(ok) if not v][:10]
    assert not zk.verify(H.CIPHER[alg], ser[0], outs[1], list(zip(insts[0]["nonces"], insts[0]["counters"])), insts[0]["data"])
    for k in (0, 511, 1023):
        exp, out, *_ = H.oracle_proof_bytes(alg, insts[k])
        assert ser[k] == exp and outs[k] == out
    zk.lib.ZkpReset()


@pytest.mark.parametrize("alg,B", [(0, 128), (1, 128)])
def test_pipelined_steps_rotate_working_sets_and_list_copies(zk, oracle, alg, B):
    """resident_prove_pipelined (csrc/prover.hip) prepares batches two ahead: three working sets, two copies of the quotient MSM's ahead-prepared
    data (comb patterns by default at this batch size -- and, for AES-128, the wire combs after their calibration solve; the sorted bucket list
    in the ZKPROVE_COMB=0 re-run below), solve / transforms / patterns of later batches beside the MSMs of earlier ones.  Seven steps take every set and both list copies
    through at least two rounds; every step proves the same resident inputs, so the last batch must be byte-identical to ONE
    unpipelined step (which uses none of that machinery) -- and instance 0 to the oracle."""
    _fresh(zk, alg)
    rng = H.SplitMix64(0x5EED0000 + 70 + alg)
    insts = [H.make_instance(alg, rng) for _ in range(B)]
    wits, outs = [], []
    for inst in insts:
        _, w, out = zk.request_witness(H.request_json(alg, inst))
        wits.append(w); outs.append(out)
    args = (alg, np.array(wits, np.uint32), [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
    rb = zk.ResidentBatch(*args)
    rb.step()
    ref, errs = rb.fetch()
    assert not errs.any()
    ref = [zk.proof_serialize(ref[k], alg != 0) for k in range(B)]
    for nsteps in (7, 1, 2):
        rb.steps(nsteps)
        got, errs = rb.fetch()
        assert not errs.any()
        assert [zk.proof_serialize(got[k], alg != 0) for k in range(B)] == ref, "pipelined run of %d steps differs from the single step" % nsteps
    exp, out, *_ = H.oracle_proof_bytes(alg, insts[0])
    assert ref[0] == exp and outs[0] == out
    zk.lib.ZkpReset()


def test_config4_per_gpu_share_8192_requests_through_prove_batch(zk, oracle):
    """BASELINE config 4 = 65 536 ChaCha20 proofs over 8 GPUs = 8 192 per GPU: one ProveBatch call of 8 192 requests (eight
    device chunks of 1024, software-pipelined).  Order preserved, every ciphertext right, every 16th proof plus the chunk
    edges verified against ITS request (and not against the neighbour's), the last proof also under the oracle's pairing."""
    from concurrent.futures import ThreadPoolExecutor
    alg, n = 0, 8192
    _fresh(zk, alg)
    rng = H.SplitMix64(0x5EED0000 + 4)
    insts = [H.make_instance(alg, rng) for _ in range(n)]
    body = ("[" + ",".join(H.request_json(alg, i).decode() for i in insts) + "]").encode()
    res = json.loads(zk.prove_batch_raw(body))
    assert len(res) == n and all(isinstance(r, dict) and "proof" in r for r in res)
    outs = [zk.request_witness(H.request_json(alg, i))[2] for i in insts]
    assert all(base64.b64decode(res[k]["ciphertext"]) == outs[k] for k in range(n))
    assert len({r["proof"] for r in res}) == n
    picks = sorted(set(list(range(0, n, 16)) + [c * 1024 + d for c in range(8) for d in (0, 1, 1022, 1023)]))

    def check(k):
        blocks = list(zip(insts[k]["nonces"], insts[k]["counters"]))
        good = zk.verify("chacha20", base64.b64decode(res[k]["proof"]), outs[k], blocks, insts[k]["data"])
        cross = zk.verify("chacha20", base64.b64decode(res[(k + 1) % n]["proof"]), outs[k], blocks, insts[k]["data"])
        return good and not cross
    with ThreadPoolExecutor(max_workers=16) as ex:
        ok = list(ex.map(check, picks))
    assert all(ok), "requests failing: %s" % [picks[i] for i, v in enumerate(ok) if not v][:10]
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    wit, _ = H.oracle_witness(alg, insts[n - 1])
    assert pairing.groth16_verify(vk, gf.proof_parse(base64.b64decode(res[n - 1]["proof"])), wit[:2304])
    zk.lib.ZkpReset()


def test_config4_all_65536_proofs_eight_rank_streams(zk, oracle):
    """BASELINE config 4 at its FULL size on one GPU: the eight ranks' shares, 8 x 8 192 = 65 536 independent ChaCha20 proofs, each
    share one ProveBatch call over its own SplitMix64 stream (0x5EED0000 + 1 + 1000 r, the streams bench.py gives its ranks).  All
    65 536 responses carry the right ciphertext and pairwise different proofs; from every share every 64th proof and the chunk edges
    verify against THEIR request; the first and the last proof of every share verify under the oracle's independent pairing.  (An
    8-GPU node runs the same eight shares side by side; independence of the chunks: js/src/zk.ts:94-177.)"""
    from concurrent.futures import ThreadPoolExecutor
    alg, per, ranks = 0, 8192, 8
    _fresh(zk, alg)
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    seen = set()
    total = 0
    for r in range(ranks):
        rng = H.SplitMix64(0x5EED0000 + 1 + 1000 * r)
        insts = [H.make_instance(alg, rng) for _ in range(per)]
        body = ("[" + ",".join(H.request_json(alg, i).decode() for i in insts) + "]").encode()
        res = json.loads(zk.prove_batch_raw(body))
        assert len(res) == per and all(isinstance(x, dict) and "proof" in x for x in res)
        outs = [zk.request_witness(H.request_json(alg, i))[2] for i in insts]
        assert all(base64.b64decode(res[k]["ciphertext"]) == outs[k] for k in range(per))
        proofs = {x["proof"] for x in res}
        assert len(proofs) == per and not (proofs & seen)
        seen |= proofs
        total += per
        picks = sorted(set(list(range(0, per, 64)) + [c * 1024 + d for c in range(8) for d in (0, 1023)]))

        def check(k):
            blocks = list(zip(insts[k]["nonces"], insts[k]["counters"]))
            return zk.verify("chacha20", base64.b64decode(res[k]["proof"]), outs[k], blocks, insts[k]["data"])
        with ThreadPoolExecutor(max_workers=16) as ex:
            ok = list(ex.map(check, picks))
        assert all(ok), "rank %d: requests failing: %s" % (r, [picks[i] for i, v in enumerate(ok) if not v][:10])
        for k in (0, per - 1):
            wit, _ = H.oracle_witness(alg, insts[k])
            assert pairing.groth16_verify(vk, gf.proof_parse(base64.b64decode(res[k]["proof"])), wit[:2304]), "rank %d proof %d" % (r, k)
    assert total == 65536 and len(seen) == 65536
    zk.lib.ZkpReset()

@pytest.mark.parametrize("alg,B", [(0, 1024), (1, 160)])
def test_verify_batch_on_the_gpu_agrees_with_verify(zk, oracle, alg, B):
    """VerifyBatch (csrc/verifier.cpp): the requests of a circuit are checked with one pairing-product equation under random
    coefficients -- decompression, G2 subgroup checks, kSum as a batched MSM over vk.K, the BSB22 challenges and one Miller loop per
    proof on the device, one final exponentiation on the host.  A valid batch answers all-true from the device check alone; a batch
    with tampered members falls back to per-proof verification and answers exactly as Verify does (verifier/impl/verifiers.go:73-89,
    142-158); malformed and foreign elements answer false without disturbing the others."""
    _fresh(zk, alg)
    rng = H.SplitMix64(0x5EED0000 + 90 + alg)
    insts = [H.make_instance(alg, rng) for _ in range(B)]
    wits, outs = [], []
    for inst in insts:
        _, w, out = zk.request_witness(H.request_json(alg, inst))
        wits.append(w); outs.append(out)
    rb = zk.ResidentBatch(alg, np.array(wits, np.uint32), [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
    rb.step()
    proofs, errs = rb.fetch()
    assert not errs.any()
    ser = [zk.proof_serialize(proofs[k], alg != 0) for k in range(B)]
    reqs = [zk.verify_request(H.CIPHER[alg], ser[k], outs[k], list(zip(insts[k]["nonces"], insts[k]["counters"])), insts[k]["data"]) for k in range(B)]
    res, ms = zk.verify_batch_raw(json.dumps(reqs).encode(), timed=True)
    assert res == [True] * B and ms > 0, "device batch check declined or failed: %s" % [k for k, v in enumerate(res) if not v][:10]
    print("verify_batch alg %d: %d proofs, device %.1f ms" % (alg, B, ms))
    # tampered members: a proof under its neighbour's public inputs, a flipped ciphertext byte, a proof with A and Krs swapped
    bad = [dict(r) for r in reqs]
    bad[3] = dict(reqs[3], proof=reqs[4]["proof"])
    ct = bytearray(outs[7]); ct[0] ^= 1
    bad[7] = zk.verify_request(H.CIPHER[alg], ser[7], bytes(ct), list(zip(insts[7]["nonces"], insts[7]["counters"])), insts[7]["data"])
    sw = bytearray(ser[B - 1]); sw[0:32], sw[96:128] = ser[B - 1][96:128], ser[B - 1][0:32]
    bad[B - 1] = zk.verify_request(H.CIPHER[alg], bytes(sw), outs[B - 1], list(zip(insts[B - 1]["nonces"], insts[B - 1]["counters"])), insts[B - 1]["data"])
    expect = [zk.verify_raw(json.dumps(r).encode()) for r in (bad[3], bad[7], bad[B - 1])]
    assert expect == [False, False, False]
    res = zk.verify_batch_raw(json.dumps(bad).encode())
    assert res == [k not in (3, 7, B - 1) for k in range(B)]
    # malformed / foreign elements
    mixed = [reqs[0], 5, {"cipher": "rot13"}, reqs[1], {"cipher": H.CIPHER[alg], "proof": "AAAA", "publicSignals": {}}] + reqs[2:20]
    assert zk.verify_batch_raw(json.dumps(mixed).encode()) == [True, False, False, True, False] + [True] * 18
    assert zk.verify_batch_raw(b"[]") == [] and zk.verify_batch_raw(b"{") == []
    zk.lib.ZkpReset()

def test_bucket_pipeline_of_large_batches_stays_bit_exact():
    """Batches of 128 and more take the comb path by default (csrc/msm_comb.hip); the sort -> bucket walk -> reduction pipeline with its
    ahead-sorted, double-buffered entry lists remains the product's path when the tables do not fit (and under ZKPROVE_COMB=0), so the
    pipelined-steps test above is repeated with the comb switched off (own process: the switch is read once)."""
    env = dict(os.environ, ZKPROVE_COMB="0")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "pipelined_steps_rotate", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=H.ROOT)
    assert res.returncode == 0 and " passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]

@pytest.mark.gpu
@pytest.mark.parametrize("switches", [
    {"ZKPROVE_NTT_STREAM": "solve", "ZKPROVE_SOLVER_PACK": "0"},          # round-2 arrangement: transforms behind the solve stage on its stream, a workgroup per instruction
    {"ZKPROVE_COPY_PRIO": "lo", "ZKPROVE_SOLVER_OCC": "3"},               # what the AES programs get by default, on the ChaCha20 program
    {"ZKPROVE_COPY_PRIO": "hi", "ZKPROVE_SOLVER_OCC": "4"},
])
def test_scheduling_switches_do_not_change_a_byte(switches):
    """Where the transforms run (copy stream beside the next solve stage / solve stream behind it), the copy stream's priority, the solver
    kernel's register budget and the packing of short instructions four to a workgroup are scheduling choices: the pipelined-steps test
    (every working set, both pattern copies, byte-identical to the single step and the oracle) passes under each of them.  Own process:
    the switches are read once."""
    exp = os.path.join(H.ROOT, "zk-symmetric-crypto_amd", "libzkprove_exp.so")      # the A/B switches are read by the experiments build only
    assert os.path.exists(exp), "libzkprove_exp.so missing: run `make -C zk-symmetric-crypto_amd experiments`"
    env = dict(os.environ, ZKPROVE_LIB=exp, **switches)
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "pipelined_steps_rotate", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=H.ROOT)
    assert res.returncode == 0 and " passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]

def test_streamed_prove_batch_answers_bad_members_in_place(zk, oracle):
    """A streamed ProveBatch call (more than one device chunk) with bad members in different chunks: a request with a short key, one for
    another cipher, one that is not an object and one with broken JSON answer exactly what Prove answers for them, at their positions;
    every other request gets its proof (spot-verified) and its ciphertext."""
    alg, n = 0, 2100
    _fresh(zk, alg)
    rng = H.SplitMix64(0x5EED0000 + 321)
    insts = [H.make_instance(alg, rng) for _ in range(n)]
    parts = [H.request_json(alg, i).decode() for i in insts]
    short = json.loads(parts[100]); short["key"] = base64.b64encode(b"k" * 16).decode(); parts[100] = json.dumps(short)
    other = json.loads(parts[1500]); other["cipher"] = "aes-128-ctr"; parts[1500] = json.dumps(other)
    parts[1024] = "17"
    parts[2050] = '{"cipher" "chacha20"}'
    body = ("[" + ",".join(parts) + "]").encode()
    res = json.loads(zk.prove_batch_raw(body))
    assert len(res) == n
    assert res[100] == json.loads(zk.prove_raw(parts[100].encode())) == "key length must be 32: 16"
    assert res[1500] == json.loads(zk.prove_raw(parts[1500].encode())) == "proving params are not initialized for cipher: aes-128-ctr"
    assert res[1024] == {} and res[2050] == {"Offset": body.index(b'{"cipher" "chacha20"}') + 11}
    bad = {100, 1024, 1500, 2050}
    for k in range(n):
        if k in bad:
            continue
        assert isinstance(res[k], dict) and "proof" in res[k], (k, res[k])
    for k in (0, 99, 101, 1023, 1025, 1499, 1501, 2049, 2051, n - 1):
        out = zk.request_witness(parts[k].encode())[2]
        assert base64.b64decode(res[k]["ciphertext"]) == out
        assert zk.verify("chacha20", base64.b64decode(res[k]["proof"]), out, list(zip(insts[k]["nonces"], insts[k]["counters"])), insts[k]["data"])
    zk.lib.ZkpReset()

@pytest.mark.parametrize("alg", [0, 1])
def test_comb_paths_under_another_proving_key(zk, oracle, alg):
    """The comb tables (quotient bases; for AES also the calibrated wire positions of six base sets) are functions of the proving key.
    Everything above uses the key of toxic seed 1; here a key of seed 2 at the smallest batch that takes the comb paths (130 > 128):
    instances 0, 64 and 129 bit-identical to the oracle under ITS seed-2 key, the whole batch accepted by VerifyBatch."""
    B = 130
    _fresh(zk, alg, seed=2)
    rng = H.SplitMix64(0x5EED0000 + 400 + alg)
    insts = [H.make_instance(alg, rng) for _ in range(B)]
    wits, outs = [], []
    for inst in insts:
        _, w, out = zk.request_witness(H.request_json(alg, inst))
        wits.append(w); outs.append(out)
    rb = zk.ResidentBatch(alg, np.array(wits, np.uint32), [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
    rb.steps(3)
    proofs, errs = rb.fetch()
    assert not errs.any()
    info = np.zeros(8, np.int64)
    assert zk.lib.zkp_quotient_msm_info(alg, info.ctypes.data) == 0 and info[0] == 1 and (alg == 0 or info[5] == 1), info
    ser = [zk.proof_serialize(proofs[k], alg != 0) for k in range(B)]
    for k in (0, 64, B - 1):
        exp, out, *_ = H.oracle_proof_bytes(alg, insts[k], seed=2)
        assert ser[k] == exp and outs[k] == out, k
    reqs = [zk.verify_request(H.CIPHER[alg], ser[k], outs[k], list(zip(insts[k]["nonces"], insts[k]["counters"])), insts[k]["data"]) for k in range(B)]
    assert zk.verify_batch_raw(json.dumps(reqs).encode()) == [True] * B
    zk.lib.ZkpReset()

_VALVE_SCRIPT = r"""
import importlib, json, os, sys
import numpy as np
import torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
zk = importlib.import_module("zk-symmetric-crypto_amd")
import helpers as H
from oracle import ref
zk.lib.ZkpAllowUnpinnedKeys(1)
alg, B = 0, 256
r1cs = zk.synth.r1cs_bytes(alg)
pk, vk = zk.setup(r1cs, ref.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs)
rng = H.SplitMix64(0x5EED0000 + 55)
insts = [H.make_instance(alg, rng) for _ in range(B)]
wits = np.array([zk.request_witness(H.request_json(alg, i))[1] for i in insts], np.uint32)
rb = zk.ResidentBatch(alg, wits, [i["r"] for i in insts], [i["s"] for i in insts], [0] * B)      # reserves one working set and builds the comb tables
info = np.zeros(8, np.int64)
zk.lib.zkp_quotient_msm_info(alg, info.ctypes.data)
had_comb, table_mib = int(info[0]), int(info[3])
rb.step()
single, errs = rb.fetch()
assert not errs.any()
# take (almost) all the memory that is left: the second and third working set of the pipelined entry cannot be allocated any more
free_b, total_b = torch.cuda.mem_get_info()
hog = torch.empty(max(0, free_b - (1 << 30)), dtype=torch.uint8, device="cuda")
rb.steps(4)
got, errs = rb.fetch()
zk.lib.zkp_quotient_msm_info(alg, info.ctypes.data)
print(json.dumps(dict(had_comb=had_comb, table_mib=table_mib, comb_after=int(info[0]), same=bool(np.array_equal(got, single)), ok=bool(not errs.any()), hog_gib=hog.numel() >> 30)))
"""


def test_comb_tables_are_given_back_under_memory_pressure():
    """The comb tables are an optimisation, the working sets are not: when the pipelined entry cannot allocate its further working sets
    because the device is full (here: a tensor of the host application takes what is left), the algorithm releases its comb tables
    (prover.hip, comb_release), continues on the table-free bucket pipeline and produces the same proofs."""
    res = subprocess.run([sys.executable, "-c", _VALVE_SCRIPT.format(root=H.ROOT)], capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["had_comb"] == 1 and out["table_mib"] > 1000, out
    assert out["comb_after"] == 0 and out["same"] and out["ok"], out
    assert "comb tables" in res.stderr and "released" in res.stderr

_OOM_SCRIPT = r"""
import importlib, json, os, sys
import numpy as np
import torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
zk = importlib.import_module("zk-symmetric-crypto_amd")
import helpers as H
from oracle import ref
zk.lib.ZkpAllowUnpinnedKeys(1)
alg = 0
r1cs = zk.synth.r1cs_bytes(alg)
pk, vk = zk.setup(r1cs, ref.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs)
rng = H.SplitMix64(0x5EED0000 + 77)
insts = [H.make_instance(alg, rng) for _ in range(1024)]
wits = np.array([zk.request_witness(H.request_json(alg, i))[1] for i in insts], np.uint32)
def small():
    rb = zk.ResidentBatch(alg, wits[:8], [i["r"] for i in insts[:8]], [i["s"] for i in insts[:8]], [0] * 8)
    rb.step()
    return rb.fetch()
first, errs = small()
assert not errs.any()
free_b, total_b = torch.cuda.mem_get_info()
hog = torch.empty(max(0, free_b - (5 << 29)), dtype=torch.uint8, device="cuda")      # 2.5 GiB stay free: enough for 8 proofs, not for 1 024
failed = False
try:
    zk.ResidentBatch(alg, wits, [i["r"] for i in insts], [i["s"] for i in insts], [0] * 1024)
except RuntimeError:
    failed = True
again, errs2 = small()                                  # with the device still full
one = json.loads(zk.prove_raw(H.request_json(alg, insts[0])))
del hog
torch.cuda.empty_cache()
rb = zk.ResidentBatch(alg, wits[:256], [i["r"] for i in insts[:256]], [i["s"] for i in insts[:256]], [0] * 256)      # and grows again once there is room
rb.step()
big, errs3 = rb.fetch()
print(json.dumps(dict(failed=failed, same=bool(np.array_equal(first, again)), ok=bool(not errs2.any() and not errs3.any()), prove="proof" in one,
                      grown_same=bool(np.array_equal(big[:8], first)))))
"""


def test_a_failed_large_reservation_leaves_the_library_usable():
    """DevBuf::alloc releases before it allocates, so a reservation that fails half way leaves EMPTY buffers behind; the capacities the
    prover remembers are reset with them (prover.hip, resident_reserve).  Small batch -> a 1024-batch that cannot be reserved (the
    device is filled from the host application's side) answers with an error -> the same small batch and a Prove() still work, with the
    same proofs, instead of launching kernels on null pointers -> a larger batch is reserved again once memory is back."""
    res = subprocess.run([sys.executable, "-c", _OOM_SCRIPT.format(root=H.ROOT)], capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out == dict(failed=True, same=True, ok=True, prove=True, grown_same=True), (out, res.stderr[-2000:])


_THREE_SCRIPT = r"""
import base64, importlib, json, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
zk = importlib.import_module("zk-symmetric-crypto_amd")
import helpers as H
from oracle import ref
zk.lib.ZkpAllowUnpinnedKeys(1)
def comb(alg):
    info = np.zeros(8, np.int64)
    zk.lib.zkp_quotient_msm_info(alg, info.ctypes.data)
    return int(info[0])
order = (0, 2, 1)                       # ChaCha20, AES-256, then AES-128 arrives on a full device
for alg in order:
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, ref.toxic_from_seed(1), raw=True)
    assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)
out = {{"ok": {{}}, "comb_before": {{}}, "comb_after": {{}}}}
for rnd in range(2):
    for alg in order:
        rng = H.SplitMix64(4242 + alg)
        insts = [H.make_instance(alg, rng) for _ in range(2048)]
        res = json.loads(zk.prove_batch_raw(("[" + ",".join(H.request_json(alg, i).decode() for i in insts) + "]").encode()))
        good = 0
        for k in range(len(insts)):
            g = isinstance(res[k], dict) and "proof" in res[k]
            if g and (k % 256 == 0 or k == len(insts) - 1):
                o = zk.request_witness(H.request_json(alg, insts[k]))[2]
                g = base64.b64decode(res[k]["ciphertext"]) == o and zk.verify(H.CIPHER[alg], base64.b64decode(res[k]["proof"]), o, list(zip(insts[k]["nonces"], insts[k]["counters"])), insts[k]["data"])
            good += bool(g)
        out["ok"]["%d.%d" % (rnd, alg)] = good
        if rnd == 0 and alg == 2: out["comb_before"] = {{str(a): comb(a) for a in (0, 2)}}
out["comb_after"] = {{str(a): comb(a) for a in order}}
print(json.dumps(out))
"""

def test_a_third_circuit_takes_the_idle_neighbours_tables():
    """Three circuits with 1024-chunks on one GPU, the first two with tables of a FIXED clump size (ZKPROVE_COMB_K=18: 28 + 57 GiB, and 57 more for the third -- since
    round 5 three circuits fit one MI355X at the one-third table shares the budget would give them, so the pressure is made by pinning
    the size): tables and working sets of ChaCha20 and AES-256 fill the device, AES-128's working sets do not fit and it has no
    tables of its own to give back.  Its request must not fail for the sake of another circuit's optimisation: the least recently used
    idle neighbour gives its comb tables back (cabi.cpp release_neighbours, prover.hip relieve_pressure) and continues on the bucket
    pipeline (tables that went under pressure may be rebuilt at a later call when memory allows: comb_second_chance); every request of
    every circuit is answered in both rounds."""
    res = subprocess.run([sys.executable, "-c", _THREE_SCRIPT.format(root=H.ROOT)], capture_output=True, text=True, timeout=1500,
                         env=dict(os.environ, ZKPROVE_COMB_K="18"))
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert all(v == 2048 for v in out["ok"].values()) and len(out["ok"]) == 6, (out, res.stderr[-3000:])
    assert out["comb_before"] == {"0": 1, "2": 1}, out
    assert "comb tables" in res.stderr and "released" in res.stderr, (out, res.stderr[-3000:])

def test_two_ciphers_streamed_at_the_same_time():
    """Two host threads, one streamed ProveBatch call each (ChaCha20 and AES-256, 3 000 requests = three device chunks), at the same time,
    twice: the algorithms share the device, the pinned chunk pool and (HIP multiplexes streams) hardware queues -- every request of both
    calls is answered, the sampled proofs verify (scripts/debug/two_ciphers_concurrent.py)."""
    res = subprocess.run([sys.executable, os.path.join(H.ROOT, "scripts", "debug", "two_ciphers_concurrent.py")], capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2, res.stdout[-2000:]
    for ln in lines:
        assert ln["alg0"] == dict(ln["alg0"], ok=3000, of=3000) and ln["alg2"] == dict(ln["alg2"], ok=3000, of=3000), ln

_SHARD_SCRIPT = r"""
import base64, importlib, json, os, sys, threading
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
zk = importlib.import_module("zk-symmetric-crypto_amd")
import helpers as H
from oracle import ref
zk.lib.ZkpAllowUnpinnedKeys(1)
alg = {alg}
r1cs = zk.synth.r1cs_bytes(alg)
pk, vk = zk.setup(r1cs, ref.toxic_from_seed(1), raw=True)
assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)
zk.lib.ZkpDeviceReplicas.restype = int
nrep = zk.lib.ZkpDeviceReplicas(alg)
rng = H.SplitMix64(31337)
n = 300
insts = [H.make_instance(alg, rng) for _ in range(n)]
res = json.loads(zk.prove_batch_raw(("[" + ",".join(H.request_json(alg, i).decode() for i in insts) + "]").encode()))
ok = 0
for k in range(n):
    out = zk.request_witness(H.request_json(alg, insts[k]))[2]
    blocks = list(zip(insts[k]["nonces"], insts[k]["counters"]))
    ok += bool(base64.b64decode(res[k]["ciphertext"]) == out and zk.verify(H.CIPHER[alg], base64.b64decode(res[k]["proof"]), out, blocks, insts[k]["data"]))
# a call large enough to be STREAMED (cabi.cpp prove_batch_json): chunks of ~n/(2 replicas x 2) pulled by both replicas from one feed
m = {stream_n}
insts2 = [H.make_instance(alg, rng) for _ in range(m)]
res2 = json.loads(zk.prove_batch_raw(("[" + ",".join(H.request_json(alg, i).decode() for i in insts2) + "]").encode()))
ok2 = 0
for k in range(m):
    out = zk.request_witness(H.request_json(alg, insts2[k]))[2]
    good = base64.b64decode(res2[k]["ciphertext"]) == out
    if k % 7 == 0 or k in (m - 1, m - 2):
        blocks = list(zip(insts2[k]["nonces"], insts2[k]["counters"]))
        good = good and zk.verify(H.CIPHER[alg], base64.b64decode(res2[k]["proof"]), out, blocks, insts2[k]["data"])
    ok2 += bool(good)
# Prove from a thread that never touched HIP before (Node's libuv pool, a goroutine's OS thread): the library binds the device itself
box = []
inst = insts[7]
t = threading.Thread(target=lambda: box.append(zk.prove_deterministic_raw(H.request_json(alg, inst), inst["r"], inst["s"], inst["randomize"])))
t.start(); t.join()
exp, out, *_ = H.oracle_proof_bytes(alg, inst)
got = json.loads(box[0])
print(json.dumps(dict(replicas=nrep, verified=ok, n=n, streamed_ok=ok2, streamed_n=m, thread_proof_matches_oracle=base64.b64decode(got["proof"]) == exp)))
"""


@pytest.mark.parametrize("alg", [0, 2])
def test_in_library_sharding_over_two_device_replicas(alg):
    """The drop-in library scales without its callers knowing about devices: with more than one device it keeps a replica of the
    algorithm on each and ProveBatch cuts its request array into contiguous slices, one host thread per device (cabi.cpp,
    prove_prepared).  A 1-GPU box exercises exactly that code with ZKPROVE_DEVICES=0,0 (two replicas on device 0): every one
    of 300 proofs must verify against ITS request.  Also: Prove from a brand-new thread (hipSetDevice is per thread)."""
    env = dict(os.environ, ZKPROVE_DEVICES="0,0")
    env.pop("LOCAL_RANK", None)
    res = subprocess.run([sys.executable, "-c", _SHARD_SCRIPT.format(root=H.ROOT, alg=alg, stream_n=2500 if alg == 0 else 1300)], env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["replicas"] == 2
    assert out["verified"] == out["n"] == 300
    assert out["streamed_ok"] == out["streamed_n"]          # the streamed call: every ciphertext right, every 7th proof verified
    assert out["thread_proof_matches_oracle"]


def test_bench_two_real_ranks_prove_gather_and_verify():
    """`python bench.py --gpus 2` end to end with two PROVING ranks: the launcher starts them, each proves its own batch, the proof
    records are gathered and rank 0 verifies instance 0 of EVERY rank's slice against that rank's seed stream.  The box has one GPU, so
    both ranks use device 0 and the collectives run over gloo (ZKP_BENCH_BACKEND / ZKP_BENCH_ONE_DEVICE); on a multi-GPU node the same
    code path runs with one GPU per rank over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ZKP_BENCH_BACKEND="gloo", ZKP_BENCH_ONE_DEVICE="1")
    res = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--batch", "96", "--steps", "2", "--warmup", "1",
                          "--cpu-proofs", "0", "--master-port", str(31500 + os.getpid() % 2000)], env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["proofs_total"] == 2 * 96 * 2
    assert out["gathered_proof_records"] == 2 * 96 and out["solver_ok"] == 2 * 96
    assert out["gather_spot_check_verified"] == "2/2"                 # one verified proof from each rank's slice
    assert len(out["per_rank_proofs_per_s"]) == 2 and out["value"] > 0
    aes = out["aes256_ctr"]
    assert aes["gathered_proof_records"] == 2 * 96 and aes["gather_spot_check_verified"] == "2/2" and aes["solver_ok"] == 2 * 96
