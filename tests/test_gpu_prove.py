"""GPU parity of the whole proving path through the C ABI (InitAlgorithm / Prove / batch API) against
the CPU oracle: solved wires, the quotient stage and final proof bytes are bit-identical for the same
(witness, r, s, randomize); proofs verify under the independent pairing check; tampering is rejected."""
import base64
import json

import numpy as np
import pytest

from oracle import gnark_format as gf
from oracle import pairing
from oracle import r1cs_format as rf

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def algs(zk):
    zk.lib.ZkpAllowUnpinnedKeys(1)
    blobs = {}

    def get(alg):
        # InitAlgorithm is idempotent (library.go:92-94); calling it every time keeps tests independent of ZkpReset
        if alg not in blobs:
            blobs[alg] = (H.pk_file_bytes(alg), rf.load_r1cs_bytes(H.NAMES[alg]))
        assert zk.init_algorithm(alg, *blobs[alg]), "InitAlgorithm(%d) failed" % alg
        return alg
    return get


@pytest.mark.parametrize("alg", [0, 1, 2])
def test_prove_matches_oracle_bit_exact(zk, oracle, algs, alg):
    algs(alg)
    rng = H.SplitMix64(0x5EED0000 + alg)
    inst = H.make_instance(alg, rng)
    res = json.loads(zk.prove_deterministic_raw(H.request_json(alg, inst), inst["r"], inst["s"], inst["randomize"]))
    assert "proof" in res, res
    exp_proof, exp_out, prf, wit = H.oracle_proof_bytes(alg, inst)
    assert base64.b64decode(res["ciphertext"]) == exp_out
    got = base64.b64decode(res["proof"])
    assert len(got) == (164 if alg == 0 else 196)
    assert got == exp_proof
    # the reference's own oracle: verify(prove(x)) is true, tampered public input is rejected
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    npub = H.circuit(alg).r1cs.nb_public - 1
    pp = gf.proof_parse(got)
    assert pairing.groth16_verify(vk, pp, wit[:npub])
    bad = list(wit[:npub]); bad[-1] ^= 1
    assert not pairing.groth16_verify(vk, pp, bad)


# 64x4, 16x16 and 8x32 solver workgroup shapes; ChaCha20 batches of up to 128 take the lane program (k_solve_lanes: a workgroup per instance); AES keeps the level launches
@pytest.mark.parametrize("alg,batch", [(0, 70), (2, 66), (1, 12), (0, 7), (0, 3), (2, 4), (1, 3)])
def test_batch_wires_quotient_and_proofs(zk, oracle, algs, alg, batch):
    algs(alg)
    # solved wires and quotient-stage products are read back through the TEST library (the drop-in exports no debug entries): the same
    # objects with a registry of their own, so the batch is uploaded and stepped there
    tl = zk.testlib
    tl.ZkpAllowUnpinnedKeys(1)
    assert zk.init_algorithm(alg, H.pk_file_bytes(alg), rf.load_r1cs_bytes(H.NAMES[alg]), lib=tl)
    rng = H.SplitMix64(0xBA7C4 + alg)
    insts = [H.make_instance(alg, rng) for _ in range(batch)]
    wits = [H.oracle_witness(alg, i)[0] for i in insts]
    rb = zk.ResidentBatch(alg, np.array(wits, np.uint32), [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts], lib=tl)
    times = rb.step()
    proofs, errs = rb.fetch()
    assert not errs.any(), errs
    r1 = H.circuit(alg).r1cs
    wires, h = rb.debug(r1.nb_wires, r1.domain_size())
    N = r1.domain_size()
    for k in (0, 1, batch // 2, batch - 1):
        exp_proof, _, prf, _ = H.oracle_proof_bytes(alg, insts[k], want_aux=True)
        assert np.array_equal(wires[k], prf["wires"]), "wire vector differs for instance %d" % k
        # quotient stage: the device keeps f = A*B on the coset (the rest of computeH is folded into the bases).  Pin it
        # to the oracle's quotient h:  f = D * H + C on the coset, D = g^N - 1.
        _, oa, ob, oc = H.circuit(alg).solve(wits[k], randomize=insts[k]["randomize"], pk=H.oracle_pk(alg))
        pad = lambda v: np.concatenate([v, np.zeros((N - v.shape[0], 32), np.uint8)])
        on_coset = lambda v: oracle.le32_to_ints(oracle.ntt(oracle.ntt(pad(v), inverse=True), coset=True))
        ac, bc, cc = on_coset(oa), on_coset(ob), on_coset(oc)
        hc = oracle.le32_to_ints(oracle.ntt(prf["h"], coset=True))
        D = (pow(5, N, H.R) - 1) % H.R
        f_exp = [x * y % H.R for x, y in zip(ac, bc)]
        assert all((D * hh + c) % H.R == f for hh, c, f in zip(hc, cc, f_exp)), "oracle quotient inconsistent"
        assert oracle.le32_to_ints(h[k]) == f_exp, "quotient-stage products differ for instance %d" % k
        assert zk.proof_serialize(proofs[k], alg != 0) == exp_proof
    # every proof of the batch verifies (size-independent property), on a sample to bound test time
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    npub = r1.nb_public - 1
    for k in (2, batch - 2):
        pp = gf.proof_parse(zk.proof_serialize(proofs[k], alg != 0))
        assert pairing.groth16_verify(vk, pp, wits[k][:npub])
    assert times["total"] > 0
    # the drop-in's resident entry gives the same records
    rb2 = zk.ResidentBatch(alg, np.array(wits, np.uint32), [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
    rb2.step()
    proofs2, errs2 = rb2.fetch()
    assert not errs2.any() and np.array_equal(proofs2, proofs)


def test_prove_random_blinding_verifies_and_differs(zk, oracle, algs):
    alg = algs(0)
    inst = H.make_instance(alg, H.SplitMix64(77))
    rq = json.loads(H.request_json(alg, inst))
    a = zk.prove(rq)
    b = zk.prove(rq)
    assert a["ciphertext"] == b["ciphertext"]
    assert a["proof"] != b["proof"]            # r, s from the OS entropy source, as gnark
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    wit, _ = H.oracle_witness(alg, inst)
    for res in (a, b):
        assert pairing.groth16_verify(vk, gf.proof_parse(base64.b64decode(res["proof"])), wit[:2304])


def test_prove_batch_entry(zk, oracle, algs):
    alg = algs(1)
    rng = H.SplitMix64(4242)
    insts = [H.make_instance(alg, rng) for _ in range(3)]
    reqs = [json.loads(H.request_json(alg, i)) for i in insts]
    reqs.insert(1, {"cipher": "aes-128-ctr", "key": "AAAA", "blocks": [], "input": ""})     # bad request inside the batch
    res = json.loads(zk.prove_batch_raw(json.dumps(reqs).encode()))
    assert len(res) == 4
    assert res[1] == "key length must be 16 or 32: 3"
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    for k, inst in zip((0, 2, 3), insts):
        wit, out = H.oracle_witness(alg, inst)
        assert base64.b64decode(res[k]["ciphertext"]) == out
        assert pairing.groth16_verify(vk, gf.proof_parse(base64.b64decode(res[k]["proof"])), wit[:225])


def test_aux_test_go_literal_request(zk, algs):
    """gnark/libraries/aux_test.go:150-185: JS-shaped request; its ignored "ciphertext" field pins Prove(...).ciphertext"""
    algs(1)
    j = json.load(open(__import__("os").path.join(H.ROOT, "tests", "golden", "aux_test_request.json")))
    res = zk.prove(j)
    assert res["ciphertext"] == j["ciphertext"]
    assert len(base64.b64decode(res["proof"])) == 196


def test_compressed_proving_key_loads_and_proves_identically(zk, oracle, algs):
    """ProvingKey.WriteTo (compressed, what keygen.go:99-118 writes) and WriteRawTo load to the same key"""
    algs(0)
    inst = H.make_instance(0, H.SplitMix64(99))
    a = zk.prove_deterministic_raw(H.request_json(0, inst), inst["r"], inst["s"], 0)
    zk.lib.ZkpReset()
    assert zk.init_algorithm(0, H.pk_file_bytes(0, raw=False), rf.load_r1cs_bytes("chacha20"))
    b = zk.prove_deterministic_raw(H.request_json(0, inst), inst["r"], inst["s"], 0)
    assert a == b and b"proof" in a
    zk.lib.ZkpReset()


@pytest.mark.parametrize("alg", [0, 1])
def test_gpu_setup_matches_oracle_setup(zk, oracle, alg):
    """ZkpSetup (product key generator, gnark/keygen/keygen.go:94-132) == oracle setup for the same toxic waste,
    byte for byte, in both gnark encodings; the verifying key has the reference's vk file size."""
    r1cs = rf.load_r1cs_bytes(H.NAMES[alg])
    tox = oracle.toxic_from_seed(1)
    assert tox == zk.synth.toxic_from_seed(1)
    pk_raw, vk = zk.setup(r1cs, tox, raw=True)
    opk = H.oracle_pk(alg, 1)
    assert pk_raw == gf.pk_bytes(opk, raw=True)
    assert vk == gf.vk_bytes(gf.vk_from_pk(opk))
    assert len(vk) == (74060 if alg == 0 else 7696)
    if alg == 0:
        pk_c, vk_c = zk.setup(r1cs, tox, raw=False)
        assert vk_c == vk and pk_c == H.pk_file_bytes(0, raw=False)


def test_pipelined_steps_produce_the_same_proofs(zk, oracle, algs):
    """zkp_bench_steps (solve of pass k+1 overlapped with NTT+MSM of pass k on a second stream) == single-stream step"""
    for alg, batch in ((0, 64), (1, 64)):
        algs(alg)
        rng = H.SplitMix64(0x1234 + alg)
        insts = [H.make_instance(alg, rng) for _ in range(batch)]
        wits = np.array([H.oracle_witness(alg, i)[0] for i in insts], np.uint32)
        rb = zk.ResidentBatch(alg, wits, [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
        rb.step()
        ref_proofs, errs = rb.fetch()
        assert not errs.any()
        for n in (1, 2, 5):
            rb.steps(n)
            got, errs = rb.fetch()
            assert not errs.any() and np.array_equal(got, ref_proofs), "pipelined pass count %d" % n
        exp, *_ = H.oracle_proof_bytes(alg, insts[3])
        assert zk.proof_serialize(ref_proofs[3], alg != 0) == exp


@pytest.mark.parametrize("alg_id,flip", [(0, 1500), (2, 40)])
def test_unsatisfiable_witness_is_flagged_per_instance(zk, oracle, algs, alg_id, flip):
    """a witness whose Out does not match the cipher output violates a constraint: that instance (only) reports a
    solver error -- the reference's groth16.Prove returns an error there (panic -> {} at the ABI).  AES-256: the same through the
    levels that k_solve_chains walks (the instance's neighbours in the wavefront keep their wires and proofs)"""
    alg = algs(alg_id)
    rng = H.SplitMix64(31)
    insts = [H.make_instance(alg, rng) for _ in range(4)]
    wits = np.array([H.oracle_witness(alg, i)[0] for i in insts], np.uint32)
    wits[2, flip] ^= 1                      # flip one bit of Out
    rb = zk.ResidentBatch(alg, wits, [i["r"] for i in insts], [i["s"] for i in insts], [i["randomize"] for i in insts])
    rb.step()
    proofs, errs = rb.fetch()
    assert errs[2] != 0 and not errs[[0, 1, 3]].any()
    if alg_id == 0:
        assert (errs[2] & 0xFF) == 1
    for k in (0, 3):
        exp, *_ = H.oracle_proof_bytes(alg, insts[k])
        assert zk.proof_serialize(proofs[k], alg_id != 0) == exp
    with pytest.raises(ValueError):
        H.circuit(alg).solve([int(x) for x in wits[2]])


def test_chain_runs_cover_the_tail_of_the_aes_programs(zk):
    """csrc/solver.hip, k_solve_chains: the 321 narrow levels that end the AES programs (five permutations of dependent multiplications
    and their check rows) are one chain run -- components that read none of each other's wires, a wavefront each; ChaCha20 has no such
    run.  (Its wires are pinned by every parity test of the AES circuits; ZKPROVE_SOLVER_CHAINS=0 keeps the level-by-level walk of the
    same run bit-exact, test_gpu_kernels.py::test_alternative_field_paths_stay_bit_exact[narrow_runs_and_count_hints_inside_the_level_kernels].)"""
    import ctypes
    plans = {}
    for alg_id in (0, 1, 2):
        out = (ctypes.c_long * 4)()
        tl = zk.testlib
        r1cs = zk.synth.r1cs_bytes(alg_id)
        pk, _ = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
        tl.ZkpAllowUnpinnedKeys(1)
        assert zk.init_algorithm(alg_id, pk, r1cs, lib=tl)
        assert tl.zkp_debug_solver_plan(alg_id, out) == 0
        plans[alg_id] = list(out)
    tl.ZkpReset()
    assert plans[0][1] == 0, plans
    for a in (1, 2):
        levels, runs, inside, waves = plans[a]
        assert runs >= 1 and inside >= 300 and inside < levels and 5 <= waves <= 8, plans


@pytest.mark.parametrize("alg", [0, 1, 2])
def test_reference_full_test_prove_then_verify_through_both_abis(zk, alg):
    """gnark/libraries/test_helpers.go:292-411 RunFullTest with the product on both sides: keys from ZkpSetup,
    prover.Prove(json) then verifier.Verify(json) must be true; js/src/tests/lib.test.ts:124-158: tampered -> false"""
    zk.lib.ZkpAllowUnpinnedKeys(1)
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, zk.synth.toxic_from_seed(7), raw=True)
    zk.lib.ZkpReset()
    assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)
    rng = H.SplitMix64(2024 + alg)
    for _ in range(2):
        inst = H.make_instance(alg, rng)
        res = zk.prove(json.loads(H.request_json(alg, inst)))
        proof, ct = base64.b64decode(res["proof"]), base64.b64decode(res["ciphertext"])
        blocks = list(zip(inst["nonces"], inst["counters"]))
        assert zk.verify(H.CIPHER[alg], proof, ct, blocks, inst["data"])
        bad = bytearray(ct); bad[0] ^= 1
        assert not zk.verify(H.CIPHER[alg], proof, bytes(bad), blocks, inst["data"])
    zk.lib.ZkpReset()


def test_concurrent_init_and_prove(zk, oracle):
    """gnark/libraries/aux_test.go:40-83 TestInit: InitAlgorithm from many threads at once is safe and idempotent;
    Prove is re-entrant"""
    from concurrent.futures import ThreadPoolExecutor
    zk.lib.ZkpReset()
    zk.lib.ZkpAllowUnpinnedKeys(1)
    pk, r1cs = H.pk_file_bytes(0), rf.load_r1cs_bytes("chacha20")
    with ThreadPoolExecutor(max_workers=12) as ex:
        assert all(ex.map(lambda _: zk.init_algorithm(0, pk, r1cs), range(24)))
    rng = H.SplitMix64(5150)
    insts = [H.make_instance(0, rng) for _ in range(6)]
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(lambda i: zk.prove_deterministic_raw(H.request_json(0, i), i["r"], i["s"], 0), insts))
    for inst, raw in zip(insts, res):
        exp, out, *_ = H.oracle_proof_bytes(0, inst)
        got = json.loads(raw)
        assert base64.b64decode(got["proof"]) == exp and base64.b64decode(got["ciphertext"]) == out


@pytest.mark.parametrize("alg", [0, 2])
def test_concurrent_prove_calls_are_coalesced(zk, oracle, algs, alg):
    """Prove() from many threads at once (a Go server's goroutines, Node's libuv pool): calls that arrive while the device is busy are
    proved together in the next batch.  Every caller gets the proof of ITS request (checked against its public inputs), a bad request
    among them gets its own panic, and the wall time shows the calls did not queue one by one"""
    import threading
    import time
    algs(alg)
    vkb = gf.vk_bytes(gf.vk_from_pk(H.oracle_pk(alg)))
    assert zk.init_verifier(alg, vkb)
    rng = H.SplitMix64(31337 + alg)
    nthreads, per = 24, 4
    insts = [[H.make_instance(alg, rng) for _ in range(per)] for _ in range(nthreads)]
    reqs = [[H.request_json(alg, i) for i in row] for row in insts]
    zk.prove_raw(reqs[0][0])                                           # warm-up (allocations)
    t0 = time.perf_counter(); zk.prove_raw(reqs[0][0]); one = time.perf_counter() - t0
    out = [[None] * per for _ in range(nthreads)]
    bad = [None]

    def worker(t):
        for k in range(per):
            out[t][k] = zk.prove_raw(reqs[t][k])
        if t == 3:
            bad[0] = zk.prove_raw(json.dumps({"cipher": H.CIPHER[alg], "key": "AAAA", "blocks": [], "input": ""}).encode())

    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    wall = time.perf_counter() - t0
    assert json.loads(bad[0]).startswith("key length must be")
    for t in range(nthreads):
        for k in range(per):
            res = json.loads(out[t][k])
            rq = json.loads(reqs[t][k])
            _, exp_out = H.oracle_witness(alg, insts[t][k])
            assert base64.b64decode(res["ciphertext"]) == exp_out
            if (t * per + k) % 5 == 0:
                sig = {"ciphertext": res["ciphertext"], "blocks": rq["blocks"], "input": rq["input"]}
                assert zk.verify_raw(json.dumps({"cipher": H.CIPHER[alg], "proof": res["proof"], "publicSignals": sig}).encode())
    # 96 calls one by one would take 96 * `one`; coalesced batches must do clearly better
    assert wall < 0.5 * nthreads * per * one, "concurrent calls were not batched: %.3f s vs %.3f s serial" % (wall, nthreads * per * one)
    print("coalescing: %d concurrent Prove() calls in %.3f s (%.1f proofs/s; one call alone %.1f ms)" % (nthreads * per, wall, nthreads * per / wall, one * 1e3))


@pytest.mark.parametrize("alg,n", [(0, 3100), (1, 1030)])
def test_prove_batch_larger_than_one_gpu_chunk(zk, oracle, algs, alg, n):
    """ProveBatch with more requests than the 1024-instance device chunk: the chunks run software-pipelined on the device (working
    sets are reused from the third chunk on; the AES circuits add the commitment round trip); order preserved, a bad request in the
    middle keeps its slot, every checked proof verifies against ITS request"""
    algs(alg)
    rng = H.SplitMix64(8080 + alg)
    insts = [H.make_instance(alg, rng) for _ in range(n)]
    parts = [H.request_json(alg, i).decode() for i in insts]
    parts.insert(1500 if n > 1500 else 700, json.dumps({"cipher": H.CIPHER[alg], "key": "AAAA", "blocks": [], "input": ""}))
    bad_at = 1500 if n > 1500 else 700
    res = json.loads(zk.prove_batch_raw(("[" + ",".join(parts) + "]").encode()))
    assert len(res) == n + 1
    assert isinstance(res[bad_at], str) and res[bad_at].startswith("key length must be")
    del res[bad_at]
    assert all(isinstance(r, dict) and "proof" in r for r in res)
    npub = H.circuit(alg).r1cs.nb_public - 1
    vk_bytes = gf.vk_bytes(gf.vk_from_pk(H.oracle_pk(alg)))
    assert zk.init_verifier(alg, vk_bytes)
    picks = sorted(set([0, 1, 1022, 1023, 1024, 1025, n - 1] + list(range(7, n, 211))))
    for k in picks:
        wit, out = H.oracle_witness(alg, insts[k])
        assert base64.b64decode(res[k]["ciphertext"]) == out
        rq = json.loads(parts[k if k < bad_at else k + 1])
        sig = {"ciphertext": res[k]["ciphertext"], "blocks": rq["blocks"], "input": rq["input"]}
        assert zk.verify_raw(json.dumps({"cipher": H.CIPHER[alg], "proof": res[k]["proof"], "publicSignals": sig}).encode()), "proof %d does not verify" % k
        if k + 1 < n:       # and it is not accidentally the neighbour's proof
            assert not zk.verify_raw(json.dumps({"cipher": H.CIPHER[alg], "proof": res[k + 1]["proof"], "publicSignals": sig}).encode())
    vk = gf.vk_from_pk(H.oracle_pk(alg))
    wit, _ = H.oracle_witness(alg, insts[n - 1])
    assert pairing.groth16_verify(vk, gf.proof_parse(base64.b64decode(res[n - 1]["proof"])), wit[:npub])


@pytest.mark.parametrize("alg", [0, 2])
def test_extreme_inputs(zk, oracle, algs, alg):
    """maximum counters, all-zero and all-ones keys / plaintexts, per-block distinct nonces (aux_test.go:192-322)"""
    algs(alg)
    nblk = 2 if alg == 0 else 5
    n = 128 if alg == 0 else 80
    klen = 32
    cases = [
        dict(key=bytes(klen), nonces=[bytes(12)] * nblk, counters=[0] * nblk, data=bytes(n)),
        dict(key=b"\xff" * klen, nonces=[b"\xff" * 12] * nblk, counters=[0xFFFFFFFF] * nblk, data=b"\xff" * n),
        dict(key=bytes(range(klen)), nonces=[bytes([b] * 12) for b in range(nblk)], counters=[0xFFFFFFFF - b for b in range(nblk)], data=bytes(range(n))),
    ]
    for k, c in enumerate(cases):
        inst = dict(c, r=12345 + k, s=R_TEST - 1 - k, randomize=7 * k)
        res = json.loads(zk.prove_deterministic_raw(H.request_json(alg, inst), inst["r"], inst["s"], inst["randomize"]))
        exp, out, *_ = H.oracle_proof_bytes(alg, inst)
        assert base64.b64decode(res["ciphertext"]) == out and base64.b64decode(res["proof"]) == exp
    # blinding scalars at the edges of the tail's scalar multiplications (the doubling chain + adder wavefronts of small batches, the
    # 16-lane fixed-base trees): zero (nothing is added anywhere), one, the endomorphism's eigenvalue (k1 = 0, k2 = 1), 2^127 and 2^128 - 1
    # (the top of a half-scalar), r - 1
    LAMBDA = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    for k, (r, s) in enumerate([(0, 0), (1, 0), (0, 1), (LAMBDA, LAMBDA + 1), (1 << 127, (1 << 128) - 1), (R_TEST - 1, R_TEST - 1)]):
        inst = dict(cases[k % 3], r=r, s=s, randomize=k)
        res = json.loads(zk.prove_deterministic_raw(H.request_json(alg, inst), inst["r"], inst["s"], inst["randomize"]))
        exp, out, *_ = H.oracle_proof_bytes(alg, inst)
        assert base64.b64decode(res["proof"]) == exp, "blinding scalars r = %x, s = %x" % (r, s)


R_TEST = rf.R_MOD


def test_full_size_batch_every_proof_verifies(zk, oracle):
    """BASELINE config[1] at full size (ChaCha20, batch 1024): every proof of the batch passes the product's verifier
    (itself cross-checked against the oracle's independent pairing in tests/test_verifier.py), a tampered one fails, and
    sampled proofs are bit-identical to the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    zk.lib.ZkpAllowUnpinnedKeys(1)
    alg, B = 0, 1024
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
    zk.lib.ZkpReset()
    assert zk.init_algorithm(alg, pk, r1cs) and zk.init_verifier(alg, vk)
    rng = H.SplitMix64(0x5EED0000 + 1)
    insts = [H.make_instance(alg, rng) for _ in range(B)]
    wits, outs = [], []
    for inst in insts:
        _, w, out = zk.request_witness(H.request_json(alg, inst))
        wits.append(w); outs.append(out)
    rb = zk.ResidentBatch(alg, np.array(wits, np.uint32), [i["r"] for i in insts], [i["s"] for i in insts], [0] * B)
    rb.step()                                   # one unpipelined pass: a single working set, no batch prepared ahead
    single, errs = rb.fetch()
    assert not errs.any()
    # seven pipelined passes at the size the bench runs: all three working sets and both copies of the quotient MSM's ahead-prepared
    # data (comb patterns / sorted list) are reused, so the consumed / zdone waits of prover.hip are on the path; every pass proves the
    # same resident inputs, hence all 1024 records of the last one must equal the single pass
    rb.steps(7)
    proofs, errs = rb.fetch()
    assert not errs.any()
    assert np.array_equal(proofs, single), "pipelined passes differ from the single pass in records %s" % np.nonzero((proofs != single).any(axis=1))[0][:10]
    info = np.zeros(8, np.int64)
    assert zk.lib.zkp_quotient_msm_info(alg, info.ctypes.data) == 0 and info[0] == 1, "a 1024-batch must take the comb path of the quotient MSM"
    ser = [zk.proof_serialize(proofs[k], False) for k in range(B)]
    assert len(set(ser)) == B

    def check(k):
        return zk.verify(H.CIPHER[alg], ser[k], outs[k], list(zip(insts[k]["nonces"], insts[k]["counters"])), insts[k]["data"])
    with ThreadPoolExecutor(max_workers=16) as ex:
        ok = list(ex.map(check, range(B)))
    assert all(ok), "instances failing verification: %s" % [k for k, v in enumerate(ok) if not v][:10]
    assert not zk.verify(H.CIPHER[alg], ser[0], outs[1], list(zip(insts[0]["nonces"], insts[0]["counters"])), insts[0]["data"])
    for k in (0, 511, 1023):
        exp, out, *_ = H.oracle_proof_bytes(alg, dict(insts[k], randomize=0))
        assert ser[k] == exp and outs[k] == out
    zk.lib.ZkpReset()


def _pk_offsets(pk: bytes):
    """byte offsets of the first point of each slice of an uncompressed proving key (SURVEY App. C layout)"""
    u32 = lambda o: int.from_bytes(pk[o:o + 4], "big")
    pos = 8 + 5 * 32 + 1
    off = {"alpha": pos}
    pos += 3 * 64
    for name in ("A", "B", "Z", "K"):
        n = u32(pos)
        off[name] = pos + 4
        pos += 4 + 64 * n
    off["beta2"] = pos
    pos += 2 * 128
    off["B2"] = pos + 4
    return off


def test_init_rejects_invalid_key_points(zk, oracle):
    """ProvingKey.ReadFrom semantics (impl/library.go:104-108: error -> InitAlgorithm false): a point off the curve, a non-reduced
    coordinate, and a G2 point on the twist but outside the order-r subgroup are all refused; the intact key still loads"""
    zk.lib.ZkpAllowUnpinnedKeys(1)
    zk.lib.ZkpReset()
    pk = bytearray(H.pk_file_bytes(0, raw=True))
    r1cs = rf.load_r1cs_bytes("chacha20")
    off = _pk_offsets(pk)
    P = gf.P

    def refused(mutated):
        ok = zk.init_algorithm(0, bytes(mutated), r1cs)
        zk.lib.ZkpReset()
        return not ok

    bad = bytearray(pk); bad[off["A"] + 63] ^= 1                                   # y of A[0] off the curve
    assert refused(bad)
    bad = bytearray(pk); bad[off["K"] + 5 * 64 + 40] ^= 0x10                       # K[5]
    assert refused(bad)
    (x, y), _ = gf.g1_decode(bytes(pk), off["Z"])
    bad = bytearray(pk); bad[off["Z"]:off["Z"] + 32] = (x + P).to_bytes(32, "big")   # same point, x not reduced (x + p < 2^256)
    assert refused(bad)
    # a twist point outside the subgroup: the twist has cofactor 2p - r, so a random point on it is (almost surely) not in G2
    xx = 1
    while True:
        comp = bytearray((0).to_bytes(32, "big") + xx.to_bytes(32, "big")); comp[0] |= 0x80
        try:
            pt, _ = gf.g2_decode(bytes(comp))
            break
        except ValueError:
            xx += 1
    assert gf.g2_on_curve(pt)
    bad = bytearray(pk); bad[off["B2"]:off["B2"] + 128] = gf.g2_raw(pt)
    assert refused(bad)
    bad = bytearray(pk); bad[off["beta2"] + 127] ^= 1                               # beta2 off the twist
    assert refused(bad)
    # the compressed form goes through the same checks
    pkc = bytearray(H.pk_file_bytes(0, raw=False))
    u32 = lambda o: int.from_bytes(pkc[o:o + 4], "big")
    pos = 8 + 5 * 32 + 1 + 3 * 32
    for _ in range(4):
        pos += 4 + 32 * u32(pos)
    pos += 2 * 64 + 4                                                               # first compressed B2 point
    bad = bytearray(pkc); bad[pos:pos + 64] = gf.g2_compress(pt)
    assert refused(bad)
    assert zk.init_algorithm(0, bytes(pk), r1cs)                                     # untouched key loads
    zk.lib.ZkpReset()


def test_init_answers_truncated_and_length_inflated_keys(zk, oracle):
    """the proving-key reader against structural damage (the reader of untrusted-size fields; ProvingKey.ReadFrom errors ->
    InitAlgorithm false, impl/library.go:104-108): every truncation, and section counts that promise more points than the file holds,
    fewer, none, or 2^32 - 1, are refused with a message -- no crash, no allocation by the inflated count -- and the library is usable
    afterwards"""
    import struct
    zk.lib.ZkpAllowUnpinnedKeys(1)
    zk.lib.ZkpReset()
    r1cs = rf.load_r1cs_bytes("chacha20")
    for raw in (True, False):
        pk = H.pk_file_bytes(0, raw=raw)
        off = 8 + 5 * 32 + 1 + 3 * (64 if raw else 32)                            # the u32 count of G1.A
        n_a = int.from_bytes(pk[off:off + 4], "big")
        cuts = list(range(0, 24)) + [off - 1, off, off + 3, off + 4, off + 5, len(pk) // 3, len(pk) // 2, len(pk) - 200, len(pk) - 9, len(pk) - 1]
        cases = [pk[:c] for c in cuts] + [pk + b"\x00", pk + pk[:64]]
        for v in (0, 1, n_a - 1, n_a + 1, 0x7FFFFFFF, 0xFFFFFFFF):
            cases.append(pk[:off] + struct.pack(">I", v) + pk[off + 4:])
        cases.append(struct.pack(">Q", 1 << 40) + pk[8:])                          # domain cardinality
        cases.append(pk[:-4] + b"\xff\xff\xff\xff")
        for bad in cases:
            assert not zk.init_algorithm(0, bad, r1cs)
            assert zk.lib.ZkpDeviceReplicas(0) == 0
    assert not zk.init_algorithm(0, H.pk_file_bytes(0), r1cs[:-1])                  # the R1CS pin always holds
    assert zk.init_algorithm(0, H.pk_file_bytes(0), r1cs)
    zk.lib.ZkpReset()
