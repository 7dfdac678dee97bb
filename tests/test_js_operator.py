"""Node-side operator (js/gnark-operator.js + the N-API binding js/native/zkprove_napi.c): SURVEY section 8(f)-4.

Mirrors how the reference's JS tests drive `makeGnarkZkOperator` (js/src/tests/*.test.ts): generateWitness -> groth16Prove ->
groth16Verify, plus the batch entry.  The CPU tests cover the binding and the request JSON; the GPU test proves through Node."""
import base64
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import helpers as H

ROOT = H.ROOT
JS = os.path.join(ROOT, "js")
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None or not os.path.exists("/usr/include/node/node_api.h"), reason="node / node headers not installed")


def _addon():
    subprocess.run(["make", "-s", "-C", JS], check=True)
    return os.path.join(JS, "zkprove_napi.node")


def _run_node(args, timeout=600, env=None):
    p = subprocess.run([NODE] + args, cwd=JS, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert lines, "no result from node:\n%s\n%s" % (p.stdout[-2000:], p.stderr[-2000:])
    return json.loads(lines[-1][len("RESULT "):])


def _instances_file(tmp_path, alg, insts):
    b64 = lambda b: base64.b64encode(bytes(b)).decode()
    spec = {"algorithm": H.CIPHER[alg], "instances": []}
    for inst in insts:
        _, out = H.oracle_witness(alg, inst)
        spec["instances"].append({
            "key": b64(inst["key"]), "in": b64(inst["data"]), "out": b64(out),
            "noncesAndCounters": [{"nonce": b64(n), "counter": int(c)} for n, c in zip(inst["nonces"], inst["counters"])]})
    with open(os.path.join(tmp_path, "instances.json"), "w") as f:
        json.dump(spec, f)
    return spec


def test_addon_builds_loads_and_reports_library_errors():
    _addon()
    script = ("const a=require('./zkprove_napi.node');"
              "const caps=a.open(process.argv[1]);a.allowUnpinnedKeys(true);"
              "const r={keys:Object.keys(a).sort(),caps,"
              "init:a.initAlgorithm(0,Buffer.from('x'),Buffer.from('y')),"
              "prove:a.prove(Buffer.from(JSON.stringify({cipher:'chacha20'}))).toString(),"
              "unknown:a.prove('{\"cipher\":\"nope\"}').toString(),verify:a.verify('{}')};"
              "process.stdout.write('\\nRESULT '+JSON.stringify(r)+'\\n')")
    r = _run_node(["-e", script, os.path.join(ROOT, "zk-symmetric-crypto_amd", "libzkprove.so")])
    assert r["keys"] == sorted(["open", "initAlgorithm", "prove", "proveBatch", "proveAsync", "proveBatchAsync", "verifyBatchAsync", "verify", "initVerifier",
                                "allowUnpinnedKeys", "setup"])
    assert r["caps"] == {"extensions": True, "verifier": True, "batchVerifier": True}
    assert r["init"] is False                                   # garbage r1cs: hash pin mismatch, as the reference (library.go:111-118)
    assert json.loads(r["prove"]) == "proving params are not initialized for cipher: chacha20"
    assert json.loads(r["unknown"]) == "could not find prover fornope"
    assert r["verify"] is False


@pytest.mark.parametrize("alg", [0, 2])
def test_operator_request_json_decodes_to_the_oracle_witness(zk, tmp_path, alg):
    """generateWitness (JS) -> request JSON -> the library's request decoder -> the same witness assignment as the oracle's"""
    _addon()
    rng = H.SplitMix64(900 + alg)
    insts = [H.make_instance(alg, rng) for _ in range(3)]
    _instances_file(str(tmp_path), alg, insts)
    r = _run_node([os.path.join("test", "run_operator.js"), "witness", str(tmp_path)])
    assert len(r["requests"]) == 3
    for rq, inst in zip(r["requests"], insts):
        assert rq["cipher"] == H.CIPHER[alg] and rq["toprf"] == {} and all(b["boundary"] is None for b in rq["blocks"])
        wit, out = H.oracle_witness(alg, inst)
        _, got_wit, got_out = zk.request_witness(json.dumps(rq).encode())
        assert list(got_wit) == list(wit)
        assert bytes(got_out) == bytes(out)


@pytest.mark.gpu
def test_operator_proves_and_verifies_through_node(zk, oracle, tmp_path):
    from oracle import gnark_format as gf
    from oracle import pairing
    _addon()
    alg = 0
    d = str(tmp_path)
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
    for name, data in (("pk.chacha20", pk), ("r1cs.chacha20", r1cs), ("vk.chacha20", vk)):
        with open(os.path.join(d, name), "wb") as f:
            f.write(data)
    rng = H.SplitMix64(4711)
    insts = [H.make_instance(alg, rng) for _ in range(16)]
    _instances_file(d, alg, insts)
    r = _run_node([os.path.join("test", "run_operator.js"), "prove", d], env={"UV_THREADPOOL_SIZE": "16"})
    assert "error" not in r, r.get("error")
    assert r["okSingle"] is True and r["okBatch"] == [True] * 16 and r["okTampered"] is False
    # groth16VerifyBatch (VerifyBatch through the addon): 16 proofs = one GPU batch check; with member 1 under member 0's public signals
    # the batch check fails and the per-proof fallback answers exactly as groth16Verify would
    assert r["okVerifyBatch"] == [True] * 16 and r["okVerifyBatchTampered"] == [k != 1 for k in range(16)]
    assert r["okConcurrent"] == [True] * 16          # Promise.all over groth16Prove: coalesced into device batches by the library
    print("node: single %d ms, batch of 16 %d ms, Promise.all of 16 single calls %d ms (libuv pool %s)" % (r["msSingle"], r["msBatch"], r["msConcurrent"], r["poolSize"]))
    assert r["badErr"] == 'Failed to create gnark proof: "key length must be 32: 3"'       # operator.ts:26-30 over provers.go:120-133
    # the proofs Node got verify under the independent pairing of the oracle as well
    vkp = gf.vk_parse(vk)
    for k, p64 in enumerate([r["single"]] + r["batch"]):
        inst = insts[0 if k == 0 else k - 1]
        wit, _ = H.oracle_witness(alg, inst)
        proof = base64.b64decode(p64)
        assert len(proof) == 164
        assert pairing.groth16_verify(vkp, gf.proof_parse(proof), wit[:H.circuit(alg).r1cs.nb_public - 1])


def test_zk_layer_chunking_and_counters(tmp_path):
    """js/zk.js against the reference's rules (zk.ts:96-178, utils.ts:206-245): start counters, padding blocks, chunk split"""
    script = ("const zk=require('./zk.js');"
              "const s=zk.getPublicSignals({algorithm:'aes-128-ctr',key:Buffer.alloc(16,1),publicInput:{ciphertext:new Uint8Array(16).fill(7),iv:new Uint8Array(12),offsetBytes:240}});"
              "const c=zk.getPublicSignals({algorithm:'chacha20',key:Buffer.from([...Array(32).keys()]),publicInput:{ciphertext:Buffer.from('Ladies and Gentlemen'),iv:Buffer.from('000000000000004a00000000','hex')}});"
              "let err='';try{zk.getCounterForByteOffset('chacha20',5)}catch(e){err=e.message}"
              "const r={aesCounters:s.noncesAndCounters.map(n=>n.counter),aesIn:Buffer.from(s.in).toString('hex'),aesLen:[s.in.length,s.out.length],"
              "ccCounters:c.noncesAndCounters.map(n=>n.counter),ccOut:Buffer.from(c.out).toString('hex').slice(0,40),ccLen:c.in.length,"
              "chunks:zk.splitToChunks('aes-128-ctr',{ciphertext:new Uint8Array(256),iv:new Uint8Array(12)}).map(c=>[c.offsetBytes,c.ciphertext.length]),err};"
              "process.stdout.write('\\nRESULT '+JSON.stringify(r)+'\\n')")
    r = _run_node(["-e", script])
    assert r["aesCounters"] == [17, 18, 19, 20, 21]                       # AES-GCM payload starts at counter 2 (config.ts:35,50); offset 240 = block 15
    assert r["aesIn"] == "07" * 16 + "00" * 64 and r["aesLen"] == [80, 80]
    assert r["ccCounters"] == [1, 2] and r["ccLen"] == 128
    assert r["ccOut"] == "6e2e359a2568f98041ba0728dd0d6981e97e7aec"          # RFC 8439 2.4.2 keystream, as chacha_test.go:123-152
    assert r["chunks"] == [[0, 80], [80, 80], [160, 80], [240, 16]]
    assert r["err"] == "offset(5) must be a multiple of block size(64)"


@pytest.mark.gpu
def test_config5_aes128_16_block_items_through_node(zk, oracle, tmp_path):
    """BASELINE config 5 at full size: 256 records of 16 AES-128 blocks -> 4 chunk proofs each = 1024 proofs in one ProveBatch call
    from Node; every 8th proof (and the last) is verified, a tampered plaintext is rejected, the plaintexts match the oracle's cipher"""
    _addon()
    alg = 1
    d = str(tmp_path)
    r1cs = zk.synth.r1cs_bytes(alg)
    pk, vk = zk.setup(r1cs, zk.synth.toxic_from_seed(1), raw=True)
    for name, data in (("pk.aes128", pk), ("r1cs.aes128", r1cs), ("vk.aes128", vk)):
        with open(os.path.join(d, name), "wb") as f:
            f.write(data)
    rng = H.SplitMix64(0x5EED0000 + 4)
    b64 = lambda b: base64.b64encode(bytes(b)).decode()
    items = []
    for _ in range(256):
        items.append({"key": rng.bytes(16), "iv": rng.bytes(12), "ciphertext": rng.bytes(256)})
    with open(os.path.join(d, "items.json"), "w") as f:
        json.dump({"algorithm": "aes-128-ctr", "verifyStride": 8, "items": [{k: b64(v) for k, v in it.items()} for it in items]}, f)
    r = _run_node([os.path.join("test", "run_items.js"), d], timeout=900)
    assert "error" not in r, r.get("error")
    assert r["items"] == 256 and r["proofs"] == 1024 and r["verified"] == 129 and r["rejected"] is True
    assert r["proofBytes"] == 196
    assert r["lastChunk"]["offsetBytes"] == 240 and r["lastChunk"]["bytes"] == 16
    # plaintext of the first chunk = the oracle's AES-CTR keystream XOR (counters 2..6), of the last = block 17 then the keystream of 4 empty blocks
    from oracle import witness as W
    it = items[0]
    _, out = W.aes_witness(it["key"], [it["iv"]] * 5, [2, 3, 4, 5, 6], it["ciphertext"][:80])
    assert base64.b64decode(r["firstPlaintext"]) == bytes(out)
    it = items[-1]
    _, out = W.aes_witness(it["key"], [it["iv"]] * 5, [17, 18, 19, 20, 21], it["ciphertext"][240:] + bytes(64))
    assert base64.b64decode(r["lastChunk"]["plaintext"]) == bytes(out)
    print("config 5 through Node: %.1f items/s, %.1f proofs/s (prove %d ms, verify %d ms)" % (r["itemsPerS"], r["proofsPerS"], r["proveMs"], r["verifyMs"]))
