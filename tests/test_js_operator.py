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


def _run_node(args, timeout=600):
    p = subprocess.run([NODE] + args, cwd=JS, capture_output=True, text=True, timeout=timeout)
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
    assert r["keys"] == sorted(["open", "initAlgorithm", "prove", "proveBatch", "proveAsync", "proveBatchAsync", "verify", "initVerifier",
                                "allowUnpinnedKeys", "setup"])
    assert r["caps"] == {"extensions": True, "verifier": True}
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
    insts = [H.make_instance(alg, rng) for _ in range(5)]
    _instances_file(d, alg, insts)
    r = _run_node([os.path.join("test", "run_operator.js"), "prove", d])
    assert "error" not in r, r.get("error")
    assert r["okSingle"] is True and r["okBatch"] == [True] * 5 and r["okTampered"] is False
    assert r["badErr"] == 'Failed to create gnark proof: "key length must be 32: 3"'       # operator.ts:26-30 over provers.go:120-133
    # the proofs Node got verify under the independent pairing of the oracle as well
    vkp = gf.vk_parse(vk)
    for k, p64 in enumerate([r["single"]] + r["batch"]):
        inst = insts[0 if k == 0 else k - 1]
        wit, _ = H.oracle_witness(alg, inst)
        proof = base64.b64decode(p64)
        assert len(proof) == 164
        assert pairing.groth16_verify(vkp, gf.proof_parse(proof), wit[:H.circuit(alg).r1cs.nb_public - 1])
