'use strict'
/**
 * BASELINE config 5 through the JS layer: records longer than one chunk (AES-128-CTR, 16 blocks = 256 B) are split into chunk
 * proofs as js/src/zk.ts would (last chunk = 1 real block + 4 empty blocks at the following counters), all chunks proved in
 * ONE library call, every proof verified.   node run_items.js <dir>     (<dir>: pk/r1cs/vk + items.json, see test_js_operator.py)
 */
const fs = require('fs')
const path = require('path')
const { makeGnarkZkOperator } = require('../gnark-operator.js')
const zk = require('../zk.js')

const dir = process.argv[2]
const spec = JSON.parse(fs.readFileSync(path.join(dir, 'items.json')))
const u8 = (s) => new Uint8Array(Buffer.from(s, 'base64'))
const fetcher = {
	async fetch(engine, name) { return fs.readFileSync(path.join(dir, name)) },
	async has(engine, name) { return fs.existsSync(path.join(dir, name)) },
}

async function main() {
	const algorithm = spec.algorithm
	const operator = makeGnarkZkOperator({ algorithm, fetcher, allowUnpinnedKeys: true })
	const jobs = []
	spec.items.forEach((it, k) => {
		const key = u8(it.key)
		for(const publicInput of zk.splitToChunks(algorithm, { ciphertext: u8(it.ciphertext), iv: u8(it.iv) })) {
			jobs.push({ item: k, algorithm, operator, privateInput: { key }, publicInput })
		}
	})
	await operator.groth16ProveBatch([])          // InitAlgorithm outside the timed region
	const t0 = Date.now()
	const proofs = await zk.generateProofBatch(jobs)
	const t1 = Date.now()
	// the host-side verifier takes ~70 ms per AES proof: check every `verifyStride`-th proof and the last one
	const stride = spec.verifyStride || 1
	let verified = 0
	for(let k = 0; k < proofs.length; k++) {
		if(k % stride && k !== proofs.length - 1) {
			continue
		}

		await zk.verifyProof({ proof: proofs[k], publicInput: jobs[k].publicInput, operator })
		verified++
	}

	const t2 = Date.now()
	let rejected = false
	try {
		const bad = Object.assign({}, proofs[0], { plaintext: Uint8Array.from(proofs[0].plaintext) })
		bad.plaintext[3] ^= 0x80
		await zk.verifyProof({ proof: bad, publicInput: jobs[0].publicInput, operator })
	} catch(err) {
		rejected = err.message === 'invalid proof'
	}

	const last = jobs[jobs.length - 1]
	return {
		items: spec.items.length, proofs: proofs.length, verified, rejected,
		proveMs: t1 - t0, verifyMs: t2 - t1,
		itemsPerS: spec.items.length / ((t1 - t0) / 1e3), proofsPerS: proofs.length / ((t1 - t0) / 1e3),
		firstPlaintext: Buffer.from(proofs[0].plaintext).toString('base64'),
		lastChunk: { offsetBytes: last.publicInput.offsetBytes, bytes: last.publicInput.ciphertext.length,
			plaintext: Buffer.from(proofs[proofs.length - 1].plaintext).toString('base64') },
		proofBytes: proofs[0].proofData.length
	}
}

main().then(r => {
	process.stdout.write('\nRESULT ' + JSON.stringify(r) + '\n')
}).catch(err => {
	process.stdout.write('\nRESULT ' + JSON.stringify({ error: String(err && err.stack || err) }) + '\n')
	process.exitCode = 1
})
