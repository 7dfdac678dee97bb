'use strict'
/**
 * Driver for the operator tests (called by tests/test_js_operator.py, which prepares <dir> and checks the output).
 *   node run_operator.js witness <dir>   request JSON of every instance (host logic only)
 *   node run_operator.js prove   <dir>   single prove + batch prove + verify on the GPU, through the operator
 * <dir> holds pk.<ext>, r1cs.<ext>, vk.<ext> and instances.json = { algorithm, instances: [{key,in,out,noncesAndCounters}] } (base64 bytes).
 * Prints one JSON document on stdout (the library's own log lines go to stderr / are filtered by the caller).
 */
const fs = require('fs')
const path = require('path')
const { makeGnarkZkOperator, generateGnarkWitness } = require('../gnark-operator.js')

const [cmd, dir] = process.argv.slice(2)
const spec = JSON.parse(fs.readFileSync(path.join(dir, 'instances.json')))
const u8 = (s) => new Uint8Array(Buffer.from(s, 'base64'))
const toInput = (i) => ({
	key: u8(i.key),
	in: u8(i.in),
	out: u8(i.out),
	noncesAndCounters: i.noncesAndCounters.map(n => ({ nonce: u8(n.nonce), counter: n.counter }))
})
const fetcher = {
	async fetch(engine, name) { return fs.readFileSync(path.join(dir, name)) },
	async has(engine, name) { return fs.existsSync(path.join(dir, name)) },
}

async function main() {
	const inputs = spec.instances.map(toInput)
	if(cmd === 'witness') {
		return { requests: inputs.map(i => generateGnarkWitness(spec.algorithm, i)) }
	}

	const op = makeGnarkZkOperator({ algorithm: spec.algorithm, fetcher, allowUnpinnedKeys: true })
	const witnesses = await Promise.all(inputs.map(i => op.generateWitness(i)))
	const t0 = Date.now()
	const single = await op.groth16Prove(witnesses[0])
	const t1 = Date.now()
	const batch = await op.groth16ProveBatch(witnesses)
	const t2 = Date.now()
	// public signals of the verifier: input, output, nonces and counters -- everything but the key (verifiers.go:29-161)
	const pub = (i) => ({ out: i.out, in: i.in, noncesAndCounters: i.noncesAndCounters })
	const okSingle = await op.groth16Verify(pub(inputs[0]), single.proof)
	const okBatch = []
	for(let k = 0; k < batch.length; k++) {
		okBatch.push(await op.groth16Verify(pub(inputs[k]), batch[k].proof))
	}

	// the reference's benchmark pattern (js/src/tests/benchmark.ts:30-34): Promise.all over single groth16Prove calls.  The calls run
	// on the libuv pool (UV_THREADPOOL_SIZE threads) and the library coalesces the concurrent ones into device batches.
	const t3 = Date.now()
	const conc = await Promise.all(witnesses.map(w => op.groth16Prove(w)))
	const t4 = Date.now()
	const okConcurrent = []
	for(let k = 0; k < conc.length; k++) {
		okConcurrent.push(await op.groth16Verify(pub(inputs[k]), conc[k].proof))
	}

	// the batch verifier: all proofs of the batch in ONE call (fewer than 16 requests are checked one by one inside the library, from
	// 16 on by one pairing-product equation on the GPU), and again with member 1 tampered -- element i must equal groth16Verify's answer
	const items = batch.map((p, k) => ({ publicSignals: pub(inputs[k]), proof: p.proof }))
	const okVerifyBatch = await op.groth16VerifyBatch(items)
	const itemsBad = items.map((it, k) => k === 1 ? { publicSignals: pub(inputs[0]), proof: it.proof } : it)
	const okVerifyBatchTampered = await op.groth16VerifyBatch(itemsBad)

	const tampered = pub(inputs[0])
	tampered.out = Uint8Array.from(tampered.out)
	tampered.out[0] ^= 1
	const okTampered = await op.groth16Verify(tampered, single.proof)
	let badErr = null
	try {
		await op.groth16Prove(Buffer.from(JSON.stringify({ cipher: spec.algorithm, key: 'AAAA', blocks: [], input: '' })))
	} catch(err) {
		badErr = err.message
	}

	const enc = (p) => Buffer.from(p.proof).toString('base64')
	return {
		single: enc(single), batch: batch.map(enc), okSingle, okBatch, okTampered, badErr, okVerifyBatch, okVerifyBatchTampered,
		msSingle: t1 - t0, msBatch: t2 - t1, okConcurrent, msConcurrent: t4 - t3, poolSize: process.env.UV_THREADPOOL_SIZE || '4'
	}
}

main().then(r => {
	process.stdout.write('\nRESULT ' + JSON.stringify(r) + '\n')
}).catch(err => {
	process.stdout.write('\nRESULT ' + JSON.stringify({ error: String(err && err.stack || err) }) + '\n')
	process.exitCode = 1
})
