'use strict'
/**
 * Node-side operator for the gnark proving path, over libzkprove.so.
 *
 * Same surface as the reference's `makeGnarkZkOperator` (js/src/gnark/operator.ts:14-47): `generateWitness`,
 * `groth16Prove`, `groth16Verify`, same request JSON (js/src/gnark/utils.ts:148-198), same error text when the
 * library returns no proof.  One addition: `groth16ProveBatch(witnesses)` hands N requests to the library in ONE call
 * (`ProveBatch`), because the reference's FFI call is synchronous and `Promise.all` over `groth16Prove` proves one
 * request at a time (js/src/tests/benchmark.ts:30-34) -- a GPU needs a batch to be busy.
 *
 * The binding is the N-API addon js/native/zkprove_napi.c (koffi, which the reference uses, is not available here).
 */
const path = require('path')

// algorithm ids of the library registry (gnark/libraries/prover/impl/library.go:17-33, js/src/config.ts)
const ALGS = {
	'chacha20': { id: 0, ext: 'chacha20' },
	'aes-128-ctr': { id: 1, ext: 'aes128' },
	'aes-256-ctr': { id: 2, ext: 'aes256' },
}

const DEFAULT_LIB = path.join(__dirname, '..', 'zk-symmetric-crypto_amd', 'libzkprove.so')

let native
const initialised = {}

function loadNative(libPath) {
	if(native) {
		return native
	}

	const addon = require(path.join(__dirname, 'zkprove_napi.node'))
	const caps = addon.open(libPath || process.env.ZKPROVE_LIB || DEFAULT_LIB)
	native = { addon, caps }
	return native
}

const b64 = (u8) => Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('base64')

/** request JSON of the library; field for field what utils.ts:148-198 builds (TOPRF requests are not supported) */
function generateGnarkWitness(cipher, input) {
	if('toprf' in input) {
		throw new Error('toprf is not supported by libzkprove')
	}

	return {
		cipher,
		key: 'key' in input ? b64(input.key) : undefined,
		ciphertext: 'out' in input && input.out && input.out.length ? b64(input.out) : undefined,
		blocks: input.noncesAndCounters.map(n => ({
			nonce: b64(n.nonce),
			counter: n.counter,
			boundary: n.boundary || null
		})),
		input: b64(input.in),
		toprf: {}
	}
}

function serialiseGnarkWitness(cipher, input) {
	return Buffer.from(JSON.stringify(generateGnarkWitness(cipher, input)))
}

async function initGnark(algorithm, fetcher, logger, opts) {
	const alg = ALGS[algorithm]
	if(!alg) {
		throw new Error(`unsupported algorithm: ${algorithm}`)
	}

	const lib = loadNative(opts && opts.libPath)
	if(initialised[alg.id]) {
		return lib
	}

	const [pk, r1cs] = await Promise.all([
		fetcher.fetch('gnark', `pk.${alg.ext}`, logger),
		fetcher.fetch('gnark', `r1cs.${alg.ext}`, logger),
	])
	if(opts && opts.allowUnpinnedKeys) {
		lib.addon.allowUnpinnedKeys(true)
	}

	if(!lib.addon.initAlgorithm(alg.id, pk, r1cs)) {
		throw new Error(`InitAlgorithm failed for ${algorithm}`)
	}

	// the reference's libverify embeds its verifying keys; this library takes them at run time
	if(lib.caps.verifier && fetcher.has && await fetcher.has('gnark', `vk.${alg.ext}`)) {
		const vk = await fetcher.fetch('gnark', `vk.${alg.ext}`, logger)
		if(!lib.addon.initVerifier(alg.id, vk)) {
			throw new Error(`could not load the verifying key for ${algorithm}`)
		}
	}

	initialised[alg.id] = true
	return lib
}

function proofFromResult(rslt) {
	if(typeof rslt !== 'object' || rslt === null || !('proof' in rslt) || !rslt.proof) {
		throw new Error(`Failed to create gnark proof: ${JSON.stringify(rslt)}`)
	}

	return { proof: new Uint8Array(Buffer.from(rslt.proof, 'base64')) }
}

/**
 * @param {{ algorithm: string, fetcher: { fetch: Function, has?: Function }, libPath?: string, allowUnpinnedKeys?: boolean }} opts
 */
function makeGnarkZkOperator(opts) {
	const { algorithm, fetcher } = opts
	return {
		async generateWitness(input) {
			return serialiseGnarkWitness(algorithm, input)
		},
		async groth16Prove(witness, logger) {
			const lib = await initGnark(algorithm, fetcher, logger, opts)
			const out = await lib.addon.proveAsync(witness)
			return proofFromResult(JSON.parse(out.toString()))
		},
		/** N witnesses (as returned by generateWitness) -> N results, one library call */
		async groth16ProveBatch(witnesses, logger) {
			const lib = await initGnark(algorithm, fetcher, logger, opts)
			if(!lib.caps.extensions) {
				return Promise.all(witnesses.map(w => this.groth16Prove(w, logger)))
			}

			const parts = [Buffer.from('[')]
			witnesses.forEach((w, i) => {
				if(i) {
					parts.push(Buffer.from(','))
				}

				parts.push(Buffer.from(w.buffer, w.byteOffset, w.byteLength))
			})
			parts.push(Buffer.from(']'))
			const out = await lib.addon.proveBatchAsync(Buffer.concat(parts))
			const rslts = JSON.parse(out.toString())
			if(!Array.isArray(rslts) || rslts.length !== witnesses.length) {
				throw new Error(`Failed to create gnark proofs: ${JSON.stringify(rslts)}`)
			}

			return rslts.map(proofFromResult)
		},
		async groth16Verify(publicSignals, proof, logger) {
			const lib = await initGnark(algorithm, fetcher, logger, opts)
			const verifyParams = JSON.stringify({
				cipher: algorithm,
				proof: typeof proof === 'string' ? proof : b64(proof),
				publicSignals: generateGnarkWitness(algorithm, publicSignals)
			})
			return lib.addon.verify(verifyParams) === true
		},
		/**
		 * N (publicSignals, proof) pairs -> N booleans, one library call: the proofs of a circuit are checked together on the GPU
		 * (`VerifyBatch`: one pairing-product equation under random coefficients; a batch that fails is re-checked proof by proof),
		 * element i equals groth16Verify(items[i].publicSignals, items[i].proof)
		 */
		async groth16VerifyBatch(items, logger) {
			const lib = await initGnark(algorithm, fetcher, logger, opts)
			if(!lib.caps.batchVerifier) {
				return Promise.all(items.map(it => this.groth16Verify(it.publicSignals, it.proof, logger)))
			}

			const body = JSON.stringify(items.map(it => ({
				cipher: algorithm,
				proof: typeof it.proof === 'string' ? it.proof : b64(it.proof),
				publicSignals: generateGnarkWitness(algorithm, it.publicSignals)
			})))
			const out = JSON.parse((await lib.addon.verifyBatchAsync(Buffer.from(body))).toString())
			if(!Array.isArray(out) || out.length !== items.length) {
				throw new Error(`Failed to verify gnark proofs: ${JSON.stringify(out)}`)
			}

			return out.map(v => v === true)
		},
	}
}

module.exports = { makeGnarkZkOperator, generateGnarkWitness, serialiseGnarkWitness, initGnark, loadNative, ALGS }
