'use strict'
/**
 * Caller side of the operator: one proof per chunk of ciphertext (2 ChaCha20 blocks = 128 B, 5 AES blocks = 80 B).
 *
 * Same contract as the reference's js/src/zk.ts (`generateProof`, `verifyProof`, `generateZkWitness`, `getPublicSignals`,
 * zk.ts:13-178): the public input is ciphertext + IV + byte offset; blocks are numbered from the cipher's start counter
 * (1 for ChaCha20-Poly1305, 2 for AES-GCM, config.ts:9-58); a short last chunk is padded with empty blocks at the following
 * counters (zk.ts:122-135).  Added: `splitToChunks` + `generateProofBatch`, which turn a long record into chunk proofs and
 * prove all of them in one library call.
 */
const crypto = require('crypto')

const CONFIG = {
	'chacha20': { blockSize: 64, blocksPerChunk: 2, keySizeBytes: 32, ivSizeBytes: 12, startCounter: 1 },
	'aes-128-ctr': { blockSize: 16, blocksPerChunk: 5, keySizeBytes: 16, ivSizeBytes: 12, startCounter: 2 },
	'aes-256-ctr': { blockSize: 16, blocksPerChunk: 5, keySizeBytes: 32, ivSizeBytes: 12, startCounter: 2 },
}

const chunkSizeBytes = (alg) => CONFIG[alg].blockSize * CONFIG[alg].blocksPerChunk

function getCounterForByteOffset(algorithm, offsetInBytes) {
	const { startCounter, blockSize } = CONFIG[algorithm]
	if(offsetInBytes % blockSize !== 0) {
		throw new Error(`offset(${offsetInBytes}) must be a multiple of block size(${blockSize})`)
	}

	return startCounter + (offsetInBytes / blockSize)
}

/** keystream XOR of one block-aligned piece, counter taken from the byte offset */
function decryptCiphertext({ algorithm, key, iv, startOffset, ciphertext }) {
	const counter = getCounterForByteOffset(algorithm, startOffset)
	const ctr = Buffer.alloc(16)
	let name
	if(algorithm === 'chacha20') {
		ctr.writeUInt32LE(counter, 0)
		Buffer.from(iv).copy(ctr, 4)
		name = 'chacha20'
	} else {
		Buffer.from(iv).copy(ctr, 0)
		ctr.writeUInt32BE(counter, 12)
		name = algorithm
	}

	const c = crypto.createDecipheriv(name, Buffer.from(key), ctr)
	return new Uint8Array(Buffer.concat([c.update(Buffer.from(ciphertext)), c.final()]))
}

function concat(arrs) {
	return new Uint8Array(Buffer.concat(arrs.map(a => Buffer.from(a.buffer, a.byteOffset, a.byteLength))))
}

/**
 * publicInput: { ciphertext, iv, offsetBytes? } or an array of them; opts.key decrypts (prover side), opts.plaintext supplies the
 * claimed plaintext (verifier side).  -> { noncesAndCounters, in, out }
 */
function getPublicSignals({ publicInput, algorithm, key, plaintext }) {
	const { ivSizeBytes, blockSize } = CONFIG[algorithm]
	const expSize = chunkSizeBytes(algorithm)
	const inputs = Array.isArray(publicInput) ? publicInput : [publicInput]
	if(!inputs.length) {
		throw new Error('at least one public input is required')
	}

	const ciphertextBlocks = []
	const plaintextBlocks = []
	const noncesAndCounters = []
	const addBlock = ({ ciphertext, iv, offsetBytes = 0 }) => {
		if(iv.length !== ivSizeBytes) {
			throw new Error(`iv must be ${ivSizeBytes} bytes`)
		}

		if(ciphertext.length > blockSize) {
			throw new Error(`ciphertext must be <= ${blockSize}b`)
		}

		noncesAndCounters.push({ nonce: iv, counter: getCounterForByteOffset(algorithm, offsetBytes), boundary: undefined })
		const padded = new Uint8Array(blockSize)
		padded.set(ciphertext)
		ciphertextBlocks.push(padded)
		if(key) {
			plaintextBlocks.push(decryptCiphertext({ algorithm, key, iv, startOffset: offsetBytes, ciphertext: padded }))
		}
	}

	inputs.forEach(({ ciphertext, iv, offsetBytes = 0 }, i) => {
		for(let o = 0; o < ciphertext.length; o += blockSize) {
			addBlock({ ciphertext: ciphertext.slice(o, o + blockSize), iv, offsetBytes: offsetBytes + o })
		}

		if(i < inputs.length - 1) {
			return
		}

		const done = ciphertextBlocks.length * blockSize
		const next = offsetBytes + Math.ceil(ciphertext.length / blockSize) * blockSize
		for(let p = 0; p < expSize - done; p += blockSize) {
			addBlock({ ciphertext: new Uint8Array(), iv, offsetBytes: next + p })
		}
	})
	const pubSigs = { noncesAndCounters, in: concat(ciphertextBlocks), out: plaintext || concat(plaintextBlocks) }
	if(pubSigs.in.length !== expSize) {
		throw new Error(`Ciphertext must be exactly ${expSize}b, got ${pubSigs.in.length}b`)
	}

	return pubSigs
}

function generateZkWitness({ algorithm, privateInput: { key }, publicInput }) {
	const { keySizeBytes } = CONFIG[algorithm]
	if(key.length !== keySizeBytes) {
		throw new Error(`key must be ${keySizeBytes} bytes`)
	}

	const witness = Object.assign({ key }, getPublicSignals({ publicInput, algorithm, key }))
	return { witness, plaintextArray: witness.out }
}

async function generateProof(opts) {
	const { algorithm, operator, logger } = opts
	const { witness, plaintextArray } = generateZkWitness(opts)
	const { proof } = await operator.groth16Prove(await operator.generateWitness(witness), logger)
	return { algorithm, proofData: proof, plaintext: plaintextArray }
}

/** a record of any length -> the per-chunk public inputs a caller would prove one by one */
function splitToChunks(algorithm, { ciphertext, iv, offsetBytes = 0 }) {
	const size = chunkSizeBytes(algorithm)
	const out = []
	for(let o = 0; o < ciphertext.length; o += size) {
		out.push({ ciphertext: ciphertext.slice(o, o + size), iv, offsetBytes: offsetBytes + o })
	}

	return out
}

/** many generateProof calls, one library call: optsList[k] as for generateProof (same algorithm and operator) */
async function generateProofBatch(optsList) {
	if(!optsList.length) {
		return []
	}

	const { algorithm, operator, logger } = optsList[0]
	const built = optsList.map(o => generateZkWitness(o))
	const witnesses = await Promise.all(built.map(b => operator.generateWitness(b.witness)))
	const rslts = await operator.groth16ProveBatch(witnesses, logger)
	return rslts.map((r, k) => ({ algorithm, proofData: r.proof, plaintext: built[k].plaintextArray }))
}

async function verifyProof(opts) {
	const { proof: { proofData, algorithm, plaintext }, operator, logger, publicInput } = opts
	const publicSignals = getPublicSignals({ algorithm, plaintext, publicInput })
	if(!await operator.groth16Verify(publicSignals, proofData, logger)) {
		throw new Error('invalid proof')
	}
}

module.exports = {
	CONFIG, chunkSizeBytes, getCounterForByteOffset, decryptCiphertext, getPublicSignals, generateZkWitness,
	generateProof, generateProofBatch, splitToChunks, verifyProof
}
