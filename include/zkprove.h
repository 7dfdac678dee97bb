/* libzkprove -- C ABI of the MI355X-native Groth16 prover (drop-in for the reference's cgo library
 * `bin/gnark/${platform}-${arch}-libprove.so`, built from gnark/libraries/prover/libprove.go by
 * gnark/scripts/build.sh:15-20 and loaded through koffi by js/src/gnark/utils.ts:63-92).
 *
 * Section 1 is byte-compatible with the header cgo generates for libprove.go; the JS operator
 * (js/src/gnark/operator.ts:14-47) can load this library unmodified.
 * Section 2 are additive extensions (batching, deterministic blinding, device-resident benchmark).
 * Nothing else is exported (linker version script generated from this header).  The kernel-level entry points of the parity tests
 * and the micro-benchmarks live in a library of their own, libzkprove_test.so (include/zkprove_test.h), built from the same objects.
 */
#ifndef ZKPROVE_H
#define ZKPROVE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- 1. reference ABI (libprove.go) */
typedef struct { void *data; long long len; long long cap; } GoSlice;     /* js/src/gnark/utils.ts:47-51 */
typedef struct { void *r0; long long r1; } ProveReturn;                   /* js/src/gnark/utils.ts:53-56 */

/* libprove.go:18-19 -- exported no-op that forces the cgo binding to be linked */
void enforce_binding(void);
/* libprove.go:21-24 -> impl/library.go:81-133.  algorithmID: 0 chacha20, 1 aes-128-ctr, 2 aes-256-ctr
 * (3-5 = *-toprf: not supported, returns 0).  Checks the SHA-256 pins of both artefacts
 * (impl/library.go:35-49), parses them and uploads the proving key to the GPU.  Idempotent and
 * callable concurrently (aux_test.go:40-83).  Returns 1 on success, 0 on any failure (never throws). */
unsigned char InitAlgorithm(unsigned char algorithmID, GoSlice provingKey, GoSlice r1cs);
/* libprove.go:31-48 -> impl/library.go:135-160.  params = UTF-8 JSON request
 *   {"cipher":str,"key":b64,"blocks":[{"nonce":b64,"counter":u32,"boundary":u32|null}],"input":b64,"toprf":{}}
 * returns a malloc'd JSON document {"proof":b64,"ciphertext":b64}; on failure the JSON encoding of the
 * reference's panic value (a JSON string, or {} for error objects) -- libprove.go:34-44.
 * The caller releases r0 with Free (utils.ts:223). */
/* Thread-safe.  Calls that arrive while the device is busy with another Prove of the same algorithm are proved together in the
 * next device batch (each caller still gets its own response), so N caller threads approach ProveBatch throughput. */
ProveReturn Prove(GoSlice params);
/* libprove.go:26-29 */
void Free(void *pointer);
/* libprove.go:50-86 -- TOPRF helpers; out of scope: return the JSON string
 * "toprf is not supported by libzkprove". */
ProveReturn GenerateOPRFRequestData(GoSlice params);
ProveReturn TOPRFFinalize(GoSlice params);

/* ---------------------------------------------------------------- 1b. reference verifier ABI (libverify.go)
 * gnark/libraries/verifier/libverify.go:18-26 -> verifier/impl/library.go:162-182, verifiers.go:29-161.
 * params = {"cipher":str,"proof":b64,"publicSignals":{"ciphertext":b64,"blocks":[{nonce,counter,boundary}],"input":b64}}.
 * Returns 1 iff the Groth16 proof verifies (host-side BN254 pairing); any malformed input -> 0, never throws.
 * The reference embeds vk.* at build time; here the key is supplied with ZkpInitVerifier (gnark VerifyingKey.WriteTo
 * bytes, e.g. the vk returned by ZkpSetup or one of the reference's generated/vk.* files). */
unsigned char Verify(GoSlice params);
void VFree(void *pointer);
/* libverify.go:28-64 (server side of TOPRF; out of scope like the two prover-side helpers above): bound by the unchanged loader
 * (js/src/gnark/utils.ts:89-90), return the JSON string "toprf is not supported by libzkprove"; release r0 with VFree. */
ProveReturn OPRFEvaluate(GoSlice params);
ProveReturn GenerateThresholdKeys(GoSlice params);
unsigned char ZkpInitVerifier(unsigned char algorithmID, GoSlice vk);
/* Extension: JSON array of Verify requests -> JSON array of booleans (element i answers request i as Verify would).  The requests of
 * one circuit are checked together on the GPU: one pairing-product equation under random 128-bit coefficients (two MSM-like sums over
 * the proofs' points, kSum as a batched MSM over vk.K, one Miller loop per proof on the device, one final exponentiation on the host);
 * a batch that fails is re-checked proof by proof.  Free the result with VFree.  (verifier/impl/verifiers.go:73-89,142-158 per proof) */
ProveReturn VerifyBatch(GoSlice params);
ProveReturn zkp_verify_batch_timed(GoSlice params, double *device_ms);

/* ---------------------------------------------------------------- 2. extensions */
/* Accept proving keys whose SHA-256 differs from the pinned KeyHash (impl/library.go:37,42,47).
 * Needed because the reference's pk.* files are not distributable with this repo; default 0 (strict,
 * as the reference).  The R1CS pin is always enforced. */
void ZkpAllowUnpinnedKeys(int allow);
/* Devices.  hipSetDevice is per thread, so the library binds the device itself at every entry point, whichever thread calls.
 *  - ZkpSetDevice(k) or LOCAL_RANK=k (one process per GPU, torchrun): this process uses device k only.
 *  - neither: InitAlgorithm keeps one replica of the algorithm on EVERY visible device (HIP_VISIBLE_DEVICES narrows that) and
 *    ProveBatch / coalesced Prove calls split their requests over the replicas in contiguous slices (>= 64 requests per device;
 *    independent proofs, no exchange between devices), so the unchanged callers use a whole 8-GPU node through the same ABI.
 *  - ZKPROVE_DEVICES="all" | "0,2,3" overrides both (a device may be listed twice).
 * Call ZkpSetDevice before InitAlgorithm; returns 0, or -1 for a device that does not exist.
 * ZkpDeviceReplicas(id) = number of device replicas of an initialised algorithm (0 = not initialised). */
int ZkpSetDevice(int device);
int ZkpDeviceReplicas(int algorithmID);
/* HIP multiplexes the streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4); one algorithm uses four streams, so with
 * two or three circuits loaded the parallel chains of a single Prove() share queues (AES-256: 18.6 ms at 4 queues, 17.3 at 8).  The library
 * does not touch the environment by itself; this call sets GPU_MAX_HW_QUEUES=n for the process and only has an effect BEFORE the
 * first HIP call of the process (exporting the variable before start-up is equivalent).  Returns 0, -1 for n outside 1..64. */
int ZkpSetHardwareQueues(int n);
/* Drop all initialised algorithms (frees HBM). */
void ZkpReset(void);
/* params = JSON array of Prove requests (all for the same cipher); returns a JSON array of Prove
 * responses.  One GPU batch per call: this is how callers reach GPU throughput, since the reference
 * FFI is synchronous (utils.ts:200-212; benchmark.ts:30-34 is effectively serial). */
ProveReturn ProveBatch(GoSlice params);
/* Inner boundary: the single third-party call of the reference's provers, groth16.Prove(cp.r1cs, cp.pk, wtns)
 * (gnark/libraries/prover/impl/provers.go:179 ChaCha20, :269 AES; SURVEY.md section 8b "inner boundary"), for a Go host that keeps its
 * own request handling and witness construction (frontend.NewWitness, provers.go:175,265).
 *   witness = wtns.MarshalBinary(): u32 nbPublic | u32 nbSecret | u32 n | n x 32-byte big-endian canonical field elements (public
 *             part first, without the constant-one wire), all big-endian -- gnark v0.14 backend/witness.
 *   returns   r1 >= 0: r0 = gProof.WriteTo bytes (164 B ChaCha20 / 196 B AES: what gProof.ReadFrom takes), r1 = their length;
 *             r1 <  0: r0 = UTF-8 error text of length -r1 (what groth16.Prove would have returned as `error`: witness of the wrong
 *                      shape, "constraint #n is not satisfied", algorithm not initialised).  Release r0 with Free either way.
 * Blinding r, s and the value of the AES circuits' hints.Randomize wire come from /dev/urandom as in gnark (crypto/rand);
 * ZkpProveWitnessDeterministic fixes them (canonical little-endian 32-byte scalars, NULL = zero) for parity tests. */
ProveReturn ZkpProveWitness(unsigned char algorithmID, GoSlice witness);
ProveReturn ZkpProveWitnessDeterministic(unsigned char algorithmID, GoSlice witness, const uint8_t *r32, const uint8_t *s32, const uint8_t *randomize32);
/* Prove with caller-supplied blinding (canonical little-endian 32-byte scalars r, s and the value of the
 * AES circuits' hints.Randomize wire; may be NULL = zero).  gnark samples these from crypto/rand, which is
 * the only source of non-determinism in a proof; fixing them makes proofs reproducible for parity tests. */
ProveReturn ZkpProveDeterministic(GoSlice params, const uint8_t *r32, const uint8_t *s32, const uint8_t *randomize32);

/* ProveBatch with caller-supplied blinding: r32 / s32 / randomize32 = [n][32] canonical little-endian scalars, element i for request i of
 * the array (n must equal its length; all three arrays are required).  Makes batch proofs reproducible byte for byte (parity tests of the
 * batch-size-switched code paths against the CPU oracle); everything else is ProveBatch. */
ProveReturn ZkpProveBatchDeterministic(GoSlice params, long n, const uint8_t *r32, const uint8_t *s32, const uint8_t *randomize32);

/* Groth16 key generation on the GPU (single-party; replaces gnark/keygen/keygen.go:94-132 = groth16.Setup + pk/vk WriteTo).
 * toxic192 = alpha, beta, gamma, delta, tau, sigma as canonical little-endian 32-byte scalars.  raw != 0 writes
 * uncompressed points (ProvingKey.WriteRawTo), else compressed (WriteTo).  Buffers are malloc'd; release with Free. */
int ZkpSetup(GoSlice r1cs, const uint8_t *toxic192, int raw, void **pk_out, long long *pk_len, void **vk_out, long long *vk_len);

/* Device-resident batch benchmark (bench.py).  inputs: [batch][n_inputs] u32 witness assignment
 * (public inputs without the constant one, then secret), r/s/randomize: [batch][32] canonical LE.
 * zkp_bench_step = one pass of the whole proving path over the resident batch; times7 (ms, HIP events):
 * solve, commit, ntt, msm_g1, msm_g2, finalize, total, z_segments (times7 has room for 8 floats;
 * z_segments = the dominant kernel k_msm_accumulate<Fp, true> of the quotient MSM alone).  proofs: [batch][384] canonical LE affine
 * coordinates Ar(64) Bs(128: x.a0 x.a1 y.a0 y.a1) Krs(64) commitment(64) pok(64). */
long zkp_n_inputs(int algorithmID);
int zkp_bench_upload(int algorithmID, long batch, const uint32_t *inputs, const uint8_t *r, const uint8_t *s, const uint8_t *randomize);
int zkp_bench_step(int algorithmID, float *times7);
/* nsteps passes back to back; the witness solve of pass k+1 and the proof tail of pass k-1 run on their own streams under the
 * NTT + MSM stage of pass k.
 * times8: phase times of the last pass, [6] = whole call / nsteps */
int zkp_bench_steps(int algorithmID, int nsteps, float *times8);
int zkp_bench_fetch(int algorithmID, uint8_t *proofs, uint32_t *errs);
/* Split mode: ONE proof (or a small batch) over several GPUs, the north_star's "RCCL over xGMI only for the final bucket
 * reduction" (latency mode).  One process per GPU; rank r keeps slice r of every base set of the key:
 *   ZkpSetSplit(rank, world)  before InitAlgorithm;
 *   zkp_bench_upload(...)     the same batch on every rank (same r, s);
 *   zkp_split_partials(id, out)   solve + transforms (replicated) + this rank's slices of the five MSMs;
 *                                 out = [batch][384]: partial sums Ar, Bs1, K, Z (G1 affine, 64 B each), Bs2 (G2 affine, 128 B), Montgomery LE
 *   all_gather of the records by the caller (RCCL on the GPU box, gloo in the CPU-side test harness);
 *   zkp_split_finish(id, world, gathered)   gathered = [world][batch][384]: sums them, blinds, finishes;
 *   zkp_bench_fetch(...)      every rank holds the same proofs, byte-identical to the unsplit prover's. */
int ZkpSetSplit(int rank, int world);
int zkp_split_partials(int algorithmID, uint8_t *out384);
int zkp_split_finish(int algorithmID, int world, const uint8_t *gathered);
/* how the MSMs of the current resident batch run: info8 = {1 comb path of the quotient MSM | 0 bucket pipeline, clump size k | window bits c,
 * clumps | windows, table MiB, mixed additions per proof of its walk, 1 when the wire MSMs use wire combs for their non-0/+-1 positions,
 * such positions in G1.A, wire-comb table MiB}.  Mixed comb tables (the memory budget fills the device: a prefix of clumps holds k + 1
 * bases): [1] is the smaller size k, [2] the clump count -- bases - clumps * k of them are the large ones, up to the last clump's slack */
int zkp_quotient_msm_info(int algorithmID, long *info8);
/* request -> witness assignment (host logic only, no GPU): fills inputs[n_inputs], output bytes; returns
 * 0, or -1 with the reference's panic text in zkp_last_error() */
int zkp_request_witness(GoSlice params, uint32_t *inputs, long inputs_cap, uint8_t *output, long output_cap, long *n_inputs, long *n_output);
/* serialise one 384-byte proof record (zkp_bench_fetch layout) as gnark Proof.WriteTo bytes; returns length */
long zkp_proof_serialize(const uint8_t *proof384, int has_commitment, uint8_t *out, long cap);
/* R1CS decoder check (no GPU): returns 0 and fills stats[8] = constraints, public, secret, internal,
 * instructions, calldata words, levels, coefficients */
int zkp_r1cs_stats(GoSlice r1cs, uint64_t *stats8);

/* last error text of this thread (InitAlgorithm / bench / witness-level entries); zkp_last_panic: the reference-style panic value of
 * the last failed zkp_request_witness on this thread */
const char *zkp_last_error(void);
const char *zkp_last_panic(void);
int zkp_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
