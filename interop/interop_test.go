// Interoperability test for maintainers with a Go toolchain, the reference checkout and the real proving keys.
// NOT compiled or run in this repository's image (no Go, gnark v0.14.0 / gnark-crypto v0.19.2 are not vendored) -- it is the
// recipe that pins, against real gnark, the three conventions this repository could only restate from the published
// sources (DESIGN.md section 5):
//
//   1. the ProvingKey.WriteTo layout  -- InitAlgorithm must accept resources/gnark/pk.* with its pinned SHA-256
//      (gnark/libraries/prover/impl/library.go:35-49);
//   2. the BSB22 commitment challenge -- hash_to_field over commitment.Marshal() with DST "bsb22-commitment", and the
//      Pedersen proof of knowledge folding (AES circuits only);
//   3. the compressed-point flag polarity and G2 coordinate order of proof / key points.
//
// If all three match, proofs made by libzkprove.so verify under the REFERENCE verifier (verifier.Verify,
// gnark/libraries/verifier/impl/library.go:162-182 -- the check of gnark/libraries/test_helpers.go:332-339); if one does
// not, InitAlgorithm returns 0 with a diagnostic naming the key section and offset (1, 3) or Verify returns false (2, 3).
//
// Place this file in <reference>/gnark/libraries/ (package libraries_test), put libzkprove.so and include/zkprove.h where
// the cgo flags below point, then:   go test -run TestLibzkproveInterop ./libraries/
package libraries_test

/*
#cgo CFLAGS: -I${SRCDIR}/../../../zkprove/include
#cgo LDFLAGS: -L${SRCDIR}/../../../zkprove/zk-symmetric-crypto_amd -lzkprove -Wl,-rpath,${SRCDIR}/../../../zkprove/zk-symmetric-crypto_amd
#include <stdlib.h>
#include "zkprove.h"
*/
import "C"

import (
	"bytes"
	"crypto/rand"
	"encoding/json"
	"errors"
	"os"
	"testing"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/frontend"

	aes_v2 "github.com/reclaimprotocol/zk-symmetric-crypto/gnark/circuits/aesV2"
	"github.com/reclaimprotocol/zk-symmetric-crypto/gnark/circuits/chachaV3"
	prover "github.com/reclaimprotocol/zk-symmetric-crypto/gnark/libraries/prover/impl"
	verifier "github.com/reclaimprotocol/zk-symmetric-crypto/gnark/libraries/verifier/impl"
	"github.com/reclaimprotocol/zk-symmetric-crypto/gnark/utils"
)

func goSlice(b []byte) C.GoSlice {
	return C.GoSlice{data: unsafe.Pointer(&b[0]), len: C.longlong(len(b)), cap: C.longlong(len(b))}
}

func gpuInit(t *testing.T, id uint8, ext string) {
	pk, err := os.ReadFile("../../resources/gnark/pk." + ext)
	if err != nil {
		t.Skipf("proving key not present: %v", err)
	}
	r1cs, err := os.ReadFile("../../resources/gnark/r1cs." + ext)
	if err != nil {
		t.Fatal(err)
	}
	// convention 1 (and 3 for the key points): the pinned-hash key must parse, every point must decompress onto its curve
	if C.InitAlgorithm(C.uchar(id), goSlice(pk), goSlice(r1cs)) != 1 {
		t.Fatalf("libzkprove InitAlgorithm(%d) refused the reference's pk.%s (see the diagnostic it printed)", id, ext)
	}
}

func gpuProve(t *testing.T, params []byte) *prover.OutputParams {
	r := C.Prove(goSlice(params))
	defer C.Free(r.r0)
	var out *prover.OutputParams
	if err := json.Unmarshal(C.GoBytes(r.r0, C.int(r.r1)), &out); err != nil || out == nil || len(out.Proof) == 0 {
		t.Fatalf("libzkprove Prove did not return a proof: %s", C.GoString((*C.char)(r.r0)))
	}
	return out
}

// gpuProveWitness is the shim of INTEGRATION.md section 2: groth16.Prove(cp.r1cs, cp.pk, wtns) replaced by ZkpProveWitness
// (gnark/libraries/prover/impl/provers.go:179,269).  Convention pinned: witness.MarshalBinary's layout (header of three big-endian
// u32, 32-byte big-endian elements, public part first without the constant one) and Proof.WriteTo / ReadFrom.
func gpuProveWitness(id uint8, wtns witness.Witness) (groth16.Proof, error) {
	wb, err := wtns.MarshalBinary()
	if err != nil {
		return nil, err
	}
	r := C.ZkpProveWitness(C.uchar(id), goSlice(wb))
	defer C.Free(r.r0)
	if r.r1 < 0 {
		return nil, errors.New(string(C.GoBytes(r.r0, C.int(-r.r1))))
	}
	proof := groth16.NewProof(ecc.BN254)
	_, err = proof.ReadFrom(bytes.NewReader(C.GoBytes(r.r0, C.int(r.r1))))
	return proof, err
}

// the ChaCha20 witness exactly as ChaChaProver.Prove assigns it (provers.go:139-173), proved through the shim and checked with
// gnark's own groth16.Verify against the vk the reference verifier embeds
func TestLibzkproveWitnessBoundary(t *testing.T) {
	gpuInit(t, 0, "chacha20")
	key := make([]byte, 32)
	nonce := make([]byte, 12)
	input := make([]byte, 128)
	rand.Read(key)
	rand.Read(nonce)
	rand.Read(input)
	params, _ := json.Marshal(&prover.InputParams{Cipher: "chacha20", Key: key, Blocks: []prover.Block{{Nonce: nonce, Counter: 1}, {Nonce: nonce, Counter: 2}}, Input: input})
	var ref *prover.OutputParams
	json.Unmarshal(prover.Prove(params), &ref) // for the ciphertext (the native cipher of the reference)
	w := &chachaV3.ChaChaCircuit{}
	for b := 0; b < 2; b++ {
		w.Counter[b] = utils.Uint32ToBits(uint32(1 + b))
		copy(w.Nonce[b][:], utils.BytesToUint32LEBits(nonce))
	}
	copy(w.In[:], utils.BytesToUint32BEBits(input))
	copy(w.Out[:], utils.BytesToUint32BEBits(ref.Ciphertext))
	copy(w.Key[:], utils.BytesToUint32LEBits(key))
	wtns, err := frontend.NewWitness(w, ecc.BN254.ScalarField())
	if err != nil {
		t.Fatal(err)
	}
	proof, err := gpuProveWitness(0, wtns)
	if err != nil {
		t.Fatalf("ZkpProveWitness: %v", err)
	}
	buf := &bytes.Buffer{}
	proof.WriteTo(buf)
	signals, _ := json.Marshal(&verifier.PublicSignalsJSON{Ciphertext: ref.Ciphertext, Blocks: []verifier.Block{{Nonce: nonce, Counter: 1}, {Nonce: nonce, Counter: 2}}, Input: input})
	req, _ := json.Marshal(&verifier.InputVerifyParams{Cipher: "chacha20", Proof: buf.Bytes(), PublicSignals: signals})
	if !verifier.Verify(req) {
		t.Fatal("reference verifier rejected the proof ZkpProveWitness made from gnark's own witness")
	}
	// a witness that does not satisfy the circuit comes back as an error, as from groth16.Prove
	wrong := append([]byte{}, ref.Ciphertext...)
	wrong[0] ^= 1
	copy(w.Out[:], utils.BytesToUint32BEBits(wrong))
	bad, _ := frontend.NewWitness(w, ecc.BN254.ScalarField())
	if _, err := gpuProveWitness(0, bad); err == nil {
		t.Fatal("ZkpProveWitness proved an unsatisfied witness")
	}
	_ = aes_v2.BLOCKS
}

func runCase(t *testing.T, id uint8, ext, cipher string, keyLen, nBlocks, blockLen int) {
	gpuInit(t, id, ext)
	key := make([]byte, keyLen)
	input := make([]byte, nBlocks*blockLen)
	nonce := make([]byte, 12)
	rand.Read(key)
	rand.Read(input)
	rand.Read(nonce)
	blocks := make([]prover.Block, nBlocks)
	for i := range blocks {
		blocks[i] = prover.Block{Nonce: nonce, Counter: uint32(7 + i)}
	}
	params, _ := json.Marshal(&prover.InputParams{Cipher: cipher, Key: key, Blocks: blocks, Input: input})
	out := gpuProve(t, params)

	// conventions 2 and 3: the REFERENCE verifier (gnark's groth16.Verify with the embedded vk.*) must accept the proof
	vblocks := make([]verifier.Block, nBlocks)
	for i := range vblocks {
		vblocks[i] = verifier.Block{Nonce: nonce, Counter: uint32(7 + i)}
	}
	signals, _ := json.Marshal(&verifier.PublicSignalsJSON{Ciphertext: out.Ciphertext, Blocks: vblocks, Input: input})
	req, _ := json.Marshal(&verifier.InputVerifyParams{Cipher: cipher, Proof: out.Proof, PublicSignals: signals})
	if !verifier.Verify(req) {
		t.Fatalf("%s: reference verifier rejected a libzkprove proof", cipher)
	}
	// and the reference prover's own proof for the same request must verify under libzkprove's Verify export as well
	ref := prover.Prove(params)
	var refOut *prover.OutputParams
	json.Unmarshal(ref, &refOut)
	req2, _ := json.Marshal(&verifier.InputVerifyParams{Cipher: cipher, Proof: refOut.Proof, PublicSignals: signals})
	if C.Verify(goSlice(req2)) != 1 {
		t.Fatalf("%s: libzkprove Verify rejected a gnark proof (put generated/vk.* beside the library or set ZKPROVE_VK_DIR)", cipher)
	}
	// the batch entries (extensions of the ABI): ProveBatch -> every proof accepted by the reference verifier; VerifyBatch over a
	// mix of libzkprove and gnark proofs -> all true, and false exactly for a tampered member (the GPU batch check falls back to
	// per-proof verification when a member fails)
	const n = 24
	arr := []json.RawMessage{}
	for i := 0; i < n; i++ {
		arr = append(arr, params)
	}
	body, _ := json.Marshal(arr)
	rb := C.ProveBatch(goSlice(body))
	var outs []*prover.OutputParams
	if err := json.Unmarshal(C.GoBytes(rb.r0, C.int(rb.r1)), &outs); err != nil || len(outs) != n {
		t.Fatalf("%s: ProveBatch answered %s", cipher, C.GoString((*C.char)(rb.r0)))
	}
	C.Free(rb.r0)
	vreqs := []json.RawMessage{}
	for _, o := range outs {
		sig, _ := json.Marshal(&verifier.PublicSignalsJSON{Ciphertext: o.Ciphertext, Blocks: vblocks, Input: input})
		r, _ := json.Marshal(&verifier.InputVerifyParams{Cipher: cipher, Proof: o.Proof, PublicSignals: sig})
		if !verifier.Verify(r) {
			t.Fatalf("%s: reference verifier rejected a ProveBatch proof", cipher)
		}
		vreqs = append(vreqs, r)
	}
	// ZkpProveBatchDeterministic (round 5): the same array under caller-supplied blinding -- two calls give byte-identical proofs, and the
	// reference verifier accepts them (the batch-size-switched kernel paths of 24 requests: bucket pipeline before, comb tables after a
	// large batch has been served)
	blind := make([]byte, 3*n*32)
	for i := 0; i < 3*n; i++ {
		blind[i*32] = byte(i + 1) // small canonical little-endian scalars
	}
	det := func() []*prover.OutputParams {
		r := C.ZkpProveBatchDeterministic(goSlice(body), C.long(n), (*C.uint8_t)(unsafe.Pointer(&blind[0])), (*C.uint8_t)(unsafe.Pointer(&blind[n*32])),
			(*C.uint8_t)(unsafe.Pointer(&blind[2*n*32])))
		var o []*prover.OutputParams
		if err := json.Unmarshal(C.GoBytes(r.r0, C.int(r.r1)), &o); err != nil || len(o) != n {
			t.Fatalf("%s: ZkpProveBatchDeterministic answered %s", cipher, C.GoString((*C.char)(r.r0)))
		}
		C.Free(r.r0)
		return o
	}
	d1, d2 := det(), det()
	for i := range d1 {
		if !bytes.Equal(d1[i].Proof, d2[i].Proof) {
			t.Fatalf("%s: deterministic batch proofs differ between two calls (element %d)", cipher, i)
		}
		sig, _ := json.Marshal(&verifier.PublicSignalsJSON{Ciphertext: d1[i].Ciphertext, Blocks: vblocks, Input: input})
		r, _ := json.Marshal(&verifier.InputVerifyParams{Cipher: cipher, Proof: d1[i].Proof, PublicSignals: sig})
		if !verifier.Verify(r) {
			t.Fatalf("%s: reference verifier rejected a deterministic-batch proof", cipher)
		}
	}
	vreqs = append(vreqs, req2) // gnark's own proof
	bad, _ := json.Marshal(&verifier.InputVerifyParams{Cipher: cipher, Proof: outs[0].Proof, PublicSignals: signals}) // proof 0 under the first request's signals is fine ...
	_ = bad
	tampered := append([]byte{}, input...)
	tampered[0] ^= 1
	sigT, _ := json.Marshal(&verifier.PublicSignalsJSON{Ciphertext: out.Ciphertext, Blocks: vblocks, Input: tampered})
	reqT, _ := json.Marshal(&verifier.InputVerifyParams{Cipher: cipher, Proof: out.Proof, PublicSignals: sigT})
	vreqs = append(vreqs, reqT) // ... a flipped input byte is not
	vbody, _ := json.Marshal(vreqs)
	rv := C.VerifyBatch(goSlice(vbody))
	var oks []bool
	if err := json.Unmarshal(C.GoBytes(rv.r0, C.int(rv.r1)), &oks); err != nil || len(oks) != len(vreqs) {
		t.Fatalf("%s: VerifyBatch answered %s", cipher, C.GoString((*C.char)(rv.r0)))
	}
	C.VFree(rv.r0)
	for i, ok := range oks {
		if ok != (i != len(oks)-1) {
			t.Fatalf("%s: VerifyBatch element %d = %v", cipher, i, ok)
		}
	}
}

func TestLibzkproveInterop(t *testing.T) {
	runCase(t, 0, "chacha20", "chacha20", 32, 2, 64)
	runCase(t, 1, "aes128", "aes-128-ctr", 16, 5, 16)
	runCase(t, 2, "aes256", "aes-256-ctr", 32, 5, 16)
}
