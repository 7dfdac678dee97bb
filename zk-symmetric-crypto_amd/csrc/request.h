// Request layer of the prover's C ABI: JSON request -> checked, ciphered, packed witness record.
//
// Replaces gnark/libraries/prover/impl/library.go:135-160 (json.Unmarshal into InputParams, dispatch by cipher) and
// impl/provers.go:48-96 (InputParams / Block), :116-173 (ChaChaProver.Prove up to the witness) and :199-263 (AESProver.Prove),
// utils/bytes.go:12-48 (bit packing).  Pure host code (no HIP): the unit is also built under AddressSanitizer (make asan).
//
// Two decoders produce the same Request: the DOM reader of hostutil.cpp, which carries Go's encoding/json semantics in full
// (case-insensitive keys, later duplicates win, nulls, arrays of numbers for []uint8, type errors), and a single-pass scanner for the
// one shape the JS client actually sends; the scanner declines (returns false) on anything it does not recognise byte for byte, and
// the DOM reader then decides -- so what a request means never depends on which of the two read it.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "hostutil.h"

namespace zk {

struct AlgInfo {
    const char *name;
    const char *key_hash;      // impl/library.go:37,42,47
    const char *circuit_hash;  // impl/library.go:38,43,48
    int kind;                  // 0 chacha, 1 aes
};
extern const AlgInfo ALGS[3];
extern const char *const TOPRF_NAMES[3];
constexpr int CHACHA_BLOCKS = 2;   // gnark/circuits/chachaV3/common.go:8
constexpr int AES_BLOCKS = 5;      // gnark/circuits/aesV2/common.go:10

// A "panic" of the reference: carries the JSON encoding of the panic value (libprove.go:34-44)
struct Panic {
    std::string json;
};
[[noreturn]] void panic_str(const std::string &msg);
[[noreturn]] void panic_err();      // json.Marshal of an error object without exported fields: {}

// One decoded request.  Byte fields keep their TRUE length (the reference's checks print it) and the first `capacity` bytes.
struct Request {
    int alg = -1;
    uint8_t key[32];
    size_t key_len = 0;
    uint8_t input[128];
    size_t input_len = 0;
    uint8_t nonce[AES_BLOCKS][12];
    size_t nonce_len[AES_BLOCKS] = {0, 0, 0, 0, 0};
    uint32_t counter[AES_BLOCKS] = {0, 0, 0, 0, 0};
    size_t nblocks = 0;              // true number of blocks; the first AES_BLOCKS are kept
};

// impl/library.go:135-145 on the DOM: throws Panic
Request decode_request(const Json &j);
// the same for the canonical shape, straight from the bytes; false: not recognised (nothing is decided, use the DOM)
bool scan_request(const char *p, size_t n, Request &rq);
// whole-request entry: scanner first, DOM otherwise; a syntax error answers {"Offset": base_offset + position} like Go's SyntaxError
Request parse_request(const char *p, size_t n, size_t base_offset = 0);

// The witness record the device expands (solver.hip, k_assign_inputs_raw): the request's bytes and the cipher output, 320 bytes.
//   chacha20 : counter[2] u32 LE | nonce[2][12] | in[128] | out[128] | key[32]
//   aes      : nonce[5][12] | counter[5] u32 LE | in[80] | out[80] | key[16 | 32] (zero padded)
constexpr size_t RAW_STRIDE = 320;
struct RawWitness {
    uint8_t b[RAW_STRIDE];
};
// provers.go:120-150 / 203-238: the reference's checks (same order, same panic texts), the native cipher, the record.
// out_len = 128 / 80 bytes of cipher output in `output`
void build_raw(const Request &rq, RawWitness &raw, uint8_t output[128], size_t &out_len);
// the witness assignment gnark's frontend.NewWitness builds from it (public part first, without the constant one): the u32 form of the
// benchmark / test entries; mirrors the device kernel
void expand_raw(int alg, const RawWitness &raw, std::vector<uint32_t> &w);

std::string response_json(const uint8_t *proof, size_t proof_len, const uint8_t *ct, size_t ct_len);

}  // namespace zk
