// Host-side helpers of the request path: SHA-256, base64, a small JSON reader, the two ciphers, and the
// handful of 256-bit integer operations the host needs (everything heavy runs on the GPU).
#pragma once
#include <utility>
#include <exception>
#include <mutex>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <stdint.h>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace zk {

void sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
std::string hex_encode(const uint8_t *p, size_t n);

std::string base64_encode(const uint8_t *p, size_t n);
bool base64_decode(const std::string &s, std::vector<uint8_t> &out);   // std encoding, padding required as Go's

// ---- JSON (RFC 8259 subset sufficient for Go's encoding/json request shapes) ----
struct Json {
    enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    bool b = false;
    double num = 0;
    std::string raw_num;   // textual number, for exact integer checks
    std::string s;
    std::vector<Json> a;
    std::vector<std::pair<std::string, Json>> o;
    // Go's json.Unmarshal matches keys case-insensitively, preferring an exact match
    const Json *get(const char *key) const;
};
// returns false and fills err on malformed input; err_offset (optional) = Go's json.SyntaxError.Offset for the same input
bool json_parse(const char *p, size_t n, Json &out, std::string &err, size_t *err_offset = nullptr);
std::string json_quote(const std::string &s);   // Go json.Marshal string escaping (HTML-safe form)

// boundaries [lo, hi) of the elements of a top-level JSON array, found by a scanner that only tracks strings and nesting (the elements
// are parsed later, independently and in parallel).  false + Go's SyntaxError offset on malformed structure; not_array: the top-level
// value is something else
bool json_split_array(const char *p, size_t n, std::vector<std::pair<size_t, size_t>> &el, size_t &err_off, bool &not_array);

// host-side parallel loop (request decoding, witness packing, proof serialisation are independent per request).  The loops run on a
// small persistent pool (at most 15 workers + the caller, started on first use): creating 15 threads per call cost ~0.5 ms, half of
// what a 1 024-request chunk needs in all.  Several callers may loop at the same time (the decode stage and the serialise stage of a
// streamed ProveBatch do); an exception thrown by fn reaches the caller.
void parallel_for_impl(size_t n, void (*call)(void *ctx, size_t i), void *ctx);
template <class Fn>
inline void parallel_for(size_t n, Fn fn) {
    if (n < 64) { for (size_t i = 0; i < n; i++) fn(i); return; }
    parallel_for_impl(n, [](void *ctx, size_t i) { (*static_cast<Fn *>(ctx))(i); }, &fn);
}


// ---- ciphers (the reference computes the ciphertext itself: provers.go:139-150, 227-238) ----
void chacha20_block(const uint8_t key[32], uint32_t counter, const uint8_t nonce[12], uint8_t out[64]);
// key_len 16 or 32
void aes_encrypt_block(const uint8_t *key, size_t key_len, const uint8_t in[16], uint8_t out[16]);

// ---- 256-bit helpers: little-endian 8 x u32 limbs ----
struct U256 {
    uint32_t l[8];
};
extern const U256 FR_MODULUS, FP_MODULUS, FP_HALF;   // r, p, (p-1)/2
int u256_cmp(const U256 &a, const U256 &b);
// big-endian bytes (any length) reduced mod m (m < 2^255)
U256 u256_from_be_mod(const uint8_t *p, size_t n, const U256 &m);
void u256_to_be(const U256 &v, uint8_t out[32]);
// uniformly random element of [0, r) from the OS entropy source (fr.Element.SetRandom)
U256 fr_random();
void fr_random_fill(U256 *out, size_t n);
// RFC 9380 expand_message_xmd(SHA-256): out_len <= 255 * 32 uniform bytes; false when dst / out_len exceed the RFC's limits
bool expand_message_xmd(const uint8_t *msg, size_t len, const uint8_t *dst, size_t dlen, uint8_t *out, size_t out_len);
// ... -> one Fr element (gnark-crypto fr.Hash(msg, dst, 1)), canonical
U256 hash_to_fr(const uint8_t *msg, size_t len, const char *dst);

}  // namespace zk
