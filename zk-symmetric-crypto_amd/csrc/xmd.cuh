// SHA-256 and RFC 9380 expand_message_xmd (section 5.3.1, H = SHA-256) as device functions: one thread hashes one message.
// Used by k_bsb22_challenge (prover.hip): challenge = fr.Hash(commitment.Marshal(), "bsb22-commitment", 1)[0] of gnark's Groth16
// prover (backend/groth16/bn254/prove.go) stays on the device, so the solve stage needs no host round trip.  The function is the general
// one -- any message / tag / output length the RFC allows -- so that the code that runs in the prover is the code the parity tests check
// against the RFC's published vectors (appendix K.1; tests/test_gpu_pins.py through zkp_xmd_device).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace zk {

__device__ __forceinline__ uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ inline void sha256_block(uint32_t st[8], const uint32_t w_in[16]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
        0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
        0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = w_in[i];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = sha_rotr(w[i - 15], 7) ^ sha_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = sha_rotr(w[i - 2], 17) ^ sha_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
        const uint32_t S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
        const uint32_t ch = (e & f) ^ (~e & g);
        const uint32_t t1 = h + S1 + ch + K[i] + w[i];
        const uint32_t S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
        const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// streaming SHA-256 over bytes (big-endian words, FIPS 180-4 padding)
struct Sha256Dev {
    uint32_t st[8], w[16];
    uint32_t fill;        // bytes in w
    uint32_t total;       // message bytes so far (< 2^29)
    __device__ void init() {
        st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a; st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
        for (int i = 0; i < 16; i++) w[i] = 0;
        fill = 0; total = 0;
    }
    __device__ void flush() {
        sha256_block(st, w);
        for (int i = 0; i < 16; i++) w[i] = 0;
        fill = 0;
    }
    __device__ void raw(uint8_t b) {
        w[fill >> 2] |= (uint32_t)b << (24 - 8 * (fill & 3));
        if (++fill == 64) flush();
    }
    __device__ void byte(uint8_t b) { raw(b); total++; }
    __device__ void bytes(const uint8_t *p, uint32_t n) { for (uint32_t i = 0; i < n; i++) byte(p[i]); }
    __device__ void zeros(uint32_t n) { for (uint32_t i = 0; i < n; i++) byte(0); }
    __device__ void finish(uint8_t out[32]) {
        const uint32_t bits = total * 8;
        raw(0x80);
        if (fill > 56) flush();
        w[15] = bits;            // (w[14] = 0: messages are far below 2^32 bits)
        sha256_block(st, w);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i]; }
    }
};

// RFC 9380 5.3.1: out_len <= 255 * 32 and dlen <= 255 are the caller's responsibility (checked by the host wrappers)
__device__ inline void expand_message_xmd_dev(const uint8_t *msg, uint32_t mlen, const uint8_t *dst, uint32_t dlen, uint8_t *out, uint32_t out_len) {
    const uint32_t ell = (out_len + 31) / 32;
    uint8_t b0[32], bi[32];
    Sha256Dev h;
    h.init();
    h.zeros(64);                                                   // Z_pad
    h.bytes(msg, mlen);
    h.byte((uint8_t)(out_len >> 8)); h.byte((uint8_t)out_len);     // l_i_b_str
    h.byte(0);
    h.bytes(dst, dlen); h.byte((uint8_t)dlen);                     // DST_prime
    h.finish(b0);
    for (int k = 0; k < 32; k++) bi[k] = 0;
    for (uint32_t i = 1; i <= ell; i++) {
        h.init();
        for (int k = 0; k < 32; k++) h.byte(b0[k] ^ bi[k]);
        h.byte((uint8_t)i);
        h.bytes(dst, dlen); h.byte((uint8_t)dlen);
        h.finish(bi);
        const uint32_t off = 32 * (i - 1);
        for (uint32_t k = 0; k < 32 && off + k < out_len; k++) out[off + k] = bi[k];
    }
}

}  // namespace zk
