// see request.h
#include "request.h"
#include <string.h>

namespace zk {

const AlgInfo ALGS[3] = {
    {"chacha20", "e74dbfb793af8e3b62f8be035fa20764e1076ac415057204dbc1336e3079d12f", "517cdf7642c84dcb165a29f0accea413178fc11b21d8998461156a1acce6ebbd", 0},
    {"aes-128-ctr", "91c509d3324b7c0c8eaf7cba9b0910a4d880533d1e613d37b48b8747c8f2412b", "8ef58f13706568eed44f9aa40b97da7c2f2a7f1bce47b7bfe52f7df860a049fb", 1},
    {"aes-256-ctr", "bc8277d9f9a3e06b411210a5b78fc05146529971ef93e7a58d48cd7c24e17586", "7f589a6d1a24efa6d01f663c11b671e0abef196c15cadb93d2c6b226df7bf0dd", 1},
};
const char *const TOPRF_NAMES[3] = {"chacha20-toprf", "aes-128-ctr-toprf", "aes-256-ctr-toprf"};

[[noreturn]] void panic_str(const std::string &msg) { throw Panic{json_quote(msg)}; }
[[noreturn]] void panic_err() { throw Panic{"{}"}; }

namespace {

bool get_bytes(const Json *j, std::vector<uint8_t> &out) {     // Go: []uint8 <- base64 string | null
    out.clear();
    if (!j || j->t == Json::NUL) return true;
    if (j->t == Json::STR) return base64_decode(j->s, out);
    if (j->t == Json::ARR) {                                     // Go also accepts an array of numbers
        for (auto &e : j->a) {
            if (e.t != Json::NUM || e.num < 0 || e.num > 255 || e.num != (double)(int)e.num) return false;
            out.push_back((uint8_t)e.num);
        }
        return true;
    }
    return false;
}
bool get_u32(const Json *j, uint32_t &out) {
    out = 0;
    if (!j || j->t == Json::NUL) return true;
    if (j->t != Json::NUM) return false;
    for (char ch : j->raw_num) if (ch < '0' || ch > '9') return false;   // Go rejects -1, 1.5, 1e3 for uint32
    if (j->num > 4294967295.0) return false;
    out = (uint32_t)strtoull(j->raw_num.c_str(), nullptr, 10);
    return true;
}
void keep(const std::vector<uint8_t> &v, uint8_t *dst, size_t cap, size_t &len) {
    len = v.size();
    if (!v.empty()) memcpy(dst, v.data(), v.size() < cap ? v.size() : cap);
}
int cipher_id(const std::string &cipher) {
    for (int i = 0; i < 3; i++) if (cipher == ALGS[i].name) return i;
    for (int i = 0; i < 3; i++)
        if (cipher == TOPRF_NAMES[i]) panic_str("proving params are not initialized for cipher: " + cipher);   // never initialisable here
    panic_str("could not find prover for" + cipher);            // library.go:158 (sic, no space)
}

}  // namespace

// impl/library.go:135-145: decode + dispatch
Request decode_request(const Json &j) {
    if (j.t == Json::NUL) panic_err();      // json "null" leaves *InputParams nil -> nil dereference panic (runtime.Error -> {})
    if (j.t != Json::OBJ) panic_err();
    Request rq;
    std::string cipher;
    if (const Json *c = j.get("cipher")) {
        if (c->t == Json::STR) cipher = c->s;
        else if (c->t != Json::NUL) panic_err();
    }
    std::vector<uint8_t> tmp;
    if (!get_bytes(j.get("key"), tmp)) panic_err();
    keep(tmp, rq.key, sizeof(rq.key), rq.key_len);
    if (!get_bytes(j.get("input"), tmp)) panic_err();
    keep(tmp, rq.input, sizeof(rq.input), rq.input_len);
    if (const Json *bl = j.get("blocks")) {
        if (bl->t == Json::ARR) {
            for (auto &b : bl->a) {
                uint32_t ctr = 0, boundary = 0;
                tmp.clear();
                if (b.t == Json::OBJ) {
                    if (!get_bytes(b.get("nonce"), tmp) || !get_u32(b.get("counter"), ctr) || !get_u32(b.get("boundary"), boundary)) panic_err();
                } else if (b.t != Json::NUL) panic_err();
                if (rq.nblocks < (size_t)AES_BLOCKS) {
                    keep(tmp, rq.nonce[rq.nblocks], 12, rq.nonce_len[rq.nblocks]);
                    rq.counter[rq.nblocks] = ctr;
                }
                rq.nblocks++;
            }
        } else if (bl->t != Json::NUL) panic_err();
    }
    if (const Json *t = j.get("toprf")) {
        if (t->t != Json::OBJ && t->t != Json::NUL) panic_err();
    }
    rq.alg = cipher_id(cipher);
    return rq;
}

// ---------------------------------------------------------------------------------------------------- scanner
namespace {
struct Scan {
    const char *p, *e;
    void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    // a string without escapes or control characters: [lo, hi) of its content
    bool plain_string(const char *&lo, const char *&hi) {
        if (p >= e || *p != '"') return false;
        lo = ++p;
        while (p < e && *p != '"') {
            const unsigned char c = (unsigned char)*p;
            if (c < 0x20 || c == '\\' || c >= 0x80) return false;      // escapes / non-ASCII: the DOM reader's business
            p++;
        }
        if (p >= e) return false;
        hi = p++;
        return true;
    }
    // digits only, no leading zero unless the number is 0, at most 10 digits, <= 2^32 - 1
    bool u32(uint32_t &v) {
        const char *s = p;
        uint64_t x = 0;
        while (p < e && *p >= '0' && *p <= '9' && p - s < 11) x = x * 10 + (uint64_t)(*p++ - '0');
        if (p == s || p - s > 10 || (s[0] == '0' && p - s > 1) || x > 0xFFFFFFFFull) return false;
        if (p < e && (*p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+' || (*p >= '0' && *p <= '9'))) return false;
        v = (uint32_t)x;
        return true;
    }
    bool lit(const char *w) {
        const size_t n = strlen(w);
        if ((size_t)(e - p) < n || memcmp(p, w, n) != 0) return false;
        p += n;
        return true;
    }
};
// strict standard base64 (padding required, no line breaks) into a bounded buffer; true length in len
bool b64_fixed(const char *lo, const char *hi, uint8_t *dst, size_t cap, size_t &len) {
    static const int8_t *MAP = [] {
        static int8_t m[256];
        memset(m, -1, sizeof(m));
        const char *B = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; i++) m[(uint8_t)B[i]] = (int8_t)i;
        return m;
    }();
    const size_t n = (size_t)(hi - lo);
    if (n % 4) return false;
    len = 0;
    for (size_t i = 0; i < n; i += 4) {
        int v[4], pad = 0;
        for (int k = 0; k < 4; k++) {
            const char ch = lo[i + k];
            if (ch == '=') {
                if (i + 4 != n || k < 2) return false;
                v[k] = 0; pad++;
            } else {
                if (pad) return false;
                v[k] = MAP[(uint8_t)ch];
                if (v[k] < 0) return false;
            }
        }
        const uint32_t x = ((uint32_t)v[0] << 18) | ((uint32_t)v[1] << 12) | ((uint32_t)v[2] << 6) | (uint32_t)v[3];
        const uint8_t o[3] = {(uint8_t)(x >> 16), (uint8_t)(x >> 8), (uint8_t)x};
        for (int k = 0; k < 3 - pad; k++) {
            if (len < cap) dst[len] = o[k];
            len++;
        }
    }
    return true;
}
bool ieq_known(const char *lo, const char *hi, const char *name) {
    const size_t n = strlen(name);
    if ((size_t)(hi - lo) != n) return false;
    for (size_t i = 0; i < n; i++) {
        char c = lo[i];
        if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
        if (c != name[i]) return false;
    }
    return true;
}
bool exact(const char *lo, const char *hi, const char *name) { return (size_t)(hi - lo) == strlen(name) && memcmp(lo, name, hi - lo) == 0; }
}  // namespace

bool scan_request(const char *data, size_t n, Request &rq) {
    Scan s{data, data + n};
    rq = Request();
    const char *cipher_lo = nullptr, *cipher_hi = nullptr;
    unsigned seen = 0;                       // bit per known field: a second occurrence is the DOM reader's business (later wins)
    static const char *KNOWN[5] = {"cipher", "key", "blocks", "input", "toprf"};
    s.ws();
    if (s.p >= s.e || *s.p != '{') return false;
    s.p++;
    s.ws();
    if (s.p < s.e && *s.p == '}') return false;                       // {}: no cipher -> a panic text the DOM path words
    for (;;) {
        s.ws();
        const char *klo, *khi;
        if (!s.plain_string(klo, khi)) return false;
        s.ws();
        if (s.p >= s.e || *s.p != ':') return false;
        s.p++;
        s.ws();
        int field = -1;
        for (int f = 0; f < 5; f++) {
            if (exact(klo, khi, KNOWN[f])) field = f;
            else if (ieq_known(klo, khi, KNOWN[f])) return false;     // "Cipher": Go matches it; the DOM reader knows how
        }
        if (field >= 0) {
            if (seen & (1u << field)) return false;
            seen |= 1u << field;
        }
        const char *lo, *hi;
        switch (field) {
            case 0:
                if (!s.plain_string(cipher_lo, cipher_hi)) return false;
                break;
            case 1:
                if (!s.plain_string(lo, hi) || !b64_fixed(lo, hi, rq.key, sizeof(rq.key), rq.key_len)) return false;
                break;
            case 3:
                if (!s.plain_string(lo, hi) || !b64_fixed(lo, hi, rq.input, sizeof(rq.input), rq.input_len)) return false;
                break;
            case 2: {
                if (s.p >= s.e || *s.p != '[') return false;
                s.p++;
                s.ws();
                if (s.p < s.e && *s.p == ']') { s.p++; break; }
                for (;;) {
                    s.ws();
                    if (s.p >= s.e || *s.p != '{') return false;
                    s.p++;
                    unsigned bseen = 0;
                    uint8_t nonce[12];
                    size_t nlen = 0;
                    uint32_t ctr = 0;
                    s.ws();
                    if (s.p < s.e && *s.p == '}') s.p++;
                    else
                        for (;;) {
                            s.ws();
                            const char *blo, *bhi;
                            if (!s.plain_string(blo, bhi)) return false;
                            s.ws();
                            if (s.p >= s.e || *s.p != ':') return false;
                            s.p++;
                            s.ws();
                            int bf = exact(blo, bhi, "nonce") ? 0 : exact(blo, bhi, "counter") ? 1 : exact(blo, bhi, "boundary") ? 2 : -1;
                            if (bf < 0 || (bseen & (1u << bf))) return false;      // unknown / repeated / differently-cased member
                            bseen |= 1u << bf;
                            if (bf == 0) {
                                if (!s.plain_string(lo, hi) || !b64_fixed(lo, hi, nonce, 12, nlen)) return false;
                            } else {
                                uint32_t v = 0;
                                if (s.p < s.e && *s.p == 'n') { if (!s.lit("null")) return false; }
                                else if (!s.u32(v)) return false;
                                if (bf == 1) ctr = v;
                            }
                            s.ws();
                            if (s.p < s.e && *s.p == ',') { s.p++; continue; }
                            if (s.p < s.e && *s.p == '}') { s.p++; break; }
                            return false;
                        }
                    if (rq.nblocks < (size_t)AES_BLOCKS) {
                        memcpy(rq.nonce[rq.nblocks], nonce, nlen < 12 ? nlen : 12);
                        rq.nonce_len[rq.nblocks] = nlen;
                        rq.counter[rq.nblocks] = ctr;
                    }
                    rq.nblocks++;
                    s.ws();
                    if (s.p < s.e && *s.p == ',') { s.p++; continue; }
                    if (s.p < s.e && *s.p == ']') { s.p++; break; }
                    return false;
                }
                break;
            }
            case 4:                                                     // "toprf": {} (what the JS client sends for these ciphers) or null
                if (s.p < s.e && *s.p == '{') { s.p++; s.ws(); if (s.p >= s.e || *s.p != '}') return false; s.p++; }
                else if (!s.lit("null")) return false;
                break;
            default:                                                    // unknown member: only the simplest values are skipped here
                if (s.p < s.e && *s.p == '"') { if (!s.plain_string(lo, hi)) return false; }
                else if (s.p < s.e && *s.p == 'n') { if (!s.lit("null")) return false; }
                else { uint32_t v; if (!s.u32(v)) return false; }
                break;
        }
        s.ws();
        if (s.p < s.e && *s.p == ',') { s.p++; continue; }
        if (s.p < s.e && *s.p == '}') { s.p++; break; }
        return false;
    }
    s.ws();
    if (s.p != s.e || !cipher_lo) return false;
    // only the three ciphers this library proves: every other name has a panic text of its own (DOM path)
    for (int i = 0; i < 3; i++)
        if (exact(cipher_lo, cipher_hi, ALGS[i].name)) rq.alg = i;
    return rq.alg >= 0;
}

Request parse_request(const char *p, size_t n, size_t base_offset) {
    Request rq;
    if (scan_request(p, n, rq)) return rq;
    Json j;
    std::string err;
    size_t off = 0;
    // json.Unmarshal fails with *json.SyntaxError, impl.Prove panics with it and libprove.go:34-44 marshals it: {"Offset":n}
    if (!json_parse(p, n, j, err, &off)) throw Panic{"{\"Offset\":" + std::to_string(base_offset + off) + "}"};
    return decode_request(j);
}

// ---------------------------------------------------------------------------------------------------- witness record
void build_raw(const Request &rq, RawWitness &raw, uint8_t output[128], size_t &out_len) {
    const AlgInfo &ai = ALGS[rq.alg];
    auto len_s = [](size_t n) { return std::to_string(n); };
    memset(raw.b, 0, sizeof(raw.b));
    auto put32 = [](uint8_t *d, uint32_t v) { d[0] = (uint8_t)v; d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)(v >> 16); d[3] = (uint8_t)(v >> 24); };
    if (ai.kind == 0) {
        if (rq.key_len != 32) panic_str("key length must be 32: " + len_s(rq.key_len));
        if (rq.nblocks != (size_t)CHACHA_BLOCKS) panic_str("blocks array length must be " + len_s(CHACHA_BLOCKS) + ": " + len_s(rq.nblocks));
        for (size_t i = 0; i < rq.nblocks; i++)
            if (rq.nonce_len[i] != 12) panic_str("block[" + len_s(i) + "] nonce length must be 12: " + len_s(rq.nonce_len[i]));
        if (rq.input_len != 64 * CHACHA_BLOCKS) panic_str("input length must be " + len_s(64 * CHACHA_BLOCKS) + ": " + len_s(rq.input_len));
        out_len = 64 * CHACHA_BLOCKS;
        for (int b = 0; b < CHACHA_BLOCKS; b++) {
            uint8_t ks[64];
            chacha20_block(rq.key, rq.counter[b], rq.nonce[b], ks);
            for (int i = 0; i < 64; i++) output[64 * b + i] = rq.input[64 * b + i] ^ ks[i];
            put32(raw.b + 4 * b, rq.counter[b]);
            memcpy(raw.b + 8 + 12 * b, rq.nonce[b], 12);
        }
        memcpy(raw.b + 32, rq.input, 128);
        memcpy(raw.b + 160, output, 128);
        memcpy(raw.b + 288, rq.key, 32);
    } else {
        if (rq.key_len != 32 && rq.key_len != 16) panic_str("key length must be 16 or 32: " + len_s(rq.key_len));
        if (rq.nblocks != (size_t)AES_BLOCKS) panic_str("blocks array length must be " + len_s(AES_BLOCKS) + ": " + len_s(rq.nblocks));
        for (size_t i = 0; i < rq.nblocks; i++)
            if (rq.nonce_len[i] != 12) panic_str("block[" + len_s(i) + "] nonce length must be 12: " + len_s(rq.nonce_len[i]));
        if (rq.input_len != 16 * AES_BLOCKS) panic_str("input length must be " + len_s(16 * AES_BLOCKS) + ": " + len_s(rq.input_len));
        // the circuit is compiled for one key size: a 16-byte key sent to aes-256-ctr (or vice versa) makes
        // frontend.NewWitness fail in the reference (schema mismatch) -> panic(err) -> {}
        if ((rq.alg == 1) != (rq.key_len == 16)) panic_err();
        out_len = 16 * AES_BLOCKS;
        for (int b = 0; b < AES_BLOCKS; b++) {
            uint8_t iv[16], ks[16];
            memcpy(iv, rq.nonce[b], 12);
            iv[12] = (uint8_t)(rq.counter[b] >> 24); iv[13] = (uint8_t)(rq.counter[b] >> 16); iv[14] = (uint8_t)(rq.counter[b] >> 8); iv[15] = (uint8_t)rq.counter[b];
            aes_encrypt_block(rq.key, rq.key_len, iv, ks);
            for (int i = 0; i < 16; i++) output[16 * b + i] = rq.input[16 * b + i] ^ ks[i];
            memcpy(raw.b + 12 * b, rq.nonce[b], 12);
            put32(raw.b + 60 + 4 * b, rq.counter[b]);
        }
        memcpy(raw.b + 80, rq.input, 80);
        memcpy(raw.b + 160, output, 80);
        memcpy(raw.b + 240, rq.key, rq.key_len);
    }
}

void expand_raw(int alg, const RawWitness &raw, std::vector<uint32_t> &w) {
    const uint8_t *r = raw.b;
    w.clear();
    auto le32 = [](const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    auto be32 = [](const uint8_t *p) { return (uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24); };
    if (ALGS[alg].kind == 0) {
        w.reserve(2560);
        auto bits = [&](uint32_t v) { for (int i = 0; i < 32; i++) w.push_back((v >> i) & 1); };       // utils/bytes.go:32-48
        for (int b = 0; b < CHACHA_BLOCKS; b++) bits(le32(r + 4 * b));                                  // Counter [2][32]
        for (int b = 0; b < CHACHA_BLOCKS; b++) for (int k = 0; k < 3; k++) bits(le32(r + 8 + 12 * b + 4 * k));   // Nonce [2][3][32], LE words
        for (int k = 0; k < 32; k++) bits(be32(r + 32 + 4 * k));                                        // In, BE words
        for (int k = 0; k < 32; k++) bits(be32(r + 160 + 4 * k));                                       // Out
        for (int k = 0; k < 8; k++) bits(le32(r + 288 + 4 * k));                                        // Key (secret), LE words
    } else {
        const size_t klen = alg == 1 ? 16 : 32;
        for (int i = 0; i < 60; i++) w.push_back(r[i]);                                                 // Nonce [5][12]
        for (int b = 0; b < AES_BLOCKS; b++) w.push_back(le32(r + 60 + 4 * b));                         // Counter [5]
        for (int i = 0; i < 80; i++) w.push_back(r[80 + i]);                                            // In
        for (int i = 0; i < 80; i++) w.push_back(r[160 + i]);                                           // Out
        for (size_t i = 0; i < klen; i++) w.push_back(r[240 + i]);                                      // Key (secret)
    }
}

std::string response_json(const uint8_t *proof, size_t proof_len, const uint8_t *ct, size_t ct_len) {
    std::string o;
    o.reserve(40 + (proof_len + ct_len) * 4 / 3 + 8);
    o += "{\"proof\":\"";
    o += base64_encode(proof, proof_len);
    o += "\",\"ciphertext\":\"";
    o += base64_encode(ct, ct_len);
    o += "\"}";
    return o;
}

}  // namespace zk
