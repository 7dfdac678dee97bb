#include "hostutil.h"
#include <mutex>
#include <stdio.h>
#include <string.h>
#include <cmath>
#include "common.h"

namespace zk {

// ---------------------------------------------------------------------------------- SHA-256
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t h[8], const uint8_t *blk) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) sha256_block(h, msg + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
}
std::string hex_encode(const uint8_t *p, size_t n) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}

// ----------------------------------------------------------------------------------- base64
static const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
std::string base64_encode(const uint8_t *p, size_t n) {
    std::string s;
    s.reserve((n + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 2 < n; i += 3) {
        uint32_t v = (p[i] << 16) | (p[i + 1] << 8) | p[i + 2];
        s += B64[v >> 18]; s += B64[(v >> 12) & 63]; s += B64[(v >> 6) & 63]; s += B64[v & 63];
    }
    if (i + 1 == n) { uint32_t v = p[i] << 16; s += B64[v >> 18]; s += B64[(v >> 12) & 63]; s += "=="; }
    else if (i + 2 == n) { uint32_t v = (p[i] << 16) | (p[i + 1] << 8); s += B64[v >> 18]; s += B64[(v >> 12) & 63]; s += B64[(v >> 6) & 63]; s += '='; }
    return s;
}
bool base64_decode(const std::string &s, std::vector<uint8_t> &out) {
    out.clear();
    int8_t map[256];
    memset(map, -1, sizeof(map));
    for (int i = 0; i < 64; i++) map[(uint8_t)B64[i]] = (int8_t)i;
    std::string t;
    for (char ch : s) if (ch != '\r' && ch != '\n') t += ch;     // Go's decoder skips newlines
    if (t.size() % 4) return false;
    for (size_t i = 0; i < t.size(); i += 4) {
        int v[4], pad = 0;
        for (int k = 0; k < 4; k++) {
            char ch = t[i + k];
            if (ch == '=') {
                if (i + 4 != t.size() || k < 2) return false;
                v[k] = 0; pad++;
            } else {
                if (pad) return false;
                v[k] = map[(uint8_t)ch];
                if (v[k] < 0) return false;
            }
        }
        uint32_t x = (v[0] << 18) | (v[1] << 12) | (v[2] << 6) | v[3];
        out.push_back((uint8_t)(x >> 16));
        if (pad < 2) out.push_back((uint8_t)(x >> 8));
        if (pad < 1) out.push_back((uint8_t)x);
    }
    return true;
}

// ------------------------------------------------------------------------------------- worker pool of parallel_for
namespace {
struct PoolJob {
    size_t n;
    void (*call)(void *, size_t);
    void *ctx;
    std::atomic<size_t> next{0}, done{0};
    int users = 0;                                     // workers inside run_some (guarded by Pool::m): the job lives on its caller's stack
    std::exception_ptr error;
    std::mutex err_lock;
    void run_some() {                                  // claims blocks of 32 indices until none are left
        for (;;) {
            const size_t i0 = next.fetch_add(32);
            if (i0 >= n) return;
            const size_t i1 = i0 + 32 < n ? i0 + 32 : n;
            try {
                for (size_t i = i0; i < i1; i++) call(ctx, i);
            } catch (...) {                            // an exception must not leave a worker: keep the first one for the caller
                std::lock_guard<std::mutex> lk(err_lock);
                if (!error) error = std::current_exception();
            }
            done.fetch_add(i1 - i0);
        }
    }
};
struct Pool {
    std::mutex m;
    std::condition_variable wake, finished;
    std::vector<PoolJob *> jobs;                       // jobs that may still have unclaimed indices
    std::vector<std::thread> workers;
    bool started = false;
    void start() {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt > 16) nt = 16;
        for (unsigned t = 1; t < nt; t++)
            workers.emplace_back([this] {
                std::unique_lock<std::mutex> lk(m);
                for (;;) {
                    wake.wait(lk, [this] { return !jobs.empty(); });
                    PoolJob *j = jobs.front();
                    if (j->next.load() >= j->n) { jobs.erase(jobs.begin()); continue; }      // exhausted: nothing left to claim
                    j->users++;
                    lk.unlock();
                    j->run_some();
                    lk.lock();
                    j->users--;
                    finished.notify_all();
                }
            });
        for (auto &w : workers) w.detach();            // they sleep on the condition variable for the life of the process
        started = true;
    }
};
Pool &pool() { static Pool *p = new Pool(); return *p; }     // never destroyed: the detached workers use it
}  // namespace
void parallel_for_impl(size_t n, void (*call)(void *ctx, size_t i), void *ctx) {
    Pool &P = pool();
    PoolJob job;
    job.n = n; job.call = call; job.ctx = ctx;
    {
        std::lock_guard<std::mutex> lk(P.m);
        if (!P.started) P.start();
        P.jobs.push_back(&job);
    }
    P.wake.notify_all();
    job.run_some();                                    // the caller works too
    {
        std::unique_lock<std::mutex> lk(P.m);
        P.finished.wait(lk, [&] { return job.done.load() >= n && job.users == 0; });
        for (size_t k = 0; k < P.jobs.size(); k++)
            if (P.jobs[k] == &job) { P.jobs.erase(P.jobs.begin() + k); break; }
    }
    if (job.error) std::rethrow_exception(job.error);
}

// ------------------------------------------------------------------------------------- JSON
namespace {
struct JP {
    const char *p, *e;
    std::string err;
    int depth = 0;
    const char *errpos = nullptr;      // where the first error was detected (the offending byte, or e at end of input)
    void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool fail(const char *m) { if (err.empty()) { err = m; errpos = p < e ? p : e; } return false; }
    // a literal (null / true / false), byte by byte so that an error points at the first byte that differs, as Go's scanner does
    bool lit(const char *w) {
        for (; *w; w++, p++) {
            if (p >= e) return fail("unexpected end of JSON input");
            if (*p != *w) return fail("invalid character in literal");
        }
        return true;
    }
    bool str(std::string &out) {
        if (p >= e || *p != '"') return fail("expected string");
        p++;
        out.clear();
        while (p < e && *p != '"') {
            unsigned char c = (unsigned char)*p;
            if (c < 0x20) return fail("control character in string");
            if (c == '\\') {
                if (++p >= e) return fail("bad escape");
                switch (*p) {
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    case '/': out += '/'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case 'u': {
                        if (e - p < 5) return fail("bad \\u escape");
                        unsigned cp = 0;
                        for (int k = 1; k <= 4; k++) {
                            char h = p[k];
                            cp <<= 4;
                            if (h >= '0' && h <= '9') cp |= h - '0';
                            else if (h >= 'a' && h <= 'f') cp |= h - 'a' + 10;
                            else if (h >= 'A' && h <= 'F') cp |= h - 'A' + 10;
                            else return fail("bad \\u escape");
                        }
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 63)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); }
                        break;
                    }
                    default: return fail("bad escape");
                }
                p++;
            } else {
                out += *p++;
            }
        }
        if (p >= e) return fail("unterminated string");
        p++;
        return true;
    }
    bool val(Json &j) {
        if (++depth > 200) return fail("nesting too deep");
        ws();
        if (p >= e) return fail("unexpected end of JSON input");
        bool ok = true;
        if (*p == '{') {
            j.t = Json::OBJ;
            p++;
            ws();
            if (p < e && *p == '}') p++;
            else
                while (true) {
                    ws();
                    std::string k;
                    if (!str(k)) { ok = false; break; }
                    ws();
                    if (p >= e || *p != ':') { ok = fail("expected ':'"); break; }
                    p++;
                    Json v;
                    if (!val(v)) { ok = false; break; }
                    j.o.emplace_back(std::move(k), std::move(v));
                    ws();
                    if (p < e && *p == ',') { p++; continue; }
                    if (p < e && *p == '}') { p++; break; }
                    ok = fail("expected ',' or '}'");
                    break;
                }
        } else if (*p == '[') {
            j.t = Json::ARR;
            p++;
            ws();
            if (p < e && *p == ']') p++;
            else
                while (true) {
                    Json v;
                    if (!val(v)) { ok = false; break; }
                    j.a.push_back(std::move(v));
                    ws();
                    if (p < e && *p == ',') { p++; continue; }
                    if (p < e && *p == ']') { p++; break; }
                    ok = fail("expected ',' or ']'");
                    break;
                }
        } else if (*p == '"') {
            j.t = Json::STR;
            ok = str(j.s);
        } else if (*p == 'n') { j.t = Json::NUL; ok = lit("null"); }
        else if (*p == 't') { j.t = Json::BOOL; j.b = true; ok = lit("true"); }
        else if (*p == 'f') { j.t = Json::BOOL; j.b = false; ok = lit("false"); }
        else if (*p == '-' || (*p >= '0' && *p <= '9')) {
            const char *s0 = p;
            if (*p == '-') p++;
            if (p >= e || !(*p >= '0' && *p <= '9')) ok = fail("bad number");
            else {
                if (*p == '0') p++; else while (p < e && *p >= '0' && *p <= '9') p++;
                if (p < e && *p == '.') { p++; if (p >= e || !(*p >= '0' && *p <= '9')) ok = fail("bad number"); while (p < e && *p >= '0' && *p <= '9') p++; }
                if (ok && p < e && (*p == 'e' || *p == 'E')) { p++; if (p < e && (*p == '+' || *p == '-')) p++; if (p >= e || !(*p >= '0' && *p <= '9')) ok = fail("bad number"); while (p < e && *p >= '0' && *p <= '9') p++; }
                if (ok) { j.t = Json::NUM; j.raw_num.assign(s0, p - s0); j.num = strtod(j.raw_num.c_str(), nullptr); }
            }
        } else ok = fail("invalid character looking for beginning of value");
        depth--;
        return ok;
    }
};
bool ieq(const std::string &a, const char *b) {
    size_t n = strlen(b);
    if (a.size() != n) return false;
    for (size_t i = 0; i < n; i++) {
        char x = a[i], y = b[i];
        if (x >= 'A' && x <= 'Z') x += 32;
        if (y >= 'A' && y <= 'Z') y += 32;
        if (x != y) return false;
    }
    return true;
}
}  // namespace

const Json *Json::get(const char *key) const {
    if (t != OBJ) return nullptr;
    const Json *found = nullptr;
    for (auto &kv : o) {     // later duplicates win, as in Go
        if (kv.first == key || ieq(kv.first, key)) found = &kv.second;
    }
    return found;
}
bool json_parse(const char *p, size_t n, Json &out, std::string &err, size_t *err_offset) {
    JP jp{p, p + n};
    // Go's json.SyntaxError.Offset = bytes read when the scanner gave up: the offending byte included, len(data) at end of input
    auto offset = [&](const char *at) { return at >= p + n ? n : (size_t)(at - p) + 1; };
    if (!jp.val(out)) { err = jp.err; if (err_offset) *err_offset = offset(jp.errpos ? jp.errpos : jp.p); return false; }
    jp.ws();
    if (jp.p != jp.e) { err = "invalid character after top-level value"; if (err_offset) *err_offset = offset(jp.p); return false; }
    return true;
}
bool json_split_array(const char *p, size_t n, std::vector<std::pair<size_t, size_t>> &el, size_t &err_off, bool &not_array) {
    size_t i = 0;
    not_array = false;
    auto ws = [&] { while (i < n && (p[i] == ' ' || p[i] == '\t' || p[i] == '\n' || p[i] == '\r')) i++; };
    ws();
    if (i >= n) { err_off = n; return false; }
    if (p[i] != '[') { not_array = true; return false; }
    i++;
    ws();
    if (i < n && p[i] == ']') {
        i++; ws();
        if (i != n) { err_off = i + 1; return false; }
        return true;
    }
    for (;;) {
        ws();
        const size_t lo = i;
        int depth = 0;
        bool in_str = false;
        for (; i < n; i++) {
            const char c = p[i];
            if (in_str) {
                if (c == '\\') i++;
                else if (c == '"') in_str = false;
                continue;
            }
            if (c == '"') in_str = true;
            else if (c == '{' || c == '[') depth++;
            else if (c == '}' || c == ']') { if (depth == 0) break; depth--; }
            else if (c == ',' && depth == 0) break;
        }
        if (i >= n) { err_off = n; return false; }               // unexpected end of JSON input
        size_t hi = i;
        while (hi > lo && (p[hi - 1] == ' ' || p[hi - 1] == '\t' || p[hi - 1] == '\n' || p[hi - 1] == '\r')) hi--;
        if (hi == lo || p[i] == '}') { err_off = i + 1; return false; }      // "[1,]" / "[,1]" / stray '}': invalid character
        el.emplace_back(lo, hi);
        if (p[i] == ',') { i++; continue; }
        i++;                                                      // ']'
        ws();
        if (i != n) { err_off = i + 1; return false; }
        return true;
    }
}
std::string json_quote(const std::string &s) {
    std::string o = "\"";
    char buf[8];
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '<': case '>': case '&':
                snprintf(buf, sizeof(buf), "\\u%04x", c); o += buf; break;
            default:
                if (c < 0x20) { snprintf(buf, sizeof(buf), "\\u%04x", c); o += buf; }
                else o += (char)c;
        }
    }
    return o + "\"";
}

// ----------------------------------------------------------------------------------- ChaCha20
static inline uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
#define QR(a, b, c, d) a += b; d = rotl(d ^ a, 16); c += d; b = rotl(b ^ c, 12); a += b; d = rotl(d ^ a, 8); c += d; b = rotl(b ^ c, 7);
void chacha20_block(const uint8_t key[32], uint32_t counter, const uint8_t nonce[12], uint8_t out[64]) {
    uint32_t st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) st[4 + i] = key[4 * i] | (key[4 * i + 1] << 8) | (key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
    st[12] = counter;
    for (int i = 0; i < 3; i++) st[13 + i] = nonce[4 * i] | (nonce[4 * i + 1] << 8) | (nonce[4 * i + 2] << 16) | ((uint32_t)nonce[4 * i + 3] << 24);
    uint32_t w[16];
    memcpy(w, st, sizeof(w));
    for (int r = 0; r < 10; r++) {
        QR(w[0], w[4], w[8], w[12]) QR(w[1], w[5], w[9], w[13]) QR(w[2], w[6], w[10], w[14]) QR(w[3], w[7], w[11], w[15])
        QR(w[0], w[5], w[10], w[15]) QR(w[1], w[6], w[11], w[12]) QR(w[2], w[7], w[8], w[13]) QR(w[3], w[4], w[9], w[14])
    }
    for (int i = 0; i < 16; i++) {
        uint32_t v = w[i] + st[i];
        out[4 * i] = (uint8_t)v; out[4 * i + 1] = (uint8_t)(v >> 8); out[4 * i + 2] = (uint8_t)(v >> 16); out[4 * i + 3] = (uint8_t)(v >> 24);
    }
}

// ---------------------------------------------------------------------------------------- AES
static uint8_t SBOX[256];
static std::once_flag sbox_once;       // aes_encrypt_block runs on up to 16 host threads (ProveBatch) and in concurrent Prove calls
static inline uint8_t xt(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1B : 0)); }
static void sbox_fill() {
    // multiplicative inverse via log tables over generator 3
    uint8_t exp[256], lg[256];
    uint8_t x = 1;
    for (int i = 0; i < 255; i++) { exp[i] = x; lg[x] = (uint8_t)i; x = (uint8_t)(x ^ xt(x)); }
    for (int a = 0; a < 256; a++) {
        uint8_t inv = a ? exp[(255 - lg[a]) % 255] : 0;
        uint8_t s = inv, r = inv;
        for (int k = 0; k < 4; k++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        SBOX[a] = s ^ 0x63;
    }
}
static void sbox_init() { std::call_once(sbox_once, sbox_fill); }
void aes_encrypt_block(const uint8_t *key, size_t key_len, const uint8_t in[16], uint8_t out[16]) {
    sbox_init();
    const int nk = (int)key_len / 4, nr = nk + 6;
    uint8_t w[15 * 16];
    memcpy(w, key, key_len);
    uint8_t rcon = 1;
    for (int i = nk; i < 4 * (nr + 1); i++) {
        uint8_t t[4];
        memcpy(t, w + 4 * (i - 1), 4);
        if (i % nk == 0) {
            uint8_t t0 = t[0];
            t[0] = SBOX[t[1]] ^ rcon; t[1] = SBOX[t[2]]; t[2] = SBOX[t[3]]; t[3] = SBOX[t0];
            rcon = xt(rcon);
        } else if (nk > 6 && i % nk == 4) {
            for (int k = 0; k < 4; k++) t[k] = SBOX[t[k]];
        }
        for (int k = 0; k < 4; k++) w[4 * i + k] = w[4 * (i - nk) + k] ^ t[k];
    }
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ w[i];
    for (int r = 1; r <= nr; r++) {
        for (int i = 0; i < 16; i++) t[i] = SBOX[s[(i + 4 * (i % 4)) % 16]];   // SubBytes + ShiftRows
        if (r != nr) {
            for (int c = 0; c < 4; c++) {
                uint8_t *a = t + 4 * c;
                uint8_t x = a[0] ^ a[1] ^ a[2] ^ a[3], a0 = a[0];
                uint8_t b0 = a[0] ^ x ^ xt(a[0] ^ a[1]), b1 = a[1] ^ x ^ xt(a[1] ^ a[2]), b2 = a[2] ^ x ^ xt(a[2] ^ a[3]), b3 = a[3] ^ x ^ xt(a[3] ^ a0);
                a[0] = b0; a[1] = b1; a[2] = b2; a[3] = b3;
            }
        }
        for (int i = 0; i < 16; i++) s[i] = t[i] ^ w[16 * r + i];
    }
    memcpy(out, s, 16);
}

// ------------------------------------------------------------------------------------- U256
const U256 FR_MODULUS = {{0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}};
const U256 FP_MODULUS = {{0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}};
const U256 FP_HALF = {{0x6c3e7ea3u, 0x9e10460bu, 0xb438e546u, 0xcbc0b548u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u}};
int u256_cmp(const U256 &a, const U256 &b) {
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] > b.l[i]) return 1;
        if (a.l[i] < b.l[i]) return -1;
    }
    return 0;
}
static void u256_sub(U256 &a, const U256 &b) {
    uint64_t br = 0;
    for (int i = 0; i < 8; i++) { uint64_t x = (uint64_t)a.l[i] - b.l[i] - br; a.l[i] = (uint32_t)x; br = (x >> 63) & 1; }
}
U256 u256_from_be_mod(const uint8_t *p, size_t n, const U256 &m) {
    U256 acc = {{0}};
    for (size_t i = 0; i < n; i++) {
        for (int bit = 7; bit >= 0; bit--) {
            // acc = 2*acc + bit  (acc < m < 2^255 so no overflow of 256 bits)
            uint32_t c = (p[i] >> bit) & 1;
            for (int k = 0; k < 8; k++) { uint32_t nc = acc.l[k] >> 31; acc.l[k] = (acc.l[k] << 1) | c; c = nc; }
            if (u256_cmp(acc, m) >= 0) u256_sub(acc, m);
        }
    }
    return acc;
}
void u256_to_be(const U256 &v, uint8_t out[32]) {
    for (int i = 0; i < 8; i++) {
        uint32_t w = v.l[7 - i];
        out[4 * i] = w >> 24; out[4 * i + 1] = w >> 16; out[4 * i + 2] = w >> 8; out[4 * i + 3] = w;
    }
}
// Uniform scalars in [0, r) from the OS entropy source by REJECTION sampling, as gnark-crypto's fr.Element.SetRandom does (ecc/bn254/fr
// element.go: 32 random bytes, the two bits above bit 253 cleared, the candidate re-drawn while it is >= r; acceptance 0.756) -- exactly
// uniform, and a compare instead of the 384-step shift-and-subtract reduction this function used to run per scalar (3 us each: at three
// scalars per proof that alone capped a streamed ProveBatch's decode stage at ~110 k requests/s).  One read per ~n candidates.
void fr_random_fill(U256 *out, size_t n) {
    if (!n) return;
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f) throw ZkError("cannot read /dev/urandom");
    setvbuf(f, nullptr, _IONBF, 0);
    std::vector<uint8_t> buf;
    size_t have = 0;
    while (have < n) {
        const size_t want = (n - have) + (n - have) / 2 + 8;        // ~ (n - have) / 0.756 candidates
        buf.resize(32 * want);
        if (fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); throw ZkError("cannot read /dev/urandom"); }
        for (size_t c = 0; c < want && have < n; c++) {
            U256 v;
            memcpy(v.l, buf.data() + 32 * c, 32);
            v.l[7] &= 0x3fffffffu;                                  // 254 bits
            if (u256_cmp(v, FR_MODULUS) < 0) out[have++] = v;
        }
    }
    fclose(f);
}
U256 fr_random() {
    U256 v;
    fr_random_fill(&v, 1);
    return v;
}
// RFC 9380 section 5.3.1 expand_message_xmd with H = SHA-256 (b_in_bytes 32, s_in_bytes 64): out_len uniform bytes from msg under dst.
// Checked against the published vectors of RFC 9380 appendix K.1 (tests/test_pins.py).
bool expand_message_xmd(const uint8_t *msg, size_t len, const uint8_t *dst, size_t dlen, uint8_t *out, size_t out_len) {
    const size_t ell = (out_len + 31) / 32;
    if (ell > 255 || out_len > 65535 || dlen > 255) return false;
    std::vector<uint8_t> b(64 + len + 3 + dlen + 1, 0);                     // Z_pad || msg || l_i_b_str || 0 || DST_prime
    memcpy(b.data() + 64, msg, len);
    b[64 + len] = (uint8_t)(out_len >> 8); b[64 + len + 1] = (uint8_t)out_len; b[64 + len + 2] = 0;
    memcpy(b.data() + 64 + len + 3, dst, dlen);
    b[64 + len + 3 + dlen] = (uint8_t)dlen;
    uint8_t b0[32], bi[32];
    sha256(b.data(), b.size(), b0);
    std::vector<uint8_t> t(32 + 1 + dlen + 1);                              // (b_0 xor b_(i-1)) || i || DST_prime
    memcpy(t.data() + 33, dst, dlen); t[33 + dlen] = (uint8_t)dlen;
    memset(bi, 0, 32);
    for (size_t i = 1; i <= ell; i++) {
        for (int k = 0; k < 32; k++) t[k] = b0[k] ^ bi[k];
        t[32] = (uint8_t)i;
        sha256(t.data(), t.size(), bi);
        const size_t off = 32 * (i - 1), take = out_len - off < 32 ? out_len - off : 32;
        memcpy(out + off, bi, take);
    }
    return true;
}
// gnark-crypto fr.Hash(msg, dst, 1)[0]: L = 48 uniform bytes per element, big-endian, reduced mod r  [gnark-mem: DESIGN.md section 5]
U256 hash_to_fr(const uint8_t *msg, size_t len, const char *dst) {
    uint8_t uni[48];
    if (!expand_message_xmd(msg, len, (const uint8_t *)dst, strlen(dst), uni, sizeof(uni))) throw ZkError("hash_to_fr: domain separation tag too long");
    return u256_from_be_mod(uni, sizeof(uni), FR_MODULUS);
}

}  // namespace zk

// ---- ZKPROVE_MEMLOG=1: device allocations >= 256 MiB with the frames that asked for them (sizing the HBM budget)
#include <execinfo.h>
namespace zk {
void require_headroom(size_t bytes) {
    constexpr size_t DEVICE_HEADROOM = (size_t)4 << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    if (free_b < bytes + DEVICE_HEADROOM)
        throw ZkError("out of device memory: " + std::to_string(bytes >> 20) + " MiB asked, " + std::to_string(free_b >> 20) + " MiB free (4096 MiB are kept for the runtime)");
}
void memlog_large_alloc(size_t bytes) {
    static const bool on = getenv("ZKPROVE_MEMLOG") != nullptr;
    if (!on) return;
    void *fr[6];
    const int nf = backtrace(fr, 6);
    char **sym = backtrace_symbols(fr, nf);
    fprintf(stderr, "zkprove: alloc %8.1f MiB", bytes / 1048576.0);
    for (int i = 1; i < nf && sym; i++) {
        const char *a = strchr(sym[i], '('), *b = a ? strchr(a, '+') : nullptr;
        if (a && b && b > a + 1) fprintf(stderr, " < %.*s", (int)(b - a - 1 > 60 ? 60 : b - a - 1), a + 1);
    }
    fprintf(stderr, "\n");
    free(sym);
}
}  // namespace zk
