// Groth16 verifier behind the reference's libverify C ABI.
//
// Replaces gnark/libraries/verifier/libverify.go:18-26 (`Verify`, `VFree`) and
// verifier/impl/{library.go:162-182, verifiers.go:29-161}: rebuild the public witness from the JSON public
// signals, parse the proof, run groth16.Verify (gnark v0.14, not in the mount).  Host-only C++ (one pairing check
// per proof is not a GPU workload); BN254 ate pairing over the tower Fp12 = Fp2[w]/(w^6 - (9+u)).
// The reference embeds its vk.* files at build time; here the verifying key is supplied with ZkpInitVerifier
// (keys are synthetic in this repo).
#include <dlfcn.h>
#include <string.h>
#include <memory>
#include <mutex>
#include "../../include/zkprove.h"
#include "curve.cuh"
#include "hostutil.h"
#include "common.h"

namespace zk {
namespace {

#include "pairing_consts.inc"

// ---- Fp12 = Fp2[w] / (w^6 - xi), xi = 9 + u
struct Fp12 {
    Fp2 c[6];
    static Fp12 one() { Fp12 r; for (auto &x : r.c) x = Fp2::zero(); r.c[0] = Fp2::one(); return r; }
    bool is_one() const {
        if (!(c[0] == Fp2::one())) return false;
        for (int i = 1; i < 6; i++) if (!c[i].is_zero()) return false;
        return true;
    }
};
static Fp2 XI;
static Fp2 TWIST_B;
static std::once_flag g_once;
static void consts_init() {
    std::call_once(g_once, [] {
        XI = Fp2{Fp::from_u32(9), Fp::one()};
        TWIST_B = Fp2{Fp::from_u32(3), Fp::zero()} * XI.inv();
    });
}
static Fp12 mul(const Fp12 &a, const Fp12 &b) {
    Fp2 t[11];
    for (auto &x : t) x = Fp2::zero();
    for (int i = 0; i < 6; i++) {
        if (a.c[i].is_zero()) continue;
        for (int j = 0; j < 6; j++) {
            if (b.c[j].is_zero()) continue;
            t[i + j] = t[i + j] + a.c[i] * b.c[j];
        }
    }
    Fp12 r;
    for (int k = 0; k < 6; k++) r.c[k] = t[k];
    for (int k = 6; k < 11; k++) r.c[k - 6] = r.c[k - 6] + t[k] * XI;
    return r;
}
static Fp12 pow_words(const Fp12 &a, const uint32_t *e, int nbits) {
    Fp12 acc = Fp12::one();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = mul(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, a);
    }
    return acc;
}
// line through twist point T with twist slope lam, at P in E(Fp): y_P - lam x_P w + (lam x_T - y_T) w^3
static Fp12 line(const G2Affine &T, const Fp2 &lam, const G1Affine &P) {
    Fp12 l;
    for (auto &x : l.c) x = Fp2::zero();
    l.c[0] = Fp2{P.y, Fp::zero()};
    const Fp nx = P.x.neg();
    l.c[1] = Fp2{lam.a0 * nx, lam.a1 * nx};
    l.c[3] = lam * T.x - T.y;
    return l;
}
static Fp12 miller(const G1Affine &P, const G2Affine &Q) {
    if (P.is_inf() || Q.is_inf()) return Fp12::one();
    Fp12 f = Fp12::one();
    G2Affine T = Q;
    for (int i = ATE_LOOP_BITS - 2; i >= 0; i--) {
        Fp2 x2 = T.x.sqr();
        Fp2 lam = (x2.dbl() + x2) * T.y.dbl().inv();
        f = mul(mul(f, f), line(T, lam, P));
        Fp2 x3 = lam.sqr() - T.x.dbl();
        Fp2 y3 = lam * (T.x - x3) - T.y;
        T = G2Affine{x3, y3};
        if ((ATE_LOOP_WORDS[i >> 5] >> (i & 31)) & 1) {
            if (T.x == Q.x) throw ZkError("degenerate addition in the miller loop");
            Fp2 l2 = (T.y - Q.y) * (T.x - Q.x).inv();
            f = mul(f, line(T, l2, P));
            Fp2 xa = l2.sqr() - T.x - Q.x;
            Fp2 ya = l2 * (T.x - xa) - T.y;
            T = G2Affine{xa, ya};
        }
    }
    return f;
}
static bool pairing_product_is_one(const std::vector<std::pair<G1Affine, G2Affine>> &pairs) {
    Fp12 f = Fp12::one();
    for (auto &pq : pairs) f = mul(f, miller(pq.first, pq.second));
    return pow_words(f, FINAL_EXP_WORDS, FINAL_EXP_BITS).is_one();
}

// ---- decoding (gnark-crypto marshal.go)
static const uint32_t P_PLUS1_DIV4[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
static Fp fp_from_be32(const uint8_t *p, bool mask) {
    Fp r;
    for (int i = 0; i < 8; i++) { const uint8_t *q = p + 28 - 4 * i; r.l[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; }
    if (mask) r.l[7] &= 0x3fffffffu;
    U256 c; memcpy(c.l, r.l, 32);
    if (u256_cmp(c, FP_MODULUS) >= 0) throw ZkError("coordinate not reduced");
    return r.to_mont();
}
static bool lex_largest(const Fp &y_mont) {
    Fp c = y_mont.from_mont();
    U256 t; memcpy(t.l, c.l, 32);
    return u256_cmp(t, FP_HALF) > 0;
}
static bool fp_sqrt(const Fp &a, Fp &r) { r = a.pow(P_PLUS1_DIV4, 254); return r.sqr() == a; }
static bool fp2_sqrt(const Fp2 &a, Fp2 &r) {
    if (a.is_zero()) { r = Fp2::zero(); return true; }
    if (a.a1.is_zero()) {
        Fp s;
        if (fp_sqrt(a.a0, s)) { r = Fp2{s, Fp::zero()}; return true; }
        if (fp_sqrt(a.a0.neg(), s)) { r = Fp2{Fp::zero(), s}; return true; }
        return false;
    }
    Fp n;
    if (!fp_sqrt(a.a0.sqr() + a.a1.sqr(), n)) return false;
    const Fp inv2 = Fp::from_u32(2).inv();
    for (int k = 0; k < 2; k++) {
        Fp t = (k == 0 ? a.a0 + n : a.a0 - n) * inv2, x0;
        if (!fp_sqrt(t, x0) || x0.is_zero()) continue;
        Fp2 cand{x0, a.a1 * x0.dbl().inv()};
        if (cand.sqr() == a) { r = cand; return true; }
    }
    return false;
}
static bool g1_on_curve(const G1Affine &p) { return p.is_inf() || p.y.sqr() == p.x.sqr() * p.x + Fp::from_u32(3); }
static bool g2_on_curve(const G2Affine &p) { return p.is_inf() || p.y.sqr() == p.x.sqr() * p.x + TWIST_B; }
struct Cur {
    const uint8_t *p; size_t len, pos = 0;
    void need(size_t n) const { if (pos + n > len) throw ZkError("truncated"); }
    uint32_t u32be() { need(4); uint32_t v = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3]; pos += 4; return v; }
    uint64_t u64be() { uint64_t h = u32be(); return (h << 32) | u32be(); }
    G1Affine g1() {
        need(32);
        const uint32_t flag = p[pos] >> 6;
        G1Affine r;
        if (flag == 0) { need(64); r = G1Affine{fp_from_be32(p + pos, false), fp_from_be32(p + pos + 32, false)}; pos += 64; }
        else if (flag == 1) { r = G1Affine::inf(); pos += 32; }
        else {
            Fp x = fp_from_be32(p + pos, true), y;
            if (!fp_sqrt(x.sqr() * x + Fp::from_u32(3), y)) throw ZkError("invalid G1 point");
            if (lex_largest(y) != (flag == 3)) y = y.neg();
            r = G1Affine{x, y}; pos += 32;
        }
        if (!g1_on_curve(r)) throw ZkError("G1 point not on curve");
        return r;
    }
    G2Affine g2() {
        need(64);
        const uint32_t flag = p[pos] >> 6;
        G2Affine r;
        if (flag == 0) {
            need(128);
            r.x = Fp2{fp_from_be32(p + pos + 32, false), fp_from_be32(p + pos, false)};
            r.y = Fp2{fp_from_be32(p + pos + 96, false), fp_from_be32(p + pos + 64, false)};
            pos += 128;
        } else if (flag == 1) { r = G2Affine::inf(); pos += 64; }
        else {
            Fp2 x{fp_from_be32(p + pos + 32, false), fp_from_be32(p + pos, true)}, y;
            if (!fp2_sqrt(x.sqr() * x + TWIST_B, y)) throw ZkError("invalid G2 point");
            const bool largest = y.a1.is_zero() ? lex_largest(y.a0) : lex_largest(y.a1);
            if (largest != (flag == 3)) y = y.neg();
            r = G2Affine{x, y}; pos += 64;
        }
        if (!g2_on_curve(r)) throw ZkError("G2 point not on curve");
        // the twist has cofactor 2p - r: gnark-crypto's G2Affine.SetBytes (proof.ReadFrom, the embedded vk) also rejects points
        // outside the order-r subgroup, and the Miller loop below is only a pairing on it
        if (!r.is_inf()) {
            U256 rr = FR_MODULUS;
            if (!scalar_mul(r, rr.l, 254).is_inf()) throw ZkError("G2 point not in the order-r subgroup");
        }
        return r;
    }
};

struct VerifyingKey {
    G1Affine alpha1, beta1, delta1;
    G2Affine beta2, gamma2, delta2;
    std::vector<G1Affine> K;
    std::vector<std::vector<uint64_t>> pub_committed;
    std::vector<std::pair<G2Affine, G2Affine>> ckeys;   // (G, GSigmaNeg)
};
static void vk_parse(const uint8_t *p, size_t len, VerifyingKey &vk) {
    Cur c{p, len};
    vk.alpha1 = c.g1(); vk.beta1 = c.g1(); vk.beta2 = c.g2(); vk.gamma2 = c.g2(); vk.delta1 = c.g1(); vk.delta2 = c.g2();
    uint32_t nk = c.u32be();
    if ((size_t)nk * 32 > len) throw ZkError("bad K count");
    for (uint32_t i = 0; i < nk; i++) vk.K.push_back(c.g1());
    uint32_t no = c.u32be();
    if (no > 16) throw ZkError("bad commitment count");
    for (uint32_t i = 0; i < no; i++) {
        uint32_t ni = c.u32be();
        if ((size_t)ni * 8 > len) throw ZkError("bad list");
        std::vector<uint64_t> v;
        for (uint32_t k = 0; k < ni; k++) v.push_back(c.u64be());
        vk.pub_committed.push_back(v);
    }
    uint32_t nkeys = c.u32be();
    if (nkeys > 16) throw ZkError("bad key count");
    for (uint32_t i = 0; i < nkeys; i++) { G2Affine g = c.g2(), gs = c.g2(); vk.ckeys.push_back({g, gs}); }
    if (c.pos != len) throw ZkError("trailing bytes in verifying key");
}

std::unique_ptr<VerifyingKey> g_vk[3];
std::mutex g_vk_lock;
const char *NAMES[3] = {"chacha20", "aes-128-ctr", "aes-256-ctr"};

static G1Affine g1_neg(const G1Affine &p) { return p.is_inf() ? p : G1Affine{p.x, p.y.neg()}; }
static void fr_to_be(const U256 &v, uint8_t out[32]) { u256_to_be(v, out); }

static bool get_bytes(const Json *j, std::vector<uint8_t> &out) {
    out.clear();
    if (!j || j->t == Json::NUL) return true;
    if (j->t != Json::STR) return false;
    return base64_decode(j->s, out);
}

static bool verify_impl(const uint8_t *data, size_t len) {
    consts_init();
    Json j;
    std::string err;
    if (!json_parse((const char *)data, len, j, err) || j.t != Json::OBJ) { printf("%s\n", err.c_str()); return false; }
    const Json *jc = j.get("cipher"), *jp = j.get("proof"), *js = j.get("publicSignals");
    if (!jc || jc->t != Json::STR) return false;
    int alg = -1;
    for (int i = 0; i < 3; i++) if (jc->s == NAMES[i]) alg = i;
    if (alg < 0) return false;                                           // library.go:178-181: unknown cipher -> false
    std::vector<uint8_t> proof, ct, in;
    if (!get_bytes(jp, proof) || !js || js->t != Json::OBJ) return false;
    if (!get_bytes(js->get("ciphertext"), ct) || !get_bytes(js->get("input"), in)) { printf("failed to parse public signals JSON\n"); return false; }
    const Json *jb = js->get("blocks");
    std::vector<std::vector<uint8_t>> nonces;
    std::vector<uint32_t> counters;
    if (jb && jb->t == Json::ARR)
        for (auto &b : jb->a) {
            std::vector<uint8_t> n;
            uint32_t ctr = 0;
            if (b.t == Json::OBJ) {
                if (!get_bytes(b.get("nonce"), n)) return false;
                const Json *c = b.get("counter");
                if (c && c->t == Json::NUM) {
                    for (char ch : c->raw_num) if (ch < '0' || ch > '9') return false;
                    if (c->num > 4294967295.0) return false;
                    ctr = (uint32_t)strtoull(c->raw_num.c_str(), nullptr, 10);
                } else if (c && c->t != Json::NUL) return false;
            }
            nonces.push_back(n); counters.push_back(ctr);
        }
    const int nblk = alg == 0 ? 2 : 5;
    const size_t chunk = alg == 0 ? 128 : 80;
    if (ct.size() != chunk) { printf("ciphertext must be %zu bytes, not %zu\n", chunk, ct.size()); return false; }      // verifiers.go:41-44
    if (in.size() != chunk) { printf("input must be %zu bytes, not %zu\n", chunk, in.size()); return false; }
    if ((int)nonces.size() != nblk) { printf("blocks array must have %d elements, not %zu\n", nblk, nonces.size()); return false; }
    for (int b = 0; b < nblk; b++)
        if (nonces[b].size() != 12) { printf("block[%d] nonce must be 12 bytes, not %zu\n", b, nonces[b].size()); return false; }
    // public witness (without the constant one), as small integers
    std::vector<uint32_t> pub;
    if (alg == 0) {
        auto bits = [&](uint32_t v) { for (int i = 0; i < 32; i++) pub.push_back((v >> i) & 1); };
        auto le32 = [](const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
        auto be32 = [](const uint8_t *p) { return (uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24); };
        for (int b = 0; b < 2; b++) bits(counters[b]);
        for (int b = 0; b < 2; b++) for (int k = 0; k < 3; k++) bits(le32(nonces[b].data() + 4 * k));
        for (int k = 0; k < 32; k++) bits(be32(in.data() + 4 * k));
        for (int k = 0; k < 32; k++) bits(be32(ct.data() + 4 * k));
    } else {
        for (int b = 0; b < 5; b++) for (int i = 0; i < 12; i++) pub.push_back(nonces[b][i]);
        for (int b = 0; b < 5; b++) pub.push_back(counters[b]);
        for (uint8_t v : in) pub.push_back(v);
        for (uint8_t v : ct) pub.push_back(v);
    }
    std::unique_lock<std::mutex> lk(g_vk_lock);
    if (!g_vk[alg]) {
        // the reference embeds generated/vk.* at build time (verifier/impl/library.go:92-160), so its `Verify` works with no set-up call.
        // The drop-in equivalent: the same files vk.{chacha20,aes128,aes256} in $ZKPROVE_VK_DIR or, when that is unset, in the directory
        // this shared object was loaded from (`make dropin` puts them beside bin/gnark/linux-x86_64-libverify.so).
        static const char *EXT[3] = {"chacha20", "aes128", "aes256"};
        std::vector<std::string> dirs;
        if (const char *dir = getenv("ZKPROVE_VK_DIR")) dirs.push_back(dir);
        else {
            Dl_info di;
            if (dladdr((const void *)&verify_impl, &di) && di.dli_fname) {
                std::string so = di.dli_fname;
                const size_t slash = so.rfind('/');
                dirs.push_back(slash == std::string::npos ? std::string(".") : so.substr(0, slash));
            }
        }
        for (const std::string &dir : dirs) {
            const std::string path = dir + "/vk." + EXT[alg];
            if (FILE *f = fopen(path.c_str(), "rb")) {
                std::vector<uint8_t> buf;
                uint8_t tmp[65536];
                size_t n;
                while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
                fclose(f);
                std::unique_ptr<VerifyingKey> k(new VerifyingKey());
                vk_parse(buf.data(), buf.size(), *k);
                g_vk[alg] = std::move(k);
                break;
            }
        }
    }
    if (!g_vk[alg]) { printf("verifying key for %s is not initialised\n", NAMES[alg]); return false; }
    const VerifyingKey vk = *g_vk[alg];
    lk.unlock();
    // proof: Ar | Bs | Krs | u32 n | commitments | pok
    Cur c{proof.data(), proof.size()};
    G1Affine ar = c.g1();
    G2Affine bs = c.g2();
    G1Affine krs = c.g1();
    uint32_t ncom = c.u32be();
    if (ncom > 16) return false;
    std::vector<G1Affine> coms;
    for (uint32_t i = 0; i < ncom; i++) coms.push_back(c.g1());
    G1Affine pok = c.g1();
    if (c.pos != proof.size()) return false;
    if (ncom != vk.ckeys.size() || ncom != vk.pub_committed.size()) return false;
    if (vk.K.size() != 1 + pub.size() + ncom) return false;
    // kSum = K0 + sum pub_i K_{i+1} + sum challenge_i K_{npub+1+i} + sum commitments
    G1XYZZ ks = G1XYZZ::from_affine(vk.K[0]);
    for (size_t i = 0; i < pub.size(); i++) {
        if (!pub[i]) continue;
        if (pub[i] == 1) { ks.madd(vk.K[i + 1]); continue; }
        uint32_t k[8] = {pub[i], 0, 0, 0, 0, 0, 0, 0};
        ks.add(scalar_mul(vk.K[i + 1], k, 32));
    }
    for (uint32_t i = 0; i < ncom; i++) {
        // challenge = H2F(commitment.Marshal() || public committed values, "bsb22-commitment")  (gnark groth16 verify.go)
        std::vector<uint8_t> ser(64);
        U256 x, y;
        Fp cx = coms[i].x.from_mont(), cy = coms[i].y.from_mont();
        memcpy(x.l, cx.l, 32); memcpy(y.l, cy.l, 32);
        fr_to_be(x, ser.data()); fr_to_be(y, ser.data() + 32);
        for (uint64_t idx : vk.pub_committed[i]) {
            if (idx == 0 || idx > pub.size()) return false;
            uint8_t w[32] = {0};
            uint32_t v = pub[idx - 1];
            w[28] = v >> 24; w[29] = v >> 16; w[30] = v >> 8; w[31] = v;
            ser.insert(ser.end(), w, w + 32);
        }
        U256 ch = hash_to_fr(ser.data(), ser.size(), "bsb22-commitment");
        ks.add(scalar_mul(vk.K[1 + pub.size() + i], ch.l, 254));
        ks.madd(coms[i]);
    }
    if (ncom == 1) {
        // pedersen proof of knowledge: e(C, GSigmaNeg) * e(pok, G) == 1
        if (!pairing_product_is_one({{coms[0], vk.ckeys[0].second}, {pok, vk.ckeys[0].first}})) { printf("pedersen proof of knowledge rejected\n"); return false; }
    } else if (ncom > 1) {
        return false;      // folded multi-commitment verification is not needed by these circuits
    }
    const G1Affine ksa = ks.to_affine();
    // e(Ar, Bs) = e(alpha, beta) e(kSum, gamma) e(Krs, delta)
    const bool ok = pairing_product_is_one({{ar, bs}, {g1_neg(vk.alpha1), vk.beta2}, {g1_neg(ksa), vk.gamma2}, {g1_neg(krs), vk.delta2}});
    if (!ok) printf("pairing doesn't match\n");
    return ok;
}

}  // namespace
}  // namespace zk

extern "C" {

unsigned char Verify(GoSlice params) {
    try {
        return zk::verify_impl((const uint8_t *)params.data, (size_t)params.len) ? 1 : 0;
    } catch (const std::exception &e) {
        printf("%s\n", e.what());                          // library.go:164-169: recover -> false
        return 0;
    }
}
void VFree(void *pointer) { free(pointer); }

// algorithmID 0..2; vk = gnark VerifyingKey.WriteTo bytes.  Returns 1 on success.
unsigned char ZkpInitVerifier(unsigned char algorithmID, GoSlice vk) {
    if (algorithmID >= 3) return 0;
    try {
        zk::consts_init();
        std::unique_ptr<zk::VerifyingKey> k(new zk::VerifyingKey());
        zk::vk_parse((const uint8_t *)vk.data, (size_t)vk.len, *k);
        std::lock_guard<std::mutex> lk(zk::g_vk_lock);
        zk::g_vk[algorithmID] = std::move(k);
        return 1;
    } catch (const std::exception &e) {
        printf("error reading verifying key: %s\n", e.what());
        return 0;
    }
}

}  // extern "C"
