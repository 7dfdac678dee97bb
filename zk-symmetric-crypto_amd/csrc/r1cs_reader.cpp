// gnark v0.14 R1CS binary decoder (see r1cs_reader.h).  Independent of oracle/r1cs_format.py; the
// two decoders are cross-checked in tests/.
#include "r1cs_reader.h"
#include <string.h>
#include <memory>
#include "common.h"

namespace zk {
namespace {

struct Reader {
    const uint8_t *p;
    size_t len, pos = 0;
    void need(size_t n) const {
        if (pos + n > len) throw ZkError("r1cs: truncated input");
    }
    uint64_t u64le() {
        need(8);
        uint64_t v;
        memcpy(&v, p + pos, 8);
        pos += 8;
        return v;
    }
    uint8_t u8() {
        need(1);
        return p[pos++];
    }
};

// ---- ronanh/intcomp streams as stored by gnark: [binpack]? [varbyte]? trailer ----
template <class WordT>
std::vector<uint64_t> intcomp_decode(const std::vector<WordT> &w) {
    constexpr int W = sizeof(WordT) * 8;
    const uint64_t mask = W == 64 ? ~0ull : 0xFFFFFFFFull;
    std::vector<uint64_t> out;
    if (w.size() <= 1) return out;
    size_t body = w.size() - 1, pos = 0;
    uint32_t n0 = (uint32_t)(w[0] & 0xFFFFFFFFu);
    if (n0 != 0 && n0 % (4 * W) == 0) {
        uint64_t n_ints, n_words, first;
        size_t p;
        if (W == 32) {
            if (w.size() < 3) throw ZkError("r1cs: bad binpack header");
            n_ints = w[0]; n_words = w[1]; first = w[2]; p = 3;
        } else {
            n_ints = (uint64_t)w[0] & 0xFFFFFFFFu; n_words = (uint64_t)w[0] >> 32; first = w[1]; p = 2;
        }
        if (n_words > body) throw ZkError("r1cs: binpack overruns stream");
        uint64_t prev = first;
        out.reserve(n_ints);
        while (p < n_words) {
            uint32_t h = (uint32_t)(w[p++] & 0xFFFFFFFFu);
            const int signs[4] = {(int)(h >> 31) & 1, (int)(h >> 23) & 1, (int)(h >> 15) & 1, (int)(h >> 7) & 1};
            const int bits[4] = {(int)(h >> 24) & 127, (int)(h >> 16) & 127, (int)(h >> 8) & 127, (int)h & 127};
            for (int g = 0; g < 4; g++) {
                const int b = bits[g];
                if (b > W) throw ZkError("r1cs: bad bit width");
                if (p + b > n_words) throw ZkError("r1cs: binpack group overruns");
                for (int k = 0; k < W; k++) {
                    uint64_t d = 0;
                    if (b) {
                        size_t bitpos = (size_t)b * k;
                        size_t wi = bitpos / W, sh = bitpos % W;
                        d = (uint64_t)w[p + wi] >> sh;
                        if (sh + b > (size_t)W) d |= (uint64_t)w[p + wi + 1] << (W - sh);
                        if (b < 64) d &= ((1ull << b) - 1);
                    }
                    if (signs[g]) d = (d >> 1) ^ (0 - (d & 1));
                    prev = (prev + d) & mask;
                    out.push_back(prev);
                }
                p += b;
            }
        }
        if (out.size() != n_ints) throw ZkError("r1cs: binpack count mismatch");
        pos = n_words;
    }
    if (pos < body) {
        uint64_t n_ints, n_words;
        size_t q;
        if (W == 32) {
            n_ints = w[pos]; n_words = w[pos + 1]; q = pos + 2;
        } else {
            n_ints = (uint64_t)w[pos] & 0xFFFFFFFFu; n_words = (uint64_t)w[pos] >> 32; q = pos + 1;
        }
        if (pos + n_words > body) throw ZkError("r1cs: varbyte overruns stream");
        std::vector<uint8_t> bs;
        for (size_t i = q; i < pos + n_words; i++)
            for (int k = W / 8 - 1; k >= 0; k--) bs.push_back((uint8_t)((uint64_t)w[i] >> (8 * k)));
        uint64_t prev = 0;
        size_t bp = 0;
        for (uint64_t i = 0; i < n_ints; i++) {
            uint64_t v = 0;
            int shift = 0;
            while (true) {
                if (bp >= bs.size()) throw ZkError("r1cs: varbyte truncated");
                uint8_t c = bs[bp++];
                v |= (uint64_t)(c & 0x7F) << shift;
                shift += 7;
                if (!(c & 0x80)) break;
            }
            prev = (prev + v) & mask;
            out.push_back(prev);
        }
    }
    return out;
}

template <class WordT>
std::vector<uint64_t> read_stream(Reader &r) {
    uint64_t n = r.u64le();
    r.need(n * sizeof(WordT));
    std::vector<WordT> w(n);
    if (n) memcpy(w.data(), r.p + r.pos, n * sizeof(WordT));
    r.pos += n * sizeof(WordT);
    return intcomp_decode<WordT>(w);
}

// ---- minimal CBOR DOM ----
struct Cbor {
    enum T { UINT, NINT, BYTES, TEXT, ARRAY, MAP, TAG, SIMPLE } t = SIMPLE;
    uint64_t u = 0;            // UINT value, NINT raw, TAG number, SIMPLE value
    std::string s;             // BYTES / TEXT
    std::vector<Cbor> a;       // ARRAY items, MAP as k0,v0,k1,v1..., TAG content in a[0]
    const Cbor *get(const char *key) const {
        if (t != MAP) return nullptr;
        for (size_t i = 0; i + 1 < a.size(); i += 2)
            if (a[i].t == TEXT && a[i].s == key) return &a[i + 1];
        return nullptr;
    }
    const Cbor &untag() const { return t == TAG ? a[0] : *this; }
    bool is_null() const { return t == SIMPLE && (u == 22 || u == 23); }
};

Cbor cbor_parse(Reader &r, int depth = 0) {
    if (depth > 64) throw ZkError("r1cs: cbor nesting too deep");
    Cbor c;
    uint8_t ib = r.u8();
    int major = ib >> 5, info = ib & 31;
    uint64_t val = 0;
    bool indef = false;
    if (info < 24) val = info;
    else if (info == 24) val = r.u8();
    else if (info == 25) { r.need(2); val = ((uint64_t)r.p[r.pos] << 8) | r.p[r.pos + 1]; r.pos += 2; }
    else if (info == 26) { r.need(4); for (int i = 0; i < 4; i++) val = (val << 8) | r.p[r.pos + i]; r.pos += 4; }
    else if (info == 27) { r.need(8); for (int i = 0; i < 8; i++) val = (val << 8) | r.p[r.pos + i]; r.pos += 8; }
    else if (info == 31) indef = true;
    else throw ZkError("r1cs: bad cbor additional info");
    switch (major) {
        case 0: c.t = Cbor::UINT; c.u = val; break;
        case 1: c.t = Cbor::NINT; c.u = val; break;
        case 2:
        case 3:
            c.t = major == 2 ? Cbor::BYTES : Cbor::TEXT;
            if (indef) {
                while (true) {
                    r.need(1);
                    if (r.p[r.pos] == 0xFF) { r.pos++; break; }
                    c.s += cbor_parse(r, depth + 1).s;
                }
            } else {
                r.need(val);
                c.s.assign((const char *)r.p + r.pos, val);
                r.pos += val;
            }
            break;
        case 4:
            c.t = Cbor::ARRAY;
            if (indef) {
                while (true) { r.need(1); if (r.p[r.pos] == 0xFF) { r.pos++; break; } c.a.push_back(cbor_parse(r, depth + 1)); }
            } else {
                if (val > r.len) throw ZkError("r1cs: cbor array too long");
                c.a.reserve(val);
                for (uint64_t i = 0; i < val; i++) c.a.push_back(cbor_parse(r, depth + 1));
            }
            break;
        case 5:
            c.t = Cbor::MAP;
            if (indef) {
                while (true) { r.need(1); if (r.p[r.pos] == 0xFF) { r.pos++; break; } c.a.push_back(cbor_parse(r, depth + 1)); c.a.push_back(cbor_parse(r, depth + 1)); }
            } else {
                if (val > r.len) throw ZkError("r1cs: cbor map too long");
                for (uint64_t i = 0; i < val; i++) { c.a.push_back(cbor_parse(r, depth + 1)); c.a.push_back(cbor_parse(r, depth + 1)); }
            }
            break;
        case 6:
            c.t = Cbor::TAG; c.u = val;
            c.a.push_back(cbor_parse(r, depth + 1));
            break;
        default:
            c.t = Cbor::SIMPLE; c.u = info < 24 ? (uint64_t)info : val;
            if (info >= 25 && info <= 27) c.u = 0xF10A7;  // float: not used by gnark bodies
            break;
    }
    return c;
}

std::vector<uint32_t> cbor_u32_list(const Cbor *c) {
    std::vector<uint32_t> out;
    if (!c || c->is_null()) return out;
    const Cbor &v = c->untag();
    if (v.t != Cbor::ARRAY) throw ZkError("r1cs: expected cbor array");
    for (auto &e : v.a) {
        if (e.t != Cbor::UINT) throw ZkError("r1cs: expected uint in array");
        out.push_back((uint32_t)e.u);
    }
    return out;
}

constexpr uint64_t TAG_HINT = 5309735, TAG_R1C = 5309736, TAG_LOOKUP = 5309748;

}  // namespace

void r1cs_parse(const uint8_t *data, size_t len, R1cs &out) {
    Reader r{data, len};
    uint64_t total = r.u64le(), major = r.u64le(), minor = r.u64le();
    (void)r.u64le();
    if (total + 32 != len) throw ZkError("r1cs: length prefix mismatch");
    if (major != 0 || minor != 14) throw ZkError("r1cs: unsupported gnark version (need 0.14.x)");
    uint64_t levels_len = r.u64le(), instr_len = r.u64le(), calldata_len = r.u64le(), body_len = r.u64le();
    size_t end = r.pos + levels_len;
    uint64_t n_levels = r.u64le();
    if (n_levels > len) throw ZkError("r1cs: bad level count");
    out.levels.clear();
    for (uint64_t i = 0; i < n_levels; i++) {
        auto v = read_stream<uint32_t>(r);
        out.levels.emplace_back(v.begin(), v.end());
    }
    if (r.pos != end) throw ZkError("r1cs: levels section size mismatch");
    end = r.pos + instr_len;
    auto bp = read_stream<uint32_t>(r), co = read_stream<uint32_t>(r), wo = read_stream<uint32_t>(r);
    auto sc = read_stream<uint64_t>(r);
    if (r.pos != end) throw ZkError("r1cs: instruction section size mismatch");
    if (bp.size() != co.size() || bp.size() != wo.size() || bp.size() != sc.size()) throw ZkError("r1cs: instruction arrays differ in length");
    out.blueprint_id.assign(bp.begin(), bp.end());
    out.constraint_offset.assign(co.begin(), co.end());
    out.wire_offset.assign(wo.begin(), wo.end());
    out.start_calldata = sc;
    end = r.pos + calldata_len;
    uint64_t n_cd = r.u64le();
    if (n_cd > calldata_len) throw ZkError("r1cs: bad calldata count");
    out.calldata.resize(n_cd);
    for (uint64_t i = 0; i < n_cd; i++) {
        uint64_t v = 0;
        int shift = 0;
        while (true) {
            uint8_t c = r.u8();
            v |= (uint64_t)(c & 0x7F) << shift;
            shift += 7;
            if (!(c & 0x80)) break;
            if (shift > 35) throw ZkError("r1cs: calldata varint too long");
        }
        out.calldata[i] = (uint32_t)v;
    }
    if (r.pos != end) throw ZkError("r1cs: calldata section size mismatch");
    end = r.pos + body_len;
    Cbor body = cbor_parse(r);
    if (r.pos != end) throw ZkError("r1cs: body section size mismatch");
    if (body.t != Cbor::MAP) throw ZkError("r1cs: body is not a map");
    auto req_u = [&](const char *k) -> uint64_t {
        const Cbor *c = body.get(k);
        if (!c || c->t != Cbor::UINT) throw ZkError(std::string("r1cs: missing field ") + k);
        return c->u;
    };
    if (req_u("Type") != 1) throw ZkError("r1cs: not an R1CS system");
    const Cbor *sf = body.get("ScalarField");
    if (!sf || sf->s != "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001") throw ZkError("r1cs: scalar field is not BN254 fr");
    out.nb_constraints = req_u("NbConstraints");
    out.nb_internal = req_u("NbInternalVariables");
    const Cbor *pub = body.get("Public"), *sec = body.get("Secret");
    if (!pub || pub->t != Cbor::ARRAY) throw ZkError("r1cs: missing Public");
    out.nb_public = pub->a.size();
    out.nb_secret = (sec && sec->t == Cbor::ARRAY) ? sec->a.size() : 0;
    const Cbor *bps = body.get("Blueprints");
    if (!bps || bps->t != Cbor::ARRAY) throw ZkError("r1cs: missing Blueprints");
    out.bp_kind.clear();
    out.lookup_entries.clear();
    for (auto &b : bps->a) {
        if (b.t != Cbor::TAG) throw ZkError("r1cs: untagged blueprint");
        out.lookup_entries.emplace_back();
        if (b.u == TAG_HINT) out.bp_kind.push_back(BP_HINT);
        else if (b.u == TAG_R1C) out.bp_kind.push_back(BP_R1C);
        else if (b.u == TAG_LOOKUP) {
            out.bp_kind.push_back(BP_LOOKUP);
            out.lookup_entries.back() = cbor_u32_list(b.a[0].get("EntriesCalldata"));
        } else throw ZkError("r1cs: unsupported blueprint tag " + std::to_string(b.u));
    }
    out.commitments.clear();
    if (const Cbor *ci = body.get("CommitmentInfo")) {
        const Cbor &v = ci->untag();
        if (v.t == Cbor::ARRAY) {
            for (auto &c : v.a) {
                R1csCommitment rc;
                const Cbor *idx = c.get("CommitmentIndex");
                const Cbor *npc = c.get("NbPublicCommitted");
                if (!idx || idx->t != Cbor::UINT) throw ZkError("r1cs: bad commitment info");
                rc.commitment_wire = (uint32_t)idx->u;
                rc.nb_public_committed = npc && npc->t == Cbor::UINT ? (uint32_t)npc->u : 0;
                rc.private_committed = cbor_u32_list(c.get("PrivateCommitted"));
                rc.public_and_commitment_committed = cbor_u32_list(c.get("PublicAndCommitmentCommitted"));
                out.commitments.push_back(std::move(rc));
            }
        }
    }
    out.hint_names.clear();
    if (const Cbor *mh = body.get("MHintsDependencies")) {
        if (mh->t == Cbor::MAP)
            for (size_t i = 0; i + 1 < mh->a.size(); i += 2)
                if (mh->a[i].t == Cbor::UINT && mh->a[i + 1].t == Cbor::TEXT) out.hint_names[(uint32_t)mh->a[i].u] = mh->a[i + 1].s;
    }
    uint64_t n_coeff = r.u64le();
    r.need(n_coeff * 32);
    out.coeffs.resize(n_coeff);
    if (n_coeff) memcpy(out.coeffs.data(), r.p + r.pos, n_coeff * 32);
    r.pos += n_coeff * 32;
    if (r.pos != len) throw ZkError("r1cs: trailing bytes");

    // structural validation
    const size_t ni = out.nb_instructions();
    std::vector<uint8_t> seen(ni, 0);
    for (auto &l : out.levels)
        for (uint32_t i : l) {
            if (i >= ni || seen[i]) throw ZkError("r1cs: level lists do not partition the instructions");
            seen[i] = 1;
        }
    for (size_t i = 0; i < ni; i++) {
        if (!seen[i]) throw ZkError("r1cs: instruction missing from levels");
        if (out.blueprint_id[i] >= out.bp_kind.size()) throw ZkError("r1cs: blueprint id out of range");
        uint64_t s = out.start_calldata[i];
        if (s >= out.calldata.size() || s + out.calldata[s] > out.calldata.size() || out.calldata[s] < 3) throw ZkError("r1cs: calldata bounds");
    }
}

}  // namespace zk
