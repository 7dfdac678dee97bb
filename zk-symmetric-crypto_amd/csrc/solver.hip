// Batched R1CS witness solver kernels (see solver.h).
//
// Program encoding (u32 words, one record per gnark instruction, built once per circuit):
//   OP_R1C      : op, constraint, nL, nR, nO, loc(0 none | 1 L | 2 R | 3 O), solve_wire, solve_coeff, terms (cid,wid)...
//                 -- the single unknown term is removed from the term lists at build time: which wire
//                 an instruction defines is a property of the circuit, not of the witness, so all lanes
//                 (instances) follow the same control flow.
//   OP_NBITS    : op, nout, out_start, linexp
//   OP_COUNT    : op, ntab, ncol, nq, out_start, scratch_slot, table_off, linexp x (nq*ncol)   (table rows: constants in count_tables)
//   OP_RANDOMIZE: op, out_wire
//   OP_BSB22    : op, out_wire
//   OP_LOOKUP   : op, table, nq, first_out_wire, nentries, linexp x nq
//   linexp      : nterms, (cid, wid) x nterms        wid == 0xFFFFFFFF => constant coeffs[cid]
// Data layout: wires[w * B + b], a/b/c[constraint * B + b]  (Fr, Montgomery).
#include "solver.h"
#include <string.h>
#include <algorithm>
#include <array>
#include <map>
#include <string>

namespace zk {

// linear expressions of at least this many terms are evaluated by all slices of a workgroup (eval_terms_coop)
static constexpr int COOP_MIN_TERMS = 16;
// levels of at most this many instructions are "narrow": runs of them are walked by one launch (k_solve_narrow_run)
static constexpr uint32_t NARROW_MAX = 16;
static constexpr uint32_t CHAIN_MAX_WAVES = 8;      // wavefronts of a k_solve_chains workgroup: components of a chain run share them
// lane program (k_solve_lanes): instructions with a linear expression of at least this many terms are shared by the lanes of a wavefront
static constexpr uint32_t LP_WIDE_TERMS = 32;
enum : uint32_t { LP_TASK_LANES = 0, LP_TASK_WIDE = 1, LP_TASK_COUNT = 2 };
// up to this many proofs: a workgroup each.  (Measured through ProveBatch, same box, level launches vs lane program: 8 proofs 7.3 -> 5.7 ms
// per call, 32: 13.9 -> 9.3, 128: 25.4 -> 21.5, 256: 36.1 -> 32.4 -- but a workgroup per proof spends ~5 x the instructions of the
// instance-per-lane kernels, which a full 1 024-batch, bound by instruction issue, cannot afford.)
static constexpr size_t LP_MAX_B = 128;

// ------------------------------------------------------------------------------------- build
namespace {
void push_linexp(std::vector<uint32_t> &out, const uint32_t *&p) {
    uint32_t n = *p++;
    out.push_back(n);
    for (uint32_t k = 0; k < 2 * n; k++) out.push_back(*p++);
}
__global__ void k_coeff_inv(const Fr *c, Fr *ci, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ci[i] = c[i].inv();
}
}  // namespace

void SolverProgram::build(const R1cs &cs, hipStream_t s) {
    n_wires = cs.nb_wires();
    n_inputs = cs.nb_public - 1 + cs.nb_secret;
    n_constraints = cs.nb_constraints;
    n_public = cs.nb_public;
    log_domain = cs.log_domain();
    const size_t ni = cs.nb_instructions();
    std::vector<uint8_t> solved(n_wires, 0);
    for (size_t i = 0; i < cs.nb_public + cs.nb_secret; i++) solved[i] = 1;
    // lookup tables
    std::vector<int> table_of_bp(cs.bp_kind.size(), -1);
    std::vector<Fr> tables;
    size_t nent_max = 0;
    int ntables = 0;
    for (size_t bp = 0; bp < cs.bp_kind.size(); bp++) {
        if (cs.bp_kind[bp] != BP_LOOKUP) continue;
        const auto &e = cs.lookup_entries[bp];
        std::vector<Fr> vals;
        for (size_t p = 0; p < e.size();) {
            uint32_t nt = e[p++];
            if (nt != 1 || p + 2 > e.size() || e[p + 1] != WIRE_CONST || e[p] >= cs.coeffs.size())
                throw ZkError("lookup blueprint with non-constant entries is not supported");
            Fr v;
            memcpy(v.l, cs.coeffs[e[p]].data(), 32);
            vals.push_back(v);
            p += 2;
        }
        if (nent_max && vals.size() != nent_max) throw ZkError("lookup tables of differing sizes");
        nent_max = vals.size();
        table_of_bp[bp] = ntables++;
        tables.insert(tables.end(), vals.begin(), vals.end());
    }
    has_commit = !cs.commitments.empty();
    if (cs.commitments.size() > 1) throw ZkError("more than one commitment is not supported");
    if (has_commit) {
        const auto &ci = cs.commitments[0];
        if (ci.nb_public_committed || !ci.public_and_commitment_committed.empty()) throw ZkError("public committed wires are not supported");
        commit_wire = ci.commitment_wire;
        private_committed = ci.private_committed;
    }
    std::vector<uint32_t> prg, off(ni);
    std::vector<Fr> count_tab;
    std::vector<uint32_t> defer;           // deferred divisions: (level-local order) cons, loc, wire, coeff
    std::vector<uint32_t> defer_level(ni, 0);
    std::vector<uint8_t> is_commit_instr(ni, 0), is_div_instr(ni, 0), is_wide_instr(ni, 0);      // wide: see k_solve_level
    std::vector<uint8_t> covered(n_constraints, 0);      // constraint rows of a / b / c that some R1C instruction writes (solver_begin)
    n_count_rows = 0;
    n_count_slots = 0;
    for (size_t i = 0; i < ni; i++) {
        off[i] = (uint32_t)prg.size();
        const uint32_t *cd = cs.calldata.data() + cs.start_calldata[i];
        const uint32_t size = cd[0];
        const int kind = cs.bp_kind[cs.blueprint_id[i]];
        auto chk_wire = [&](uint32_t w) { if (w != WIRE_CONST && w >= n_wires) throw ZkError("wire id out of range"); };
        if (kind == BP_R1C) {
            const uint32_t n[3] = {cd[1], cd[2], cd[3]};
            if (4 + 2 * (uint64_t)(n[0] + n[1] + n[2]) != size) throw ZkError("r1c calldata size mismatch");
            size_t hdr = prg.size();
            prg.insert(prg.end(), {OP_R1C, cs.constraint_offset[i], 0u, 0u, 0u, 0u, 0u, 0u});
            const uint32_t *p = cd + 4;
            uint32_t loc = 0;
            for (int which = 0; which < 3; which++) {
                uint32_t kept = 0;
                for (uint32_t k = 0; k < n[which]; k++, p += 2) {
                    chk_wire(p[1]);
                    if (p[0] >= cs.coeffs.size()) throw ZkError("coefficient id out of range");
                    if (p[1] != WIRE_CONST && !solved[p[1]]) {
                        if (loc) throw ZkError("instruction with more than one unknown wire");
                        loc = which + 1;
                        prg[hdr + 6] = p[1]; prg[hdr + 7] = p[0];
                        solved[p[1]] = 1;
                        continue;
                    }
                    prg.push_back(p[0]); prg.push_back(p[1]);
                    kept++;
                }
                prg[hdr + 2 + which] = kept;
                if (kept >= (uint32_t)COOP_MIN_TERMS) is_wide_instr[i] = 1;
            }
            prg[hdr + 5] = loc;
            if (loc == 1 || loc == 2) is_div_instr[i] = 1;
            if (cs.constraint_offset[i] >= n_constraints) throw ZkError("constraint index out of range");
            covered[cs.constraint_offset[i]] = 1;
        } else if (kind == BP_HINT) {
            const uint32_t hid = cd[1], nin = cd[2];
            const uint32_t out_start = cd[size - 2], out_end = cd[size - 1];
            if (out_end < out_start || out_end > n_wires) throw ZkError("hint outputs out of range");
            const uint32_t *p = cd + 3;
            if (hid == HINT_NBITS) {
                if (nin != 1) throw ZkError("nBits hint with != 1 input");
                prg.insert(prg.end(), {OP_NBITS, out_end - out_start, out_start});
                if (p[0] >= (uint32_t)COOP_MIN_TERMS) is_wide_instr[i] = 1;
                push_linexp(prg, p);
            } else if (hid == HINT_COUNT) {
                // inputs: [nTable, nCol, table rows..., query rows...]; first two are constants
                std::vector<uint32_t> tmp;
                const uint32_t *q = p;
                auto const_val = [&](const uint32_t *&pp) -> uint64_t {
                    uint32_t nt = *pp++;
                    if (nt != 1 || pp[1] != WIRE_CONST) throw ZkError("countHint header is not constant");
                    // canonical value of the Montgomery coefficient is needed: small constants only
                    uint32_t cid = pp[0];
                    pp += 2;
                    // coefficient ids 0,1,2 are the literals 0,1,2; others: decode via from_mont on host is avoided
                    // by scanning small integers
                    if (cid <= 2) return cid;
                    return 0x100000000ull | cid;   // resolved below
                };
                uint64_t ntab_c = const_val(q), ncol_c = const_val(q);
                // resolve coefficient ids to small integers by brute force compare with i*R mod r computed incrementally on host
                auto resolve = [&](uint64_t v) -> uint32_t {
                    if (!(v >> 32)) return (uint32_t)v;
                    uint32_t cid = (uint32_t)v;
                    // host Montgomery "one" = coeffs[1]; add repeatedly
                    const uint32_t *one = cs.coeffs[1].data();
                    uint32_t acc[8] = {0};
                    static const uint32_t M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
                    for (uint32_t k = 1; k <= 65536; k++) {
                        uint64_t c = 0;
                        for (int j = 0; j < 8; j++) { uint64_t x = (uint64_t)acc[j] + one[j] + c; acc[j] = (uint32_t)x; c = x >> 32; }
                        bool ge = true;
                        for (int j = 7; j >= 0; j--) { if (acc[j] > M[j]) break; if (acc[j] < M[j]) { ge = false; break; } }
                        if (ge) { uint64_t br = 0; for (int j = 0; j < 8; j++) { uint64_t x = (uint64_t)acc[j] - M[j] - br; acc[j] = (uint32_t)x; br = (x >> 63) & 1; } }
                        if (memcmp(acc, cs.coeffs[cid].data(), 32) == 0) return k;
                    }
                    throw ZkError("countHint header constant too large");
                };
                const uint32_t ntab = resolve(ntab_c), ncol = resolve(ncol_c);
                if (ncol == 0 || nin < 2 + ntab * ncol || (nin - 2 - ntab * ncol) % ncol) throw ZkError("countHint input shape");
                const uint32_t nq = (nin - 2 - ntab * ncol) / ncol;
                if (out_end - out_start != ntab) throw ZkError("countHint output count");
                // table rows are constants: resolve them now so that the kernel compares a query row in O(1)
                const uint32_t table_off = (uint32_t)count_tab.size();
                for (uint32_t k = 0; k < ntab * ncol; k++) {
                    if (q[0] != 1 || q[2] != WIRE_CONST || q[1] >= cs.coeffs.size()) throw ZkError("countHint table rows must be constants");
                    Fr v;
                    memcpy(v.l, cs.coeffs[q[1]].data(), 32);
                    count_tab.push_back(v);
                    q += 3;
                }
                const size_t hdr = prg.size();
                prg.insert(prg.end(), {OP_COUNT, ntab, ncol, nq, out_start, (uint32_t)n_count_slots, table_off, 0u});
                is_wide_instr[i] = 1;
                n_count_slots++;
                // word 7: offset (from the first query) of a table with the start of every query row, so that the slices of a
                // workgroup can each take every SL-th query although the rows are variable-length linear expressions
                std::vector<uint32_t> row_start(nq);
                const size_t q0 = prg.size();
                for (uint32_t j = 0; j < nq; j++) {
                    row_start[j] = (uint32_t)(prg.size() - q0);
                    for (uint32_t col = 0; col < ncol; col++) push_linexp(prg, q);
                }
                prg[hdr + 7] = (uint32_t)(prg.size() - q0);
                prg.insert(prg.end(), row_start.begin(), row_start.end());
                if (ntab > n_count_rows) n_count_rows = ntab;
            } else if (hid == HINT_RANDOMIZE) {
                if (out_end - out_start != 1) throw ZkError("Randomize hint with != 1 output");
                prg.insert(prg.end(), {OP_RANDOMIZE, out_start});
            } else if (hid == HINT_BSB22) {
                if (!has_commit || out_start != commit_wire) throw ZkError("bsb22 hint does not match commitment info");
                // inputs: [commitment index, committed wires...] must be the PrivateCommitted list, coefficient one
                if (nin != private_committed.size() + 1) throw ZkError("bsb22 hint input count");
                const uint32_t *q = p;
                q += 1 + 2 * q[0];
                for (size_t k = 0; k < private_committed.size(); k++) {
                    if (q[0] != 1 || q[1] != 1 || q[2] != private_committed[k]) throw ZkError("bsb22 hint inputs differ from PrivateCommitted");
                    q += 3;
                }
                prg.insert(prg.end(), {OP_BSB22, out_start});
                is_commit_instr[i] = 1;
            } else {
                throw ZkError("unsupported hint id " + std::to_string(hid));
            }
            for (uint32_t w = out_start; w < out_end; w++) solved[w] = 1;
        } else {   // lookup
            const uint32_t nent = cd[1], nq = cd[2];
            int tb = table_of_bp[cs.blueprint_id[i]];
            if (tb < 0 || nent != nent_max) throw ZkError("lookup instruction without table");
            if ((size_t)cs.wire_offset[i] + nq > n_wires) throw ZkError("lookup outputs out of range");
            prg.insert(prg.end(), {OP_LOOKUP, (uint32_t)tb, nq, cs.wire_offset[i], nent});
            const uint32_t *p = cd + 3;
            for (uint32_t q = 0; q < nq; q++) push_linexp(prg, p);
            for (uint32_t q = 0; q < nq; q++) solved[cs.wire_offset[i] + q] = 1;
        }
    }
    for (size_t w = 0; w < n_wires; w++)
        if (!solved[w]) throw ZkError("circuit leaves wire " + std::to_string(w) + " undefined");
    constraints_covered = true;
    for (uint8_t c : covered) constraints_covered = constraints_covered && c;
    // levels
    std::vector<uint32_t> li;
    levels.clear();
    for (auto &l : cs.levels) {
        SolverLevel lv;
        lv.first = (uint32_t)li.size();
        lv.count = (uint32_t)l.size();
        lv.div_first = (uint32_t)(defer.size() / 4);
        // wide instructions (a linear expression of >= COOP_MIN_TERMS terms, a countHint) first: k_solve_level gives them a workgroup
        // each and packs the others four to a workgroup
        std::vector<uint32_t> ordered;
        for (int pass = 0; pass < 2; pass++)
            for (uint32_t i : l)
                if ((int)is_wide_instr[i] == 1 - pass) ordered.push_back(i);
        // ... and, within the two parts, by SHAPE (opcode, term counts, coefficient ids -- everything but the wire / constraint
        // numbers): the circuits repeat a few gadgets thousands of times, so instructions of one shape end up side by side.  The level
        // kernels do not care (one instruction per workgroup row); the persistent kernel, whose LANES are instructions, then runs
        // wavefronts of same-shaped instructions and its branches on the program words stay (nearly) uniform.
        {
            auto shape = [&](uint32_t i) {
                const uint32_t *rec = prg.data() + off[i];
                const size_t end = i + 1 < ni ? off[i + 1] : prg.size();
                uint64_t h = 1469598103934665603ull;
                auto mix = [&](uint32_t v) { h = (h ^ v) * 1099511628211ull; };
                const size_t n = end - off[i];
                mix(rec[0]);
                if (rec[0] == OP_R1C) {
                    mix(rec[2]); mix(rec[3]); mix(rec[4]); mix(rec[5]); mix(rec[7]);
                    for (size_t k = 8; k < n; k += 2) mix(rec[k]);                           // coefficient ids of the terms
                } else if (rec[0] == OP_NBITS) {
                    mix(rec[1]); mix(rec[3]);
                    for (size_t k = 4; k < n; k += 2) mix(rec[k]);
                } else if (rec[0] == OP_LOOKUP) {
                    mix(rec[1]); mix(rec[2]);
                } else mix((uint32_t)n);
                return h;
            };
            std::vector<std::pair<uint64_t, uint32_t>> keyed;
            size_t nwide_l = 0;
            for (uint32_t i : l) nwide_l += is_wide_instr[i];
            for (int part = 0; part < 2; part++) {
                const size_t lo = part == 0 ? 0 : nwide_l, hi = part == 0 ? nwide_l : ordered.size();
                keyed.clear();
                for (size_t q = lo; q < hi; q++) keyed.push_back({shape(ordered[q]), ordered[q]});
                std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint64_t, uint32_t> &x, const std::pair<uint64_t, uint32_t> &y) { return x.first < y.first; });
                for (size_t q = lo; q < hi; q++) ordered[q] = keyed[q - lo].second;
            }
        }
        lv.wide = 0;
        for (uint32_t i : l) lv.wide += is_wide_instr[i];
        // countHints in front of the other wide instructions: solver_run_levels gives them a launch of their own (k_solve_counts)
        std::stable_partition(ordered.begin(), ordered.begin() + lv.wide, [&](uint32_t i) { return prg[off[i]] == OP_COUNT; });
        lv.n_count = 0;
        for (uint32_t i : l) lv.n_count += prg[off[i]] == OP_COUNT ? 1u : 0u;
        for (uint32_t i : ordered) {
            li.push_back(i);
            if (is_commit_instr[i]) lv.has_commit = true;
            if (is_div_instr[i]) {
                const uint32_t *rec = prg.data() + off[i];
                defer.insert(defer.end(), {rec[1], rec[5], rec[6], rec[7]});
            }
        }
        lv.div_count = (uint32_t)(defer.size() / 4) - lv.div_first;
        lv.narrow = lv.count <= NARROW_MAX && !lv.has_commit;
        for (uint32_t i : ordered)
            if (prg[off[i]] == OP_COUNT) lv.narrow = false;      // a countHint wants the whole device
        levels.push_back(lv);
    }
    prg.resize(prg.size() + 16, 0);          // k_solve_narrow_run's prefetcher reads 16-word strides of a record whatever its opcode: the last record must not end the buffer
    prog.upload(prg.data(), prg.size(), s);
    instr_off.upload(off.data(), off.size(), s);
    level_instr.upload(li.data(), li.size(), s);
    {
        std::vector<uint32_t> desc;
        size_t max_counts = 0;
        for (auto &lv : levels) desc.insert(desc.end(), {lv.first, lv.count, lv.wide, lv.div_first, lv.div_count});
        (void)max_counts;
        level_desc.upload(desc.data(), desc.size(), s);
    }
    // ---- chain runs (k_solve_chains, below): maximal runs of narrow levels, every instruction a multiplication row
    {
        chain_runs.clear();
        std::vector<uint32_t> c_rec, c_first;                       // 16 words per instruction (layout: k_solve_chains)
        std::vector<Fr> c_const(1, Fr::zero());                     // folded constants; entry 0 = zero
        std::map<std::array<uint32_t, 8>, uint32_t> const_at;
        auto intern = [&](const Fr &v) -> uint32_t {
            if (v.is_zero()) return 0;
            std::array<uint32_t, 8> key;
            memcpy(key.data(), v.l, 32);
            auto it = const_at.find(key);
            if (it == const_at.end()) { it = const_at.emplace(key, (uint32_t)c_const.size()).first; c_const.push_back(v); }
            return it->second;
        };
        // Terms over wire 0 (the circuit's constant ONE: k_assign_inputs writes Fr::one() there for every proof) and constant terms are the
        // same number for every proof: folded per side here, with the values the device code gives them (term_apply).  Field additions are
        // exact on canonical values, so the order of a side's terms does not show in the result.
        auto emit = [&](uint32_t i) {
            const uint32_t *rec = prg.data() + off[i];
            Fr K[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
            uint32_t dyn[3] = {0, 0, 0};
            std::vector<uint32_t> terms;
            const uint32_t *q = rec + 8;
            for (int side = 0; side < 3; side++)
                for (uint32_t k = 0; k < rec[2 + side]; k++, q += 2) {
                    const uint32_t cid = q[0], wid = q[1];
                    if (wid != WIRE_CONST && wid != 0) { dyn[side]++; terms.push_back(cid); terms.push_back(wid); continue; }
                    Fr cv;
                    memcpy(cv.l, cs.coeffs[cid].data(), 32);
                    const Fr one = Fr::one();
                    Fr v = cv;
                    if (wid == 0)
                        switch (cid) {
                            case 0: v = Fr::zero(); break;
                            case 1: v = one; break;
                            case 2: v = one.dbl(); break;
                            case 3: v = one.neg(); break;
                            case 4: v = one.dbl().neg(); break;
                            default: break;
                        }
                    K[side] = K[side] + v;
                }
            uint32_t w16[16] = {0};
            if (terms.size() > 8) {
                w16[4] = 0xFFFFFFFFu; w16[5] = off[i];              // long rows (the checks): the ordinary evaluation of the program's record
            } else {
                w16[0] = rec[1]; w16[1] = rec[5]; w16[2] = rec[6]; w16[3] = rec[7];
                w16[4] = dyn[0] | dyn[1] << 8 | dyn[2] << 16;
                for (int side = 0; side < 3; side++) w16[5 + side] = intern(K[side]);
                for (size_t k = 0; k < terms.size(); k++) w16[8 + k] = terms[k];
            }
            c_rec.insert(c_rec.end(), w16, w16 + 16);
        };
        for (size_t l = 0; l < levels.size();) {
            if (!levels[l].narrow) { l++; continue; }
            size_t e = l;
            while (e < levels.size() && levels[e].narrow) e++;
            const size_t l0 = l;
            l = e;
            if (e - l0 < 2) continue;
            bool ok = true;
            std::vector<uint32_t> ins;                           // the run's instructions in level order
            for (size_t q = l0; q < e && ok; q++) {
                if (levels[q].div_count) ok = false;
                for (uint32_t k = 0; k < levels[q].count && ok; k++) {
                    const uint32_t i = li[levels[q].first + k];
                    const uint32_t *rec = prg.data() + off[i];
                    if (rec[0] != OP_R1C || (rec[5] != 0 && rec[5] != 3) || rec[2] + rec[3] + rec[4] > 96 || (rec[5] == 3 && rec[6] == 0)) ok = false;      // (wire 0 stays a constant)
                    ins.push_back(i);
                }
            }
            if (!ok || ins.empty()) continue;
            // components: an instruction joins the producers (inside the run) of the wires it reads
            std::vector<uint32_t> parent(ins.size());
            for (uint32_t k = 0; k < ins.size(); k++) parent[k] = k;
            auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
            std::map<uint32_t, uint32_t> producer;               // wire -> position in ins
            for (uint32_t k = 0; k < ins.size(); k++) {
                const uint32_t *rec = prg.data() + off[ins[k]];
                const uint32_t nt = rec[2] + rec[3] + rec[4];
                for (uint32_t t2 = 0; t2 < nt; t2++) {
                    auto it = producer.find(rec[8 + 2 * t2 + 1]);
                    if (it != producer.end()) { const uint32_t ra = find(k), rb = find(it->second); if (ra != rb) parent[ra] = rb; }
                }
                if (rec[5] == 3) producer[rec[6]] = k;
            }
            std::map<uint32_t, std::vector<uint32_t>> comp;      // root -> members (level order, as ins is)
            for (uint32_t k = 0; k < ins.size(); k++) comp[find(k)].push_back(k);
            std::vector<const std::vector<uint32_t> *> by_size;
            for (auto &kv : comp) by_size.push_back(&kv.second);
            std::stable_sort(by_size.begin(), by_size.end(), [](const std::vector<uint32_t> *x, const std::vector<uint32_t> *y) { return x->size() > y->size(); });
            const uint32_t W = (uint32_t)std::min<size_t>(CHAIN_MAX_WAVES, by_size.size());
            std::vector<std::vector<uint32_t>> wave(W);
            for (const auto *m : by_size) {                      // largest first, each to the least loaded wavefront
                uint32_t best = 0;
                for (uint32_t w = 1; w < W; w++) if (wave[w].size() < wave[best].size()) best = w;
                wave[best].insert(wave[best].end(), m->begin(), m->end());
            }
            ChainRun cr;
            cr.l0 = (uint32_t)l0; cr.l1 = (uint32_t)e; cr.waves = W; cr.first = (uint32_t)c_first.size();
            for (uint32_t w = 0; w < W; w++) {
                c_first.push_back((uint32_t)(c_rec.size() / 16));
                for (uint32_t k : wave[w]) emit(ins[k]);
            }
            c_first.push_back((uint32_t)(c_rec.size() / 16));
            chain_runs.push_back(cr);
        }
        c_rec.resize(c_rec.size() + 16, 0);
        if (c_first.empty()) c_first.push_back(0);
        chain_recs.upload(c_rec.data(), c_rec.size(), s);
        chain_wave_first.upload(c_first.data(), c_first.size(), s);
        chain_consts.alloc(c_const.size());
        ZK_HIP(hipMemcpyAsync(chain_consts.p, c_const.data(), c_const.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
        ZK_HIP(hipStreamSynchronize(s));                            // (c_const is a local)
    }
    // ---- lane program (k_solve_lanes): the same instructions, regrouped per level by SHAPE -- everything of an instruction but its
    // wire / constraint numbers.  A shape is stored once (opcode, term counts, (coefficient id, is-constant) per term); the
    // instructions of one shape in one level become columns of numbers (field f of instruction j at base + f * count + j), so that
    // 64 lanes read 64 instructions' numbers with one coalesced load per field and run the shape's code with uniform control flow.
    // Every term owns a field (a constant term's field names wire 0, loaded and ignored): field index = position in the term list.
    // Instructions with a linear expression of >= LP_WIDE_TERMS terms stay in the encoding above and are shared by the lanes of a
    // wavefront (a task of their own); countHints are worked on by the whole workgroup.
    {
        std::vector<uint32_t> shapes, tasks, soa, lvl;
        std::map<std::vector<uint32_t>, uint32_t> shape_at;
        // A long CHECK (an R1C instruction that defines no wire: the 130-term add32 recompositions) is the slowest task of its level
        // and nobody waits for its result: it moves to the NEXT level, where it runs beside that level's own tasks (ChaCha20: beside
        // the eight nBits hints, on the eight wavefronts that had nothing to do) -- its a / b / c rows and its verdict arrive a level
        // later, before the kernel ends.  (Not across a commitment level: the caller cuts the level range there.)
        std::vector<std::array<uint32_t, 4>> deferred;
        for (size_t lidx = 0; lidx < levels.size(); lidx++) {
            const auto &lv = levels[lidx];
            const bool may_defer = lidx + 1 < levels.size() && !levels[lidx + 1].has_commit && !lv.has_commit;
            std::vector<std::array<uint32_t, 4>> wide_t, count_t, lane_t, incoming;
            incoming.swap(deferred);
            std::map<uint32_t, std::vector<std::vector<uint32_t>>> groups;      // shape offset -> field rows
            std::vector<uint32_t> group_order;
            for (uint32_t q = 0; q < lv.count; q++) {
                const uint32_t i = li[lv.first + q];
                const uint32_t *rec = prg.data() + off[i];
                const uint32_t op = rec[0];
                std::vector<uint32_t> shp, fld;
                bool is_wide = false;
                auto terms = [&](const uint32_t *t, uint32_t n) {
                    if (n >= LP_WIDE_TERMS) is_wide = true;
                    for (uint32_t k = 0; k < n; k++) {
                        const bool c = t[2 * k + 1] == WIRE_CONST;
                        shp.push_back(t[2 * k]); shp.push_back(c ? 1u : 0u);
                        fld.push_back(c ? 0u : t[2 * k + 1]);
                    }
                };
                if (op == OP_COUNT) { count_t.push_back({LP_TASK_COUNT, off[i], i, 0u}); continue; }
                if (op == OP_R1C) {
                    shp = {OP_R1C, rec[2], rec[3], rec[4], rec[5], rec[7]};
                    fld = {rec[1], rec[6]};
                    const uint32_t *t = rec + 8;
                    terms(t, rec[2]); t += 2 * rec[2];
                    terms(t, rec[3]); t += 2 * rec[3];
                    terms(t, rec[4]);
                } else if (op == OP_NBITS) {
                    shp = {OP_NBITS, rec[1], rec[3]};
                    fld = {rec[2]};
                    terms(rec + 4, rec[3]);
                } else if (op == OP_LOOKUP) {
                    shp = {OP_LOOKUP, rec[1], rec[2], rec[4]};
                    fld = {rec[3], i};                               // the instruction number: detail of a range error
                    const uint32_t *t = rec + 5;
                    for (uint32_t k = 0; k < rec[2]; k++) {
                        const uint32_t n = *t++;
                        shp.push_back(n);
                        terms(t, n);
                        t += 2 * n;
                    }
                } else {       // OP_RANDOMIZE, OP_BSB22
                    shp = {op};
                    fld = {rec[1]};
                }
                if (is_wide) {
                    if (op == OP_R1C && rec[5] == 0 && may_defer) deferred.push_back({LP_TASK_WIDE, off[i], i, 0u});
                    else wide_t.push_back({LP_TASK_WIDE, off[i], i, 0u});
                    continue;
                }
                auto it = shape_at.find(shp);
                if (it == shape_at.end()) {
                    it = shape_at.emplace(shp, (uint32_t)shapes.size()).first;
                    shapes.insert(shapes.end(), shp.begin(), shp.end());
                }
                if (!groups.count(it->second)) group_order.push_back(it->second);
                groups[it->second].push_back(std::move(fld));
            }
            for (uint32_t so : group_order) {
                const auto &rows = groups[so];
                const uint32_t gc = (uint32_t)rows.size(), nf = (uint32_t)rows[0].size();
                const uint32_t base = (uint32_t)soa.size();
                for (uint32_t f = 0; f < nf; f++)
                    for (uint32_t j = 0; j < gc; j++) soa.push_back(rows[j][f]);
                for (uint32_t j0 = 0; j0 < gc; j0 += 64) lane_t.push_back({LP_TASK_LANES, so, base + j0, (gc - j0 < 64 ? gc - j0 : 64u) | (gc << 7)});
            }
            lvl.insert(lvl.end(), {(uint32_t)(tasks.size() / 4), (uint32_t)(wide_t.size() + lane_t.size() + incoming.size()), (uint32_t)count_t.size()});
            for (const auto *v : {&wide_t, &lane_t, &incoming, &count_t})          // long instructions first, the previous level's checks behind the level's own tasks, countHints last
                for (const auto &t : *v) tasks.insert(tasks.end(), t.begin(), t.end());
        }
        if (shapes.empty()) shapes.push_back(0);
        if (tasks.empty()) tasks.push_back(0);
        if (soa.empty()) soa.push_back(0);
        lp_shapes.upload(shapes.data(), shapes.size(), s);
        lp_tasks.upload(tasks.data(), tasks.size(), s);
        lp_soa.upload(soa.data(), soa.size(), s);
        lp_level.upload(lvl.data(), lvl.size(), s);
    }
    coeffs.alloc(cs.coeffs.size());
    ZK_HIP(hipMemcpyAsync(coeffs.p, cs.coeffs.data(), cs.coeffs.size() * 32, hipMemcpyHostToDevice, s));
    coeff_inv.alloc(cs.coeffs.size());
    hipLaunchKernelGGL(k_coeff_inv, dim3(ceil_div(cs.coeffs.size(), 64)), dim3(64), 0, s, coeffs.p, coeff_inv.p, cs.coeffs.size());
    count_tables.alloc(count_tab.size() ? count_tab.size() : 1);
    if (!count_tab.empty()) ZK_HIP(hipMemcpyAsync(count_tables.p, count_tab.data(), count_tab.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
    div_list.upload(defer.data(), defer.size() ? defer.size() : 0, s);
    lookup_tables.alloc(tables.size() ? tables.size() : 1);
    if (!tables.empty()) ZK_HIP(hipMemcpyAsync(lookup_tables.p, tables.data(), tables.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
    ZK_HIP(hipStreamSynchronize(s));
}

void SolverBuffers::reserve(const SolverProgram &p, size_t B_, size_t c_B) {
    B = B_;
    const size_t N = (size_t)1 << p.log_domain;
    wires.ensure(p.n_wires * B);
    a.ensure(N * B); b.ensure(N * B); c.ensure(N * (c_B < B ? c_B : B));
    inputs.ensure(p.n_inputs * B);
    raw.ensure(SOLVER_RAW_STRIDE * B);
    randomize.ensure(B); challenge.ensure(B); err.ensure(B);
    count_scratch.ensure((p.n_count_rows ? p.n_count_rows : 1) * (p.n_count_slots ? p.n_count_slots : 1) * B);
    count_ticket.ensure((p.n_count_slots ? p.n_count_slots : 1) * (B / 8 + 1));      // tiles of at least 8 instances
    sync.ensure(2);
}

// ----------------------------------------------------------------------------------- kernels
// The program, the coefficient tables and the lookup tables are written before a launch and never inside one.  Reached through a plain
// pointer in a by-value struct the compiler cannot know that: every uniform read of a program word became a VECTOR load (~1 us per
// dependent round trip: offset -> header -> terms -> coefficient, in front of every instruction), and inside loops that store it must.
// The CONSTANT address space says it: uniform addresses get scalar loads (s_load_dwordx2..x16 through the scalar cache), divergent ones
// (the lane program's per-lane coefficients, table entries) ordinary loads.
typedef const uint32_t __attribute__((address_space(4))) *ProgPtr;
typedef const Fr __attribute__((address_space(4))) *FrTab;
__device__ __forceinline__ Fr tab_load(FrTab p) {
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = p->l[j];
    return r;
}
struct SolveCtx {
    ProgPtr prog;
    FrTab coeffs, coeff_inv, tables, count_tables;
    Fr *wires, *a, *b, *c;
    const Fr *randomize, *challenge;
    uint32_t *err, *count_scratch;
    size_t B, count_rows;
    uint32_t *count_ticket;       // per (countHint slot, instance tile): workgroups of k_solve_counts that have finished their share of the queries
    unsigned long long *dbg;      // experiments build: per (level, wavefront) time stamps of k_solve_lanes (ZKPROVE_SOLVER_DBG=1), else null
};

// Almost every wire of these circuits is a bit, so coefficient * wire and <L,w> * <R,w> are mostly products with
// 0 or 1: test for that before paying a 254-bit Montgomery multiplication (all lanes agree in practice, the
// add32 / FromBinary constraints with ~100 power-of-two terms otherwise cost ~100 sequential multiplications).
__device__ __forceinline__ Fr mul_fast(const Fr &a, const Fr &b) {
    if (a.is_zero() || b.is_zero()) return Fr::zero();
    const Fr one = Fr::one();
    if (a == one) return b;
    if (b == one) return a;
    const Fr mone = one.neg();          // -1 is the third common wire value (a - b of two bits, XOR intermediates)
    if (a == mone) return b.neg();
    if (b == mone) return a.neg();
    return a * b;
}
__device__ __forceinline__ Fr term_val(const SolveCtx &cx, uint32_t cid, uint32_t wid, size_t b) {
    if (wid == WIRE_CONST) return tab_load(cx.coeffs + cid);
    const Fr w = cx.wires[(size_t)wid * cx.B + b];
    switch (cid) {
        case 0: return Fr::zero();
        case 1: return w;
        case 2: return w.dbl();
        case 3: return w.neg();
        case 4: return w.dbl().neg();
        default: return mul_fast(w, tab_load(cx.coeffs + cid));
    }
}
// p is wave-uniform (one instruction per workgroup row): the (cid, wid) words come from scalar loads, and the
// wire loads of four terms are issued together so that a 100-term linear expression costs ~25 memory round
// trips instead of 100 (the long add32 constraints used to set every level's latency).
__device__ __forceinline__ Fr term_apply(const SolveCtx &cx, uint32_t cid, uint32_t wid, const Fr &w) {
    if (wid == WIRE_CONST) return tab_load(cx.coeffs + cid);
    switch (cid) {
        case 0: return Fr::zero();
        case 1: return w;
        case 2: return w.dbl();
        case 3: return w.neg();
        case 4: return w.dbl().neg();
        default: return mul_fast(w, tab_load(cx.coeffs + cid));
    }
}
// the same with the coefficient's value already in registers
__device__ __forceinline__ Fr term_apply_pre(uint32_t cid, uint32_t wid, const Fr &w, const Fr &cval) {
    if (wid == WIRE_CONST) return cval;
    switch (cid) {
        case 0: return Fr::zero();
        case 1: return w;
        case 2: return w.dbl();
        case 3: return w.neg();
        case 4: return w.dbl().neg();
        default: return mul_fast(w, cval);
    }
}
__device__ __forceinline__ Fr eval_terms(const SolveCtx &cx, ProgPtr &p, uint32_t n, size_t b) {
    Fr acc = Fr::zero();
    uint32_t k = 0;
    for (; k + 8 <= n; k += 8, p += 16) {
        // unconditional loads (constants read wire 0, a valid address): a conditional load would be branched around
        // and waited for one by one
        Fr v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t w = p[2 * q + 1];
            v[q] = cx.wires[(size_t)(w == WIRE_CONST ? 0u : w) * cx.B + b];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) acc = acc + term_apply(cx, p[2 * q], p[2 * q + 1], v[q]);
    }
    for (; k + 2 <= n; k += 2, p += 4) {
        const uint32_t w0 = p[1], w1 = p[3];
        const Fr v0 = cx.wires[(size_t)(w0 == WIRE_CONST ? 0u : w0) * cx.B + b];
        const Fr v1 = cx.wires[(size_t)(w1 == WIRE_CONST ? 0u : w1) * cx.B + b];
        acc = acc + term_apply(cx, p[0], w0, v0);
        acc = acc + term_apply(cx, p[2], w1, v1);
    }
    for (; k < n; k++, p += 2) acc = acc + term_val(cx, p[0], p[1], b);
    return acc;
}
__device__ __forceinline__ Fr eval_linexp(const SolveCtx &cx, ProgPtr &p, size_t b) {
    uint32_t n = *p++;
    return eval_terms(cx, p, n, b);
}
// Cooperative evaluation of a long linear expression by the SL "slices" of a workgroup (threadIdx.y): the add32 /
// FromBinary constraints have ~100 terms and, evaluated by one lane, set the latency of every dependency level.
// Must be called by all threads of the workgroup (n and p are workgroup-uniform).  Result valid in slice 0.
// Workgroup shapes (instances x slices = 256 threads): 64 x 4 for real batches; 16 x 16 and 8 x 32 for the small batches of
// single Prove() calls, where spare lanes are better spent on the terms of an expression than on instances that do not exist.
template <int LX, int SL>
__device__ __forceinline__ Fr eval_terms_coop(const SolveCtx &cx, ProgPtr &p, uint32_t n, size_t b, Fr *red) {
    const int slice = threadIdx.y, lane = threadIdx.x;
    if (n < COOP_MIN_TERMS) {
        if (slice == 0) return eval_terms(cx, p, n, b);
        p += 2 * n;
        return Fr::zero();
    }
    const uint32_t chunk = (n + SL - 1) / SL;
    const uint32_t lo = slice * chunk, hi = lo + chunk < n ? lo + chunk : n;
    ProgPtr q = p + 2 * lo;
    Fr part = lo < hi ? eval_terms(cx, q, hi - lo, b) : Fr::zero();
    p += 2 * n;
    red[slice * LX + lane] = part;
    __syncthreads();
    Fr acc = Fr::zero();
    if (slice == 0) {
#pragma unroll 4
        for (int k = 0; k < SL; k++) acc = acc + red[k * LX + lane];
    }
    __syncthreads();
    return acc;
}
template <int LX, int SL>
__device__ __forceinline__ Fr eval_linexp_coop(const SolveCtx &cx, ProgPtr &p, size_t b, Fr *red) {
    uint32_t n = *p++;
    return eval_terms_coop<LX, SL>(cx, p, n, b, red);
}
__device__ __forceinline__ void report(const SolveCtx &cx, size_t b, uint32_t code, uint32_t detail) {
    atomicCAS(&cx.err[b], 0u, code | (detail << 8));
}
__device__ __forceinline__ bool small_index(const Fr &v_mont, uint32_t bound, uint32_t &idx) {
    Fr c = v_mont.from_mont();
    uint32_t hi = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) hi |= c.l[i];
    idx = c.l[0];
    return hi == 0 && idx < bound;
}

// One instruction.  COOP: the SL slices of the workgroup work on it together (long linear expressions, countHint queries; slice 0
// finishes it); !COOP: the calling wave owns it (no barriers, no LDS) -- lead = live.
// The tail of an R1C instruction once its three linear expressions are known: check (loc 0) or solve for the one unknown wire and
// write the constraint's a / b / c values.  INLINE_DIV: an unknown in L or R is divided out here (one inversion per instruction:
// the persistent small-batch kernel); otherwise that case is left to k_solve_divisions, which shares one inversion between 16.
template <bool INLINE_DIV>
__device__ __forceinline__ void finish_r1c(const SolveCtx &cx, size_t b, uint32_t cons, uint32_t loc, uint32_t sw, uint32_t sc, Fr vL, Fr vR, Fr vO) {
    const size_t B = cx.B;
    if (loc == 0) {
        if (mul_fast(vL, vR) != vO) report(cx, b, SERR_UNSATISFIED, cons);
    } else {
        Fr x = Fr::zero();
        if (loc == 3) {
            x = mul_fast(vL, vR) - vO;
            vO = vO + x;
        } else if constexpr (INLINE_DIV) {
            // x = O / R - L (loc 1) or O / L - R (loc 2); same case analysis as k_solve_divisions
            const Fr den = loc == 1 ? vR : vL;
            if (!den.is_zero()) {
                const Fr di = den.inv();
                if (loc == 1) { x = vO * di - vL; vL = vL + x; }
                else { x = vO * di - vR; vR = vR + x; }
            } else if (!vO.is_zero()) {
                report(cx, b, SERR_UNSATISFIED, cons);
            }
        } else {
            // unknown wire sits in L or R: x = O / R - L needs a field inversion.  Deferred to k_solve_divisions,
            // which shares ONE inversion between up to 16 such constraints of this level (Montgomery's trick);
            // a/b/c written below carry the partial sums it needs.
            cx.a[(size_t)cons * B + b] = vL;
            cx.b[(size_t)cons * B + b] = vR;
            cx.c[(size_t)cons * B + b] = vO;
            return;
        }
        if (sc == 3) x = x.neg();
        else if (sc != 1) x = x * tab_load(cx.coeff_inv + sc);
        cx.wires[(size_t)sw * B + b] = x;
    }
    cx.a[(size_t)cons * B + b] = vL;
    cx.b[(size_t)cons * B + b] = vR;
    cx.c[(size_t)cons * B + b] = vO;
}
__device__ __forceinline__ void finish_nbits(const SolveCtx &cx, size_t b, uint32_t nout, uint32_t out0, Fr v) {
    v = v.from_mont();
    const Fr one = Fr::one(), zero = Fr::zero();
    for (uint32_t k = 0; k < nout; k++) {
        const bool bit = k < 256 && ((v.l[k >> 5] >> (k & 31)) & 1);
        cx.wires[(size_t)(out0 + k) * cx.B + b] = bit ? one : zero;
    }
}

template <int LX, int SL, bool COOP, bool INLINE_DIV = false>
__device__ __forceinline__ void solve_instruction(const SolveCtx &cx, const uint32_t *__restrict__ instr_off, uint32_t instr, size_t b, bool live, bool lead, Fr *red) {
    const size_t B = cx.B;
    ProgPtr p = cx.prog + instr_off[instr];
    const uint32_t op = p[0];
    if (op == OP_R1C) {
        const uint32_t cons = p[1], nL = p[2], nR = p[3], nO = p[4], loc = p[5], sw = p[6], sc = p[7];
        p += 8;
        Fr vL, vR, vO;
        if constexpr (COOP) {
            vL = eval_terms_coop<LX, SL>(cx, p, nL, b, red);
            vR = eval_terms_coop<LX, SL>(cx, p, nR, b, red);
            vO = eval_terms_coop<LX, SL>(cx, p, nO, b, red);
        } else {
            vL = eval_terms(cx, p, nL, b);
            vR = eval_terms(cx, p, nR, b);
            vO = eval_terms(cx, p, nO, b);
        }
        if (!lead) return;
        finish_r1c<INLINE_DIV>(cx, b, cons, loc, sw, sc, vL, vR, vO);
        return;
    } else if (op == OP_NBITS) {
        const uint32_t nout = p[1], out0 = p[2];
        p += 3;
        Fr v;
        if constexpr (COOP) v = eval_linexp_coop<LX, SL>(cx, p, b, red);
        else v = eval_linexp(cx, p, b);
        if (!lead) return;
        finish_nbits(cx, b, nout, out0, v);
    } else if (COOP && op == OP_COUNT) {
        // countHint of a log-derivative lookup: multiplicity of every table row among nq (~3 600 for the AES S-box) queries.
        // The slices take every SL-th query (one lane used to walk all of them: the three longest levels of the AES circuits,
        // 3-4 ms each); counters live in global memory, hence atomics.
        const uint32_t ntab = p[1], ncol = p[2], nq = p[3], out0 = p[4], slot = p[5], toff = p[6];
        ProgPtr row_start = p + 8 + p[7];
        p += 8;
        const uint32_t slice = threadIdx.y;
        uint32_t *cnt = cx.count_scratch + (size_t)slot * cx.count_rows * B;
        if (live)
            for (uint32_t r = slice; r < ntab; r += SL) cnt[(size_t)r * B + b] = 0;
        __syncthreads();
        FrTab tab = cx.count_tables + toff;
        for (uint32_t j = slice; j < nq; j += SL) {
            ProgPtr pq = p + row_start[j];
            // first column of a logderivlookup table row is the row index: O(1) candidate, then full-row compare
            uint32_t row = 0;
            bool ok = true;
            for (uint32_t col = 0; col < ncol; col++) {
                const Fr qv = eval_linexp(cx, pq, b);
                if (col == 0) ok = small_index(qv, ntab, row);
                if (ok && tab_load(tab + ((size_t)row * ncol + col)) != qv) ok = false;
            }
            if (!live) continue;
            if (ok) atomicAdd(&cnt[(size_t)row * B + b], 1u);
            else report(cx, b, SERR_COUNT_QUERY, instr);
        }
        __syncthreads();
        if (live)
            for (uint32_t r = slice; r < ntab; r += SL) cx.wires[(size_t)(out0 + r) * B + b] = Fr::from_u32(cnt[(size_t)r * B + b]);
    } else if (!lead) {
        return;                                   // the remaining ops have short expressions: slice 0 alone
    } else if (op == OP_LOOKUP) {
        const uint32_t tb = p[1], nq = p[2], w0 = p[3], nent = p[4];
        p += 5;
        for (uint32_t q = 0; q < nq; q++) {
            Fr v = eval_linexp(cx, p, b);
            uint32_t idx;
            if (!small_index(v, nent, idx)) { report(cx, b, SERR_LOOKUP_RANGE, instr); idx = 0; }
            cx.wires[(size_t)(w0 + q) * B + b] = tab_load(cx.tables + ((size_t)tb * nent + idx));
        }
    } else if (op == OP_RANDOMIZE) {
        cx.wires[(size_t)p[1] * B + b] = cx.randomize[b];
    } else if (op == OP_BSB22) {
        cx.wires[(size_t)p[1] * B + b] = cx.challenge[b];
    }
}

// The countHints of a level in a launch of their own.  One workgroup per hint and instance tile walked the ~3 600 queries of an AES S-box
// lookup with its SL slices -- 112 (8 x 32) to 900 (64 x 4) queries one after the other per slice: the four longest levels of the AES
// programs, 0.14-0.17 ms each for a single proof and 0.16-0.55 ms at 64 (1.65 of the 6.3 ms its 134 level launches take).  Here G
// workgroups share a hint: workgroup `part` takes the queries part * SL + slice, + G * SL, ...; the counters live in global memory
// (zeroed by solver_begin) and are incremented with agent-scope atomics as before; the workgroup that draws the LAST ticket of the
// (hint, tile) pair -- every workgroup fences its increments, then takes one -- writes the output wires.  No workgroup waits for another.
template <int LX, int SL>
__global__ __launch_bounds__(LX * SL) void k_solve_counts(SolveCtx cx, const uint32_t *__restrict__ instr_off, const uint32_t *__restrict__ level_instr, uint32_t G) {
    __shared__ uint32_t last_s;
    const size_t B = cx.B;
    const size_t b_raw = (size_t)blockIdx.x * LX + threadIdx.x;
    const bool live = b_raw < B;
    const size_t b = live ? b_raw : B - 1;
    const uint32_t which = blockIdx.y / G, part = blockIdx.y % G, slice = threadIdx.y;
    const uint32_t instr = level_instr[which];
    ProgPtr p = cx.prog + instr_off[instr];            // an OP_COUNT record (SolverProgram::build puts them in front of their level)
    const uint32_t ntab = p[1], ncol = p[2], nq = p[3], out0 = p[4], slot = p[5], toff = p[6];
    ProgPtr row_start = p + 8 + p[7];
    p += 8;
    uint32_t *cnt = cx.count_scratch + (size_t)slot * cx.count_rows * B;
    FrTab tab = cx.count_tables + toff;
    for (uint32_t j = part * SL + slice; j < nq; j += G * SL) {
        ProgPtr pq = p + row_start[j];
        uint32_t row = 0;
        bool ok = true;
        for (uint32_t col = 0; col < ncol; col++) {
            const Fr qv = eval_linexp(cx, pq, b);
            if (col == 0) ok = small_index(qv, ntab, row);
            if (ok && tab_load(tab + ((size_t)row * ncol + col)) != qv) ok = false;
        }
        if (!live) continue;
        if (ok) atomicAdd(&cnt[(size_t)row * B + b], 1u);
        else report(cx, b, SERR_COUNT_QUERY, instr);
    }
    __threadfence();                                   // this thread's increments are performed before the workgroup's ticket is drawn
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0)
        last_s = __hip_atomic_fetch_add(&cx.count_ticket[(size_t)slot * gridDim.x + blockIdx.x], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == G - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_s || !live) return;
    for (uint32_t r = slice; r < ntab; r += SL)
        cx.wires[(size_t)(out0 + r) * B + b] = Fr::from_u32(__hip_atomic_load(&cnt[(size_t)r * B + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// grid = (ceil(B / LX), workgroups of the level), block = (LX instances, SL slices).  The first `nwide` workgroup rows take one
// instruction each: the instruction is uniform per workgroup, so the program words are scalar loads and all control flow on them is
// wave-uniform; the slices share the evaluation of long linear expressions (and the queries of a countHint), slice 0 finishes the
// instruction.  The rows after them (64 x 4 shape only) take FOUR instructions each, one per wavefront: the instructions whose
// expressions are all short (< COOP_MIN_TERMS terms: 3/4 of the ChaCha20 program, 9/10 of the AES ones) have nothing for the other
// three slices to do, and a level's duration is the number of wavefronts it sends through the CUs times their memory round trips
// (SolverProgram orders every level wide-first and counts them).
// MINB: workgroups per CU the register budget is set for -- 1: 151 registers, 3 resident; 4: 128 registers (96 more bytes of scratch),
// 4 resident.  Same box, batch 1024: ChaCha20 88.7 ms per step with 4, 90.9 with 1; AES-256 221.1 with 4, 218.8 with 1 (the countHint
// and division-heavy levels pay for the spills) -- solver_run_levels picks by the program's level count, ZKPROVE_SOLVER_OCC=3|4 overrides.
template <int LX, int SL, int MINB = 1>
__global__ __launch_bounds__(LX * SL, MINB) void k_solve_level(SolveCtx cx, const uint32_t *__restrict__ instr_off, const uint32_t *__restrict__ level_instr,
                                                     uint32_t count, uint32_t nwide) {
    static_assert(LX * SL == 256 || (LX == 64 && SL == 16), "workgroup shape");
    __shared__ Fr red[SL * LX];
    const size_t B = cx.B;
    const size_t b_raw = (size_t)blockIdx.x * LX + threadIdx.x;
    const bool live = b_raw < B;
    const size_t b = live ? b_raw : B - 1;            // dead lanes shadow the last instance (no early exit before barriers)
    if (blockIdx.y < nwide) {
        if (blockIdx.y >= count) return;
        solve_instruction<LX, SL, true>(cx, instr_off, level_instr[blockIdx.y], b, live, threadIdx.y == 0 && live, red);
        return;
    }
    if constexpr (LX == 64) {
        const uint32_t slice = __builtin_amdgcn_readfirstlane(threadIdx.y);      // a wavefront = one slice: uniform
        const uint32_t idx = nwide + (blockIdx.y - nwide) * SL + slice;
        if (idx >= count) return;
        solve_instruction<LX, SL, false>(cx, instr_off, level_instr[idx], b, live, live, red);
    }
}

#ifdef ZK_EXPERIMENTS
// ---------------------------------------------------------------------------------------- persistent small-batch solver (EXPERIMENT: lost)
// A single Prove() call (what the reference's own harness and the unchanged JS operator issue, gnark/libraries/core_test.go:121-150) is
// latency: the level-per-launch kernels above cost 163 / 457 dependent launches at ~15 us each for ONE proof (2.9 / 6.8 ms).  Proofs
// are independent, so no device-wide barrier is needed to keep a whole solve inside one kernel (VERDICT r3 task 3): a workgroup owns
// ONE instance and walks ALL levels of [l0, l1) with __syncthreads() between them.  Measured (scripts/latency_probe.py, one box,
// profiles/r04c_lat_*.txt): SLOWER -- ChaCha20 solve stage 3.6 ms (512 threads) / 4.3 (1 024) against 2.9 ms for the 163 launches,
// AES-256 16.9 / 19.2 against 6.8 -- although every level's instructions are sorted by shape: with the INSTRUCTIONS on the lanes
// every program word is a vector load, every branch of the term loop (coefficient 0 / 1 / 2 / -1 / general, operand 0 / +-1) is
// taken by some lane of the wavefront, and a lane that needs the 254-bit product makes all 64 pay for it; the level kernels keep the
// instruction uniform per workgroup (scalar loads, uniform branches) and put the INSTANCES (or, for B <= 8, the terms) on the lanes.
// Kept in the experiments build (ZKPROVE_SOLVER_PERSIST=n: batches up to n), parity-tested there.
// Within a level the lanes are the INSTRUCTIONS (they are independent by construction of the levels):
//   phase A  wide instructions -- a linear expression of >= COOP_MIN_TERMS terms: the ~100-term add32 / FromBinary rows that used to
//            set every level's latency -- are shared by 16 lanes each (partial sums through LDS, the group's first lane finishes);
//   counts   a countHint (AES: 3 600 S-box queries) is worked on by the whole workgroup, counters in global memory as above;
//   phase B  every other instruction by one lane; an unknown in L or R is divided out in place (one binary-GCD inversion per
//            instruction -- there is nobody to share it with).
// Same arithmetic, same case analysis, hence the same wires and a / b / c rows as k_solve_level + k_solve_divisions (parity tests:
// batches of 1, 7, 12 against the oracle's wire vectors).  grid = B workgroups; used up to PERSIST_MAX_B instances.
template <int G>
__device__ __forceinline__ Fr eval_terms_slice(const SolveCtx &cx, ProgPtr p, uint32_t n, size_t b, int gl) {
    if (n < (uint32_t)COOP_MIN_TERMS) {
        if (gl != 0) return Fr::zero();
        return eval_terms(cx, p, n, b);
    }
    const uint32_t chunk = (n + G - 1) / G;
    const uint32_t lo = gl * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (lo >= hi) return Fr::zero();
    ProgPtr q = p + 2 * lo;
    return eval_terms(cx, q, hi - lo, b);
}
template <int NT>
__global__ __launch_bounds__(NT) void k_solve_persistent(SolveCtx cx, const uint32_t *__restrict__ instr_off, const uint32_t *__restrict__ level_instr,
                                                        const uint32_t *__restrict__ level_desc, uint32_t l0, uint32_t l1) {
    constexpr int G = 16, NG = NT / G;
    extern __shared__ uint4 lds_raw[];
    Fr *red = (Fr *)lds_raw;                          // [3][NT] partial sums of L, R, O
    __shared__ uint32_t n_counts, count_idx[16];
    const int tid = threadIdx.x, grp = tid / G, gl = tid % G;
    const size_t b = blockIdx.x, B = cx.B;
    for (uint32_t l = l0; l < l1; l++) {
        const uint32_t first = level_desc[5 * l], count = level_desc[5 * l + 1], wide = level_desc[5 * l + 2];
        const uint32_t *li = level_instr + first;
        if (tid == 0) n_counts = 0;
        if (wide) __syncthreads();
        // ---- phase A
        for (uint32_t base = 0; base < wide; base += NG) {
            const uint32_t idx = base + grp;
            const bool mine = idx < wide;
            Fr pL = Fr::zero(), pR = Fr::zero(), pO = Fr::zero();
            uint32_t op = 0, h1 = 0, h2 = 0, h5 = 0, h6 = 0, h7 = 0;
            if (mine) {
                ProgPtr p = cx.prog + instr_off[li[idx]];
                op = p[0];
                if (op == OP_R1C) {
                    const uint32_t nL = p[2], nR = p[3], nO = p[4];
                    h1 = p[1]; h5 = p[5]; h6 = p[6]; h7 = p[7];
                    p += 8;
                    pL = eval_terms_slice<G>(cx, p, nL, b, gl); p += 2 * nL;
                    pR = eval_terms_slice<G>(cx, p, nR, b, gl); p += 2 * nR;
                    pO = eval_terms_slice<G>(cx, p, nO, b, gl);
                } else if (op == OP_NBITS) {
                    h1 = p[1]; h2 = p[2];
                    pL = eval_terms_slice<G>(cx, p + 4, p[3], b, gl);
                } else if (op == OP_COUNT && gl == 0) {
                    const uint32_t k = atomicAdd(&n_counts, 1u);
                    if (k < 16) count_idx[k] = idx;
                }
            }
            red[tid] = pL; red[NT + tid] = pR; red[2 * NT + tid] = pO;
            __syncthreads();
            if (mine && gl == 0 && (op == OP_R1C || op == OP_NBITS)) {
                Fr vL = Fr::zero(), vR = Fr::zero(), vO = Fr::zero();
#pragma unroll 4
                for (int k = 0; k < G; k++) vL = vL + red[tid + k];
                if (op == OP_R1C) {
#pragma unroll 4
                    for (int k = 0; k < G; k++) { vR = vR + red[NT + tid + k]; vO = vO + red[2 * NT + tid + k]; }
                    finish_r1c<true>(cx, b, h1, h5, h6, h7, vL, vR, vO);
                } else finish_nbits(cx, b, h1, h2, vL);
            }
            __syncthreads();
        }
        // ---- countHints of this level (collected above), the whole workgroup on each
        const uint32_t nc = wide ? (n_counts < 16u ? n_counts : 16u) : 0u;
        for (uint32_t c = 0; c < nc; c++) {
            const uint32_t instr = li[count_idx[c]];
            ProgPtr p = cx.prog + instr_off[instr];
            const uint32_t ntab = p[1], ncol = p[2], nq = p[3], out0 = p[4], slot = p[5], toff = p[6];
            ProgPtr row_start = p + 8 + p[7];
            p += 8;
            uint32_t *cnt = cx.count_scratch + (size_t)slot * cx.count_rows * B;
            for (uint32_t r = tid; r < ntab; r += NT) cnt[(size_t)r * B + b] = 0;
            __syncthreads();
            FrTab tab = cx.count_tables + toff;
            for (uint32_t j = tid; j < nq; j += NT) {
                ProgPtr pq = p + row_start[j];
                uint32_t row = 0;
                bool ok = true;
                for (uint32_t col = 0; col < ncol; col++) {
                    const Fr qv = eval_linexp(cx, pq, b);
                    if (col == 0) ok = small_index(qv, ntab, row);
                    if (ok && tab_load(tab + ((size_t)row * ncol + col)) != qv) ok = false;
                }
                if (ok) atomicAdd(&cnt[(size_t)row * B + b], 1u);
                else report(cx, b, SERR_COUNT_QUERY, instr);
            }
            __syncthreads();
            for (uint32_t r = tid; r < ntab; r += NT) cx.wires[(size_t)(out0 + r) * B + b] = Fr::from_u32(cnt[(size_t)r * B + b]);
        }
        // ---- phase B
        for (uint32_t i = wide + tid; i < count; i += NT) solve_instruction<1, 1, false, true>(cx, instr_off, li[i], b, true, true, nullptr);
        __syncthreads();                                   // the level's wires are written: the next level may read them
    }
}

#endif  // ZK_EXPERIMENTS

// Deferred divisions of one level: thread = (instance, group of up to DIV_GROUP constraints).
// rec = (constraint, loc, wire, coeff id).  loc 1: x = c/b - a ; loc 2: x = c/a - b.
// DIV_GROUP constraints share one inversion: 16 for real batches (the inversion is 73 products' worth, the trick costs 3 per member);
// 2 for the batches of single Prove() calls, where a thread's 16 members are a sequential chain on an otherwise idle device
// (AES-256, one proof: the 11 division launches of a call 1.5 ms -> 0.6 ms)
template <int DIV_GROUP>
__device__ __forceinline__ void solve_division_group(const SolveCtx &cx, const uint32_t *__restrict__ recs, uint32_t count, uint32_t g0, size_t b) {
    const size_t B = cx.B;
    const uint32_t g1 = g0 + DIV_GROUP < count ? g0 + DIV_GROUP : count;
    Fr pref[DIV_GROUP];
    Fr acc = Fr::one();
#pragma unroll
    for (int k = 0; k < DIV_GROUP; k++) {
        pref[k] = acc;
        if (g0 + k < g1) {
            const uint32_t cons = recs[4 * (g0 + k)], loc = recs[4 * (g0 + k) + 1];
            const Fr den = loc == 1 ? cx.b[(size_t)cons * B + b] : cx.a[(size_t)cons * B + b];
            if (!den.is_zero()) acc = acc * den;
        }
    }
    Fr inv = acc.inv();
#pragma unroll
    for (int k = DIV_GROUP - 1; k >= 0; k--) {
        if (g0 + k < g1) {
            const uint32_t cons = recs[4 * (g0 + k)], loc = recs[4 * (g0 + k) + 1], sw = recs[4 * (g0 + k) + 2], sc = recs[4 * (g0 + k) + 3];
            const size_t o = (size_t)cons * B + b;
            const Fr va = cx.a[o], vb = cx.b[o], vc = cx.c[o];
            const Fr den = loc == 1 ? vb : va;
            Fr x = Fr::zero();
            if (!den.is_zero()) {
                const Fr di = inv * pref[k];
                inv = inv * den;
                if (loc == 1) { x = vc * di - va; cx.a[o] = va + x; }
                else { x = vc * di - vb; cx.b[o] = vb + x; }
            } else if (!vc.is_zero()) {
                report(cx, b, SERR_UNSATISFIED, cons);
            }
            if (sc == 3) x = x.neg();
            else if (sc != 1) x = x * tab_load(cx.coeff_inv + sc);
            cx.wires[(size_t)sw * B + b] = x;
        }
    }
}
template <int DIV_GROUP>
__global__ __launch_bounds__(256) void k_solve_divisions(SolveCtx cx, const uint32_t *__restrict__ recs, uint32_t count) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= cx.B) return;
    solve_division_group<DIV_GROUP>(cx, recs, count, blockIdx.y * DIV_GROUP, b);
}

// A RUN of consecutive narrow levels (at most NARROW_MAX instructions each, no countHint, no commitment) in ONE launch.  The AES
// programs end in such a run: 321 of the 457 levels of aes-256 hold 1-16 instructions (the running sums of the log-derivative
// arguments), i.e. 321 dependent launches of a handful of workgroups each -- 2.3 ms of a single Prove() call, and the long tail that
// made the solve stage of a 1 024-batch 28 ms.  Proofs are independent, so a workgroup that owns a tile of instances can walk all
// levels of the run by itself with __syncthreads() between them; unlike the persistent-solver experiment above the lanes stay the
// INSTANCES (program words are scalar loads, every branch on them is uniform): same instruction code, same division code, same
// results.  grid = tiles of 64 instances, block = (64 instances, SL wavefronts): every wavefront takes one instruction of the level
// (SL = 16: a narrow level is ONE round, also for a single proof, where 15 more workgroups used to run beside the first).
template <int LX, int SL>
__global__ __launch_bounds__(LX * SL) void k_solve_narrow_run(SolveCtx cx, const uint32_t *__restrict__ instr_off, const uint32_t *__restrict__ level_instr,
                                                        const uint32_t *__restrict__ level_desc, const uint32_t *__restrict__ div_list, uint32_t l0, uint32_t l1) {
    static_assert(LX == 64, "a wavefront per slice");
    __shared__ Fr red[SL * LX];
    const size_t b_raw = (size_t)blockIdx.x * LX + threadIdx.x;
    const bool live = b_raw < cx.B;
    const size_t b = live ? b_raw : cx.B - 1;
    const uint32_t slice = __builtin_amdgcn_readfirstlane(threadIdx.y);
    for (uint32_t l = l0; l < l1; l++) {
        const uint32_t first = level_desc[5 * l], count = level_desc[5 * l + 1], wide = level_desc[5 * l + 2], div_first = level_desc[5 * l + 3],
                       div_count = level_desc[5 * l + 4];
        const uint32_t *li = level_instr + first;
        // long expressions: the slices share them, one instruction at a time (barriers inside: every thread takes part)
        for (uint32_t i = 0; i < wide; i++) solve_instruction<LX, SL, true>(cx, instr_off, li[i], b, live, threadIdx.y == 0 && live, red);
        for (uint32_t i = wide + slice; i < count; i += SL) solve_instruction<LX, SL, false>(cx, instr_off, li[i], b, live, live, red);
        // The last wavefront (idle in almost every level: a narrow level holds ~3 instructions) walks the NEXT level's program now --
        // description -> instruction numbers -> offsets -> records -- so that the scalar cache holds it when the level starts: the
        // workers' chain of five dependent scalar loads then hits the cache (a level is 7 us, ~2 of them that chain)
        if (slice == SL - 1 && l + 1 < l1) {
            const uint32_t f2 = level_desc[5 * (l + 1)], c2 = level_desc[5 * (l + 1) + 1];
            uint32_t sink = level_desc[5 * (l + 1) + 2] ^ level_desc[5 * (l + 1) + 3] ^ level_desc[5 * (l + 1) + 4];
            for (uint32_t i = 0; i < c2 && i < NARROW_MAX; i++) {
                ProgPtr p = cx.prog + instr_off[level_instr[f2 + i]];
                uint32_t words = p[0] == OP_R1C ? 8 + 2 * (p[2] + p[3] + p[4]) : (p[0] == OP_NBITS ? 4 + 2 * p[3] : 16u);
                if (words > 512) words = 512;
                for (uint32_t k = 0; k < words; k += 16) sink ^= p[k];
                sink ^= p[words - 1];
            }
            const uint32_t d2 = level_desc[5 * (l + 1) + 3], dc2 = level_desc[5 * (l + 1) + 4];
            for (uint32_t k = 0; k < 4 * dc2 && k < 64; k += 16) sink ^= div_list[4 * (size_t)d2 + k];
            asm volatile("" ::"s"(sink));
        }
        __syncthreads();
        if (div_count) {
            constexpr int G = 2;                               // at most 16 divisions per level: a wavefront per pair
            for (uint32_t g = threadIdx.y; g * G < div_count; g += SL)
                if (live) solve_division_group<G>(cx, div_list + 4 * (size_t)div_first, div_count, g * G, b);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------- chain runs: a wavefront per component, no barrier
// The run of narrow levels that ends the AES programs is five permutations of 321 dependent multiplications ( (x + k)^2, x^2, x * (y + k),
// 107 rounds ) plus a few check rows: k_solve_narrow_run pays a workgroup barrier, the chain of scalar loads to the instruction and a
// store -> L2 -> load round trip of the wire per LEVEL (5.9 us, 1.9 ms of a single Prove() call and of every batch up to 128).
// SolverProgram::build splits such a run into components that read none of each other's wires; here ONE wavefront walks each component
// from the first level to the last.  No barrier: a lane reads only wires of its own proof, written by itself (program order) or before
// the launch.  The last four solved wires stay in registers (the permutations read the wires of the previous one to three levels), so
// the multiplications follow each other at register speed; the stores to a / b / c / wires drain beside them.  Same evaluation code
// as solve_instruction (term_apply, mul_fast, finish_r1c's arithmetic): same wires, same rows.
struct ChainFwd {
    Fr v0, v1, v2, v3;               // the last four wires this wavefront solved
    uint32_t i0, i1, i2, i3;
};
__device__ __forceinline__ Fr chain_wire(const SolveCtx &cx, const ChainFwd &f, uint32_t w, size_t b) {
    if (w == f.i0) return f.v0;
    if (w == f.i1) return f.v1;
    if (w == f.i2) return f.v2;
    if (w == f.i3) return f.v3;
    return cx.wires[(size_t)w * cx.B + b];
}
// What a step needs beside the wires is the same for every lane -- constraint and wire numbers, the folded constants, the terms -- but
// inside this loop the compiler cannot prove that the loop's own stores leave the program alone and fetches it with VECTOR loads: four
// dependent ~1 us round trips per step (offset -> header -> terms -> coefficient) in the first version of this kernel (4 us a step).
// Scalar loads by hand: the step's 16-word record (SolverProgram::build: constraint, location and number of the solved wire, its
// coefficient, the sides' dynamic term counts, the sides' constants, four dynamic terms) in one, the three constants in a second.
// Measured per step after that (one proof, wall_clock64 stamps): record 0.12 us, a term 0.4-0.5 us (the selections of term_apply /
// mul_fast, hence the constants folded at build time), the multiplication 1.2 us (issue-bound: 128 quarter-rate multiply-adds of ONE wavefront).
typedef uint32_t chain_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t chain_u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ Fr chain_fr(const chain_u32x8 &k) {
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = k[j];
    return r;
}
__global__ __launch_bounds__(64 * CHAIN_MAX_WAVES) void k_solve_chains(SolveCtx cx, const uint32_t *__restrict__ chain_recs, const uint32_t *__restrict__ wave_first,
                                                                      const Fr *__restrict__ consts) {
    const size_t B = cx.B;
    const size_t b_raw = (size_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < B;
    const size_t b = live ? b_raw : B - 1;               // dead lanes shadow the last instance and store nothing
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const uint32_t k0 = __builtin_amdgcn_readfirstlane(wave_first[w]), k1 = __builtin_amdgcn_readfirstlane(wave_first[w + 1]);
    ChainFwd f;
    f.v0 = f.v1 = f.v2 = f.v3 = Fr::zero();
    f.i0 = f.i1 = f.i2 = f.i3 = 0xFFFFFFFEu;             // no wire has this number (WIRE_CONST is 0xFFFFFFFF)
    // (the loops stay rolled and the multiplication is a call: a single wavefront per SIMD has nobody to hide an instruction-cache miss behind)
#pragma unroll 1
    for (uint32_t k = k0; k < k1; k++) {
        chain_u32x16 r;
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(chain_recs + 16 * (size_t)k));
        uint32_t cons, loc, sw, sc;
        Fr vL, vR, vO;
        if (r[4] == 0xFFFFFFFFu) {                       // a long row: the program's own record, every wire from memory (a lane's own stores are visible to it)
            ProgPtr p = cx.prog + r[5];
            cons = p[1]; loc = p[5]; sw = p[6]; sc = p[7];
            const uint32_t nL = p[2], nR = p[3], nO = p[4];
            p += 8;
            auto side = [&](uint32_t n) {
                Fr a = Fr::zero();
#pragma unroll 1
                for (uint32_t q = 0; q < n; q++, p += 2) a = a + term_val(cx, p[0], p[1], b);
                return a;
            };
            vL = side(nL); vR = side(nR); vO = side(nO);
        } else {
            cons = r[0]; loc = r[1]; sw = r[2]; sc = r[3];
            const uint32_t nL = r[4] & 0xFF, nLR = nL + ((r[4] >> 8) & 0xFF), nT = nLR + (r[4] >> 16);
            chain_u32x8 qL, qR, qO;
            asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %4, 0x0\n\ts_load_dwordx8 %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(qL), "=&s"(qR), "=&s"(qO) : "s"(consts + r[5]), "s"(consts + r[6]), "s"(consts + r[7]));
            vL = chain_fr(qL); vR = chain_fr(qR); vO = chain_fr(qO);
#pragma unroll 1
            for (uint32_t t = 0; t < nT; t++) {
                uint32_t cid, wid;
                switch (t) {
                    case 0: cid = r[8]; wid = r[9]; break;
                    case 1: cid = r[10]; wid = r[11]; break;
                    case 2: cid = r[12]; wid = r[13]; break;
                    default: cid = r[14]; wid = r[15]; break;
                }
                Fr cval = Fr::zero();
                if (cid > 4) {                           // the coefficient's value: the same for every lane
                    chain_u32x8 q;
                    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(q) : "s"(cx.coeffs + cid));
                    cval = chain_fr(q);
                }
                const Fr v = term_apply_pre(cid, wid, chain_wire(cx, f, wid, b), cval);
                if (t < nL) vL = vL + v;
                else if (t < nLR) vR = vR + v;
                else vO = vO + v;
            }
        }
        if (loc == 0) {
            if (live && mul_fast(vL, vR) != vO) report(cx, b, SERR_UNSATISFIED, cons);
        } else {                                         // loc == 3 (SolverProgram::build admits nothing else)
            Fr x = vL * vR - vO;                         // (mul_fast's shortcuts for 0 / +-1 give the same canonical product)
            vO = vO + x;
            if (sc == 3) x = x.neg();
            else if (sc != 1) x = x * tab_load(cx.coeff_inv + sc);
            if (live) cx.wires[(size_t)sw * B + b] = x;
            f.v3 = f.v2; f.i3 = f.i2;
            f.v2 = f.v1; f.i2 = f.i1;
            f.v1 = f.v0; f.i1 = f.i0;
            f.v0 = x; f.i0 = sw;
        }
        if (live) {
            cx.a[(size_t)cons * B + b] = vL;
            cx.b[(size_t)cons * B + b] = vR;
            cx.c[(size_t)cons * B + b] = vO;
        }
    }
}

// ---------------------------------------------------------------------------------------- one-call batches: runs of WIDE levels in one launch
// A single AES Prove() call (B <= 8; VERDICT r4 task 5): its 118 wide levels -- up to ~100 instructions each, long rows and countHints --
// cost 2.5 ms as 118 dependent launches (21 us each: the launch gap plus, inside every launch, the chain  level -> instruction number
// -> program offset -> header -> terms -> wires  of dependent loads), and the eleven division launches around them another millisecond.
// Here a run of consecutive levels is ONE launch of G workgroups (G = the widest level of the run, at most WIDE_RUN_MAX_G: a fraction of
// the device, so every workgroup is resident as soon as the kernel starts on an idle device): workgroup w solves instructions w, w + G, ...
// of a level with the same code as k_solve_level<8, 32>, then all meet at a grid barrier -- one monotonic counter in HBM, lane 0 of every
// workgroup: agent-scope release, arrive, poll, agent-scope acquire (MI355X_MICROARCH.md, barrier-counter) -- and the level's deferred
// divisions follow behind a second one.  What a launch boundary cannot do: the PROGRAM of the next level does not depend on the wires, so
// a workgroup fetches its next instruction's words while it waits at the barrier; behind the barrier only the wire values are new.
// The barrier cannot hang: the poll is BOUNDED (spin_limit iterations, ~50 ms).  A workgroup that gives up raises an abort word that
// every poll also reads, all workgroups leave, and every instance's error word becomes SERR_RETRY: the host then repeats the call with
// the per-level launches (Algorithm::resident_prove).  (Co-residency is only delayed, never denied, when another stream's kernel
// holds the wave slots: its workgroups retire and ours start; the spinning ones keep their slots.)
static constexpr uint32_t WIDE_RUN_MAX_G = 96;
static constexpr uint32_t WIDE_RUN_SPIN_LIMIT = 200000;
// all threads of the workgroup call it; false: aborted (timeout here or in another workgroup)
__device__ __forceinline__ bool grid_barrier(uint32_t *sync, uint32_t target, uint32_t spin_limit, int *ok_s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wavefront drains its stores before the workgroup's barrier ...
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // ... and one lane releases them to the other XCDs
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        uint32_t spins = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || ++spins > spin_limit) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // this CU's vector L1 forgets what other workgroups have overwritten
        *ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_s != 0;
}
template <int LX, int SL>
__global__ __launch_bounds__(LX * SL) void k_solve_wide_run(SolveCtx cx, const uint32_t *__restrict__ instr_off, const uint32_t *__restrict__ level_instr,
                                                            const uint32_t *__restrict__ level_desc, const uint32_t *__restrict__ div_list, uint32_t l0, uint32_t l1,
                                                            uint32_t *__restrict__ sync, uint32_t sync_base, uint32_t spin_limit) {
    __shared__ Fr red[SL * LX];
    __shared__ int ok_s;
    const uint32_t G = gridDim.x, w = blockIdx.x;
    const size_t b_raw = threadIdx.x;                  // one tile of instances: B <= LX
    const bool live = b_raw < cx.B;
    const size_t b = live ? b_raw : cx.B - 1;
    uint32_t epoch = sync_base;                        // arrivals the counter holds when this launch starts (earlier launches of the call)
    if (threadIdx.x == 0 && threadIdx.y == 0) ok_s = 1;
    __syncthreads();
    for (uint32_t l = l0; l < l1; l++) {
        const uint32_t first = level_desc[5 * l], count = level_desc[5 * l + 1], div_first = level_desc[5 * l + 3], div_count = level_desc[5 * l + 4];
        const uint32_t *li = level_instr + first;
        for (uint32_t i = w; i < count; i += G) solve_instruction<LX, SL, true>(cx, instr_off, li[i], b, live, threadIdx.y == 0 && live, red);
        // the next level's program for this workgroup -- description, instruction number, offset, record -- is requested now, by the
        // last slice, and sits in the scalar cache when the barrier opens (k_solve_narrow_run's prefetcher, across workgroups here)
        if (threadIdx.y == SL - 1 && l + 1 < l1) {
            const uint32_t f2 = level_desc[5 * (l + 1)], c2 = level_desc[5 * (l + 1) + 1];
            uint32_t sink = level_desc[5 * (l + 1) + 2] ^ level_desc[5 * (l + 1) + 3] ^ level_desc[5 * (l + 1) + 4];
            if (w < c2) {
                ProgPtr p = cx.prog + instr_off[level_instr[f2 + w]];
                uint32_t words = p[0] == OP_R1C ? 8 + 2 * (p[2] + p[3] + p[4]) : (p[0] == OP_NBITS ? 4 + 2 * p[3] : 16u);
                if (words > 1024) words = 1024;
                for (uint32_t k = 0; k < words; k += 16) sink ^= p[k];
                sink ^= p[words - 1];
            }
            asm volatile("" ::"s"(sink));
        }
        epoch += G;
        if (!grid_barrier(sync, epoch, spin_limit, &ok_s)) break;
        if (div_count) {
            for (uint32_t g = w * SL + threadIdx.y; g * 2 < div_count; g += G * SL)
                if (live) solve_division_group<2>(cx, div_list + 4 * (size_t)div_first, div_count, g * 2, b);
            epoch += G;
            if (!grid_barrier(sync, epoch, spin_limit, &ok_s)) break;
        }
    }
    if (!ok_s && live && threadIdx.y == 0) cx.err[b] = SERR_RETRY;      // (ok_s is only ever cleared by an aborted barrier; every workgroup that saw one says so)
}

// ---------------------------------------------------------------------------------------- small batches: the lane program
// A single Prove() call -- what the reference's own harness and the unchanged JS operator issue (gnark/libraries/core_test.go:121-150) --
// is latency: one launch per level costs 163 (ChaCha20) / 137 (AES-256, after the narrow runs) dependent launches of ~15 us, 2.5 / 5 ms
// for one proof, and inside each the chain  instruction number -> program offset -> header -> terms -> wires  of dependent loads.
// Here ONE workgroup owns one instance and walks all levels with __syncthreads() between them (proofs are independent: no device-wide
// barrier), and the lanes are INSTRUCTIONS OF ONE SHAPE: SolverProgram::build regroups every level by shape and stores the numbers of
// the instructions of a shape as columns, so a wavefront loads 64 instructions' wire numbers with one coalesced load per term, then
// all their wire values, and runs the shape's arithmetic with control flow that is uniform across the wavefront (the shape words are
// scalar loads).  Two dependent memory round trips per 8 terms instead of one per program word.  (The first persistent kernel --
// k_solve_persistent, experiments build -- had the instructions on the lanes too, but each lane interpreted its own program record:
// vector loads of every program word, every branch of the interpreter taken by some lane; it lost to the level launches.)
// Long expressions (>= LP_WIDE_TERMS terms: the 96-term add32 / FromBinary rows) are a task of their own, the wavefront's lanes
// taking the TERMS and a butterfly of wave shuffles summing them; countHints are worked on by the whole workgroup; an unknown in L or R
// is divided out after the level's tasks, a lane per division.  Same arithmetic and case analysis as k_solve_level + k_solve_divisions,
// and field addition is exact and commutative, so wires and a / b / c rows are the same bits.
// Measured from inside (ZKPROVE_SOLVER_DBG=1, experiments build): a dependent load costs ~0.7 us, a long row ~10 us of which three
// quarters are the lone wavefront's arithmetic; with all 16 wavefronts busy a level is bound by the CU's instruction issue.  Fetching
// the wire-independent words one or two levels ahead was built twice and lost to its register bill; the kernel
// used for ChaCha20-sized programs (AES: a level of up to 100 tasks wants the whole device, and the level launches win there).
template <int CH>
__device__ __forceinline__ void lp_eval(const SolveCtx &cx, const uint32_t *__restrict__ tw, uint32_t T, uint32_t nL, uint32_t nLR, const uint32_t *__restrict__ f,
                                        uint32_t gc, size_t b, Fr &vL, Fr &vR, Fr &vO) {
    vL = Fr::zero(); vR = Fr::zero(); vO = Fr::zero();
    for (uint32_t t0 = 0; t0 < T; t0 += CH) {
        uint32_t idx[CH];
        Fr v[CH];
#pragma unroll
        for (int q = 0; q < CH; q++) idx[q] = t0 + q < T ? f[(size_t)(t0 + q) * gc] : 0u;
#pragma unroll
        for (int q = 0; q < CH; q++) v[q] = cx.wires[(size_t)idx[q] * cx.B + b];
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const uint32_t t = t0 + q;
            if (t >= T) break;
            const Fr val = term_apply(cx, tw[2 * t], tw[2 * t + 1] ? WIRE_CONST : 0u, v[q]);
            if (t < nL) vL = vL + val;
            else if (t < nLR) vR = vR + val;
            else vO = vO + val;
        }
    }
}
// butterfly sums over the wavefront, up to three values at a time (independent chains: the shuffles of one overlap the additions of
// another); which = bit 0: x, bit 1: y, bit 2: z (wavefront-uniform)
__device__ __forceinline__ void lp_wave_reduce3(Fr &x, Fr &y, Fr &z, uint32_t which) {
#pragma unroll 1
    for (int o = 32; o >= 1; o >>= 1) {
        Fr ox, oy, oz;
        if (which & 1u) {
#pragma unroll
            for (int i = 0; i < 8; i++) ox.l[i] = __shfl_xor(x.l[i], o);
        }
        if (which & 2u) {
#pragma unroll
            for (int i = 0; i < 8; i++) oy.l[i] = __shfl_xor(y.l[i], o);
        }
        if (which & 4u) {
#pragma unroll
            for (int i = 0; i < 8; i++) oz.l[i] = __shfl_xor(z.l[i], o);
        }
        if (which & 1u) x = x + ox;
        if (which & 2u) y = y + oy;
        if (which & 4u) z = z + oz;
    }
}
// sum over the wavefront of the terms t = lane, lane + 64, ... of a linear expression in the program encoding; every lane gets it
__device__ __forceinline__ Fr lp_wave_sum(const SolveCtx &cx, ProgPtr p, uint32_t n, size_t b, int lane) {
    Fr acc = Fr::zero();
    for (uint32_t t = lane; t < n; t += 64) acc = acc + term_val(cx, p[2 * t], p[2 * t + 1], b);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        Fr other;
#pragma unroll
        for (int i = 0; i < 8; i++) other.l[i] = __shfl_xor(acc.l[i], o);
        acc = acc + other;
    }
    return acc;
}
template <int NT, int CH>
__global__ __launch_bounds__(NT) void k_solve_lanes(SolveCtx cx, const uint32_t *__restrict__ shapes, const uint32_t *__restrict__ tasks, const uint32_t *__restrict__ soa,
                                                   const uint32_t *__restrict__ lvl, const uint32_t *__restrict__ level_desc, const uint32_t *__restrict__ div_list,
                                                   uint32_t l0, uint32_t l1) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t b = blockIdx.x, B = cx.B;
    for (uint32_t l = l0; l < l1; l++) {
#ifdef ZK_EXPERIMENTS
        if (cx.dbg && lane == 0 && b == 0) cx.dbg[((size_t)l * NW + wave) * 4] = wall_clock64();
#endif
        const uint32_t t_first = lvl[3 * l], n_tasks = lvl[3 * l + 1], n_count = lvl[3 * l + 2];
        for (uint32_t t = wave; t < n_tasks; t += NW) {
            const uint32_t *tk = tasks + 4 * (size_t)(t_first + t);
            const uint32_t kind = tk[0], w1 = tk[1], w2 = tk[2], w3 = tk[3];
            if (kind == LP_TASK_LANES) {
                const uint32_t n = w3 & 127u, gc = w3 >> 7;
                if ((uint32_t)lane >= n) continue;
                const uint32_t *sh = shapes + w1, *f = soa + w2 + lane;
                const uint32_t op = sh[0];
                Fr vL, vR, vO;
                if (op == OP_R1C) {
                    const uint32_t nL = sh[1], nR = sh[2], nO = sh[3], loc = sh[4], sc = sh[5];
                    const uint32_t cons = f[0], sw = f[gc];
                    lp_eval<CH>(cx, sh + 6, nL + nR + nO, nL, nL + nR, f + 2 * (size_t)gc, gc, b, vL, vR, vO);
                    finish_r1c<false>(cx, b, cons, loc, sw, sc, vL, vR, vO);
                } else if (op == OP_NBITS) {
                    const uint32_t nout = sh[1], T = sh[2], out0 = f[0];
                    lp_eval<CH>(cx, sh + 3, T, T, T, f + gc, gc, b, vL, vR, vO);
                    finish_nbits(cx, b, nout, out0, vL);
                } else if (op == OP_LOOKUP) {
                    const uint32_t tb = sh[1], nq = sh[2], nent = sh[3], w0 = f[0], instr = f[gc];
                    const uint32_t *q = sh + 4, *ff = f + 2 * (size_t)gc;
                    for (uint32_t k = 0; k < nq; k++) {
                        const uint32_t T = q[0];
                        lp_eval<CH>(cx, q + 1, T, T, T, ff, gc, b, vL, vR, vO);
                        q += 1 + 2 * T;
                        ff += (size_t)T * gc;
                        uint32_t idx;
                        if (!small_index(vL, nent, idx)) { report(cx, b, SERR_LOOKUP_RANGE, instr); idx = 0; }
                        cx.wires[(size_t)(w0 + k) * B + b] = tab_load(cx.tables + ((size_t)tb * nent + idx));
                    }
                } else if (op == OP_RANDOMIZE) {
                    cx.wires[(size_t)f[0] * B + b] = cx.randomize[b];
                } else if (op == OP_BSB22) {
                    cx.wires[(size_t)f[0] * B + b] = cx.challenge[b];
                }
            } else {       // LP_TASK_WIDE: one instruction, the lanes take its terms
                ProgPtr p = cx.prog + w1;
                const uint32_t op = p[0];
                if (op == OP_R1C || op == OP_NBITS) {
                    // the three expressions as ONE list of terms: all (coefficient, wire) words of this lane's terms are requested together,
                    // then all their wire values, then the three sums run as one butterfly -- two memory round trips and one reduction
                    // where three loops of  words -> wires -> reduce  followed each other (11.5 us per 130-term add32 row; 7 per nBits)
                    const bool r1c = op == OP_R1C;
                    const uint32_t nL = r1c ? p[2] : p[3], nLR = r1c ? p[2] + p[3] : nL, T = r1c ? nLR + p[4] : nL;
                    ProgPtr pt = p + (r1c ? 8 : 4);
                    uint32_t cw[3][2];
                    Fr wv[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const uint32_t tt = lane + 64 * k;
                        cw[k][0] = tt < T ? pt[2 * tt] : 0u;
                        cw[k][1] = tt < T ? pt[2 * tt + 1] : 0u;
                    }
                    Fr cv[3];                               // the coefficient values too: fetched behind a branch of term_apply they are a third round trip
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        wv[k] = cx.wires[(size_t)(cw[k][1] == WIRE_CONST ? 0u : cw[k][1]) * B + b];
                        cv[k] = tab_load(cx.coeffs + cw[k][0]);
                    }
                    Fr vL = Fr::zero(), vR = Fr::zero(), vO = Fr::zero();
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const uint32_t tt = lane + 64 * k;
                        if (tt < T) {
                            const Fr val = term_apply_pre(cw[k][0], cw[k][1], wv[k], cv[k]);
                            if (tt < nL) vL = vL + val;
                            else if (tt < nLR) vR = vR + val;
                            else vO = vO + val;
                        }
                    }
                    for (uint32_t tt = lane + 192; tt < T; tt += 64) {
                        const Fr val = term_val(cx, pt[2 * tt], pt[2 * tt + 1], b);
                        if (tt < nL) vL = vL + val;
                        else if (tt < nLR) vR = vR + val;
                        else vO = vO + val;
                    }
                    // (an R1C row is finished by lane 0, which already holds a one-term L: t = 0 is its own term)
                    lp_wave_reduce3(vL, vR, vO, r1c ? ((nL > 1 ? 1u : 0u) | 2u | 4u) : 1u);
                    if (r1c) {
                        if (lane == 0) finish_r1c<false>(cx, b, p[1], p[5], p[6], p[7], vL, vR, vO);
                    } else {
                        const uint32_t nout = p[1], out0 = p[2];
                        const Fr v = vL.from_mont();
                        const Fr one = Fr::one(), zero = Fr::zero();
                        for (uint32_t k = lane; k < nout; k += 64) {
                            const bool bit = k < 256 && ((v.l[k >> 5] >> (k & 31)) & 1);
                            cx.wires[(size_t)(out0 + k) * B + b] = bit ? one : zero;
                        }
                    }
                } else if (op == OP_LOOKUP) {
                    const uint32_t tb = p[1], nq = p[2], w0 = p[3], nent = p[4];
                    p += 5;
                    for (uint32_t k = 0; k < nq; k++) {
                        const uint32_t n = *p++;
                        const Fr v = lp_wave_sum(cx, p, n, b, lane);
                        p += 2 * n;
                        if (lane == 0) {
                            uint32_t idx;
                            if (!small_index(v, nent, idx)) { report(cx, b, SERR_LOOKUP_RANGE, w2); idx = 0; }
                            cx.wires[(size_t)(w0 + k) * B + b] = tab_load(cx.tables + ((size_t)tb * nent + idx));
                        }
                    }
                }
            }
        }
#ifdef ZK_EXPERIMENTS
        if (cx.dbg && lane == 0 && b == 0) cx.dbg[((size_t)l * NW + wave) * 4 + 1] = wall_clock64();
#endif
        // countHints of this level: the whole workgroup on each (same code as the slices of k_solve_level)
        for (uint32_t c = 0; c < n_count; c++) {
            ProgPtr p = cx.prog + tasks[4 * (size_t)(t_first + n_tasks + c) + 1];
            const uint32_t instr = tasks[4 * (size_t)(t_first + n_tasks + c) + 2];
            const uint32_t ntab = p[1], ncol = p[2], nq = p[3], out0 = p[4], slot = p[5], toff = p[6];
            ProgPtr row_start = p + 8 + p[7];
            p += 8;
            uint32_t *cnt = cx.count_scratch + (size_t)slot * cx.count_rows * B;
            for (uint32_t r = tid; r < ntab; r += NT) cnt[(size_t)r * B + b] = 0;
            __syncthreads();
            FrTab tab = cx.count_tables + toff;
            for (uint32_t j = tid; j < nq; j += NT) {
                ProgPtr pq = p + row_start[j];
                uint32_t row = 0;
                bool ok = true;
                for (uint32_t col = 0; col < ncol; col++) {
                    const Fr qv = eval_linexp(cx, pq, b);
                    if (col == 0) ok = small_index(qv, ntab, row);
                    if (ok && tab_load(tab + ((size_t)row * ncol + col)) != qv) ok = false;
                }
                if (ok) atomicAdd(&cnt[(size_t)row * B + b], 1u);
                else report(cx, b, SERR_COUNT_QUERY, instr);
            }
            __syncthreads();
            for (uint32_t r = tid; r < ntab; r += NT) cx.wires[(size_t)(out0 + r) * B + b] = Fr::from_u32(cnt[(size_t)r * B + b]);
        }
        // the level's divisions (an unknown in L or R; the instruction left its partial sums in the a / b / c rows): a lane each --
        // up to 515 in a level of the AES programs, and an inversion is a long dependent chain whoever else waits for it
        const uint32_t div_first = level_desc[5 * l + 3], div_count = level_desc[5 * l + 4];
        if (div_count) {
            __syncthreads();
            for (uint32_t g = tid; g < div_count; g += NT) solve_division_group<1>(cx, div_list + 4 * (size_t)div_first, div_count, g, b);
        }
#ifdef ZK_EXPERIMENTS
        if (cx.dbg && lane == 0 && b == 0) cx.dbg[((size_t)l * NW + wave) * 4 + 2] = wall_clock64();
#endif
        __syncthreads();                                   // the level's wires are written: the next level may read them
#ifdef ZK_EXPERIMENTS
        if (cx.dbg && lane == 0 && b == 0) cx.dbg[((size_t)l * NW + wave) * 4 + 3] = wall_clock64();
#endif
    }
}

__global__ __launch_bounds__(256) void k_assign_inputs(const uint32_t *__restrict__ in, Fr *__restrict__ wires, size_t n_inputs, size_t B) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (n_inputs + 1) * B) return;
    const size_t row = t / B;
    if (row == 0) { wires[t] = Fr::one(); return; }
    const uint32_t v = in[(t % B) * n_inputs + (row - 1)];      // inputs arrive instance-major [B][n_inputs]
    wires[t] = v == 0 ? Fr::zero() : (v == 1 ? Fr::one() : Fr::from_u32(v));
}

// The same from the request path's 320-byte records (request.h: build_raw / expand_raw are the host mirror, and what the u32 path of
// the benchmark entries is generated with): row r of instance b = one bit / byte / word of the record.
//   chacha20 (provers.go:155-173, utils/bytes.go:12-48): Counter[2][32] | Nonce[2][3][32] (LE words) | In[32][32] | Out[32][32] (BE words) |
//            Key[8][32] (LE words), bits LSB first;  record: counter[2] u32 LE | nonce[2][12] | in[128] | out[128] | key[32]
//   aes (provers.go:240-263): Nonce[5][12] | Counter[5] | In[80] | Out[80] | Key[16|32] byte values;
//            record: nonce[5][12] | counter[5] u32 LE | in[80] | out[80] | key
__device__ __forceinline__ uint32_t raw_le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ uint32_t raw_be32(const uint8_t *p) { return (uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24); }
__global__ __launch_bounds__(256) void k_assign_inputs_raw(const uint8_t *__restrict__ raw, int alg, Fr *__restrict__ wires, size_t n_inputs, size_t B) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (n_inputs + 1) * B) return;
    const size_t row = t / B, b = t % B;
    if (row == 0) { wires[t] = Fr::one(); return; }
    const uint8_t *r = raw + b * SOLVER_RAW_STRIDE;
    const uint32_t i = (uint32_t)row - 1;
    uint32_t v;
    if (alg == 0) {
        uint32_t word;
        if (i < 64) word = raw_le32(r + 4 * (i >> 5));
        else if (i < 256) word = raw_le32(r + 8 + 4 * ((i - 64) >> 5));              // nonce words: 12 * blk + 4 * k = 4 * (3 blk + k)
        else if (i < 1280) word = raw_be32(r + 32 + 4 * ((i - 256) >> 5));
        else if (i < 2304) word = raw_be32(r + 160 + 4 * ((i - 1280) >> 5));
        else word = raw_le32(r + 288 + 4 * ((i - 2304) >> 5));
        v = (word >> (i & 31)) & 1u;
    } else {
        if (i < 60) v = r[i];
        else if (i < 65) v = raw_le32(r + 60 + 4 * (i - 60));
        else if (i < 145) v = r[80 + (i - 65)];
        else if (i < 225) v = r[160 + (i - 145)];
        else v = r[240 + (i - 225)];
    }
    wires[t] = v == 0 ? Fr::zero() : (v == 1 ? Fr::one() : Fr::from_u32(v));
}

void solver_begin(const SolverProgram &p, SolverBuffers &buf, hipStream_t s) {
    const size_t B = buf.B, N = (size_t)1 << p.log_domain;
    // a and b: every row an R1C instruction owns is written by it; only the rows nobody writes must be zero again (the transforms
    // work in place) -- the padding up to the domain size when the program covers all constraints (it does for these circuits), all
    // rows otherwise.  c is scratch of the deferred divisions (written before read, level by level) and the destination of the
    // quotient-stage products: never cleared.  (The three full memsets were 6.4 / 12.9 GB of HBM writes per 1024-batch.)
    const size_t first_clear = p.constraints_covered ? p.n_constraints : 0;
    if (first_clear < N) {
        ZK_HIP(hipMemsetAsync(buf.a.p + first_clear * B, 0, (N - first_clear) * B * sizeof(Fr), s));
        ZK_HIP(hipMemsetAsync(buf.b.p + first_clear * B, 0, (N - first_clear) * B * sizeof(Fr), s));
    }
    ZK_HIP(hipMemsetAsync(buf.err.p, 0, B * sizeof(uint32_t), s));
    ZK_HIP(hipMemsetAsync(buf.sync.p, 0, 2 * sizeof(uint32_t), s));
    buf.sync_arrivals = 0;
    if (p.n_count_slots) {      // the counters of the countHints (every slot is counted into once per solve) and the tickets of k_solve_counts
        ZK_HIP(hipMemsetAsync(buf.count_scratch.p, 0, p.n_count_rows * p.n_count_slots * B * sizeof(uint32_t), s));
        ZK_HIP(hipMemsetAsync(buf.count_ticket.p, 0, p.n_count_slots * (B / 8 + 1) * sizeof(uint32_t), s));
    }
    if (buf.raw_alg >= 0) {
        const size_t want = buf.raw_alg == 0 ? 2560 : (buf.raw_alg == 1 ? 241 : 257);
        if (p.n_inputs != want) throw ZkError("solver: raw witness records do not match the circuit");
        hipLaunchKernelGGL(k_assign_inputs_raw, dim3(ceil_div((p.n_inputs + 1) * B, 256)), dim3(256), 0, s, buf.raw.p, buf.raw_alg, buf.wires.p, p.n_inputs, B);
    } else
        hipLaunchKernelGGL(k_assign_inputs, dim3(ceil_div((p.n_inputs + 1) * B, 256)), dim3(256), 0, s, buf.inputs.p, buf.wires.p, p.n_inputs, B);
    ZK_HIP(hipGetLastError());
}

#ifdef ZK_EXPERIMENTS
// batches up to this size take the persistent kernel (one workgroup per instance): ZKPROVE_SOLVER_PERSIST=n, default never
static size_t persist_max_b() {
    static const size_t v = [] { const char *e = getenv("ZKPROVE_SOLVER_PERSIST"); return e ? (size_t)atol(e) : (size_t)0; }();
    return v;
}
#endif
void solver_run_levels(const SolverProgram &p, SolverBuffers &buf, size_t first_level, size_t last_level, hipStream_t s) {
    SolveCtx cx{(ProgPtr)(uintptr_t)p.prog.p, (FrTab)(uintptr_t)p.coeffs.p, (FrTab)(uintptr_t)p.coeff_inv.p, (FrTab)(uintptr_t)p.lookup_tables.p, (FrTab)(uintptr_t)p.count_tables.p, buf.wires.p, buf.a.p, buf.b.p, buf.c_shared ? buf.c_shared : buf.c.p,
                buf.randomize.p, buf.challenge.p, buf.err.p, buf.count_scratch.p, buf.B, p.n_count_rows, buf.count_ticket.p, nullptr};
    if (last_level > p.levels.size()) last_level = p.levels.size();
#ifdef ZK_EXPERIMENTS
    if (buf.B <= persist_max_b() && first_level < last_level) {
        static const int nt_env = [] { const char *e = getenv("ZKPROVE_SOLVER_PERSIST_NT"); return e ? atoi(e) : 1024; }();
        static PerDeviceOnce attr;
        attr.run([] {
            ZK_HIP(hipFuncSetAttribute((const void *)k_solve_persistent<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 1024 * (int)sizeof(Fr)));
            ZK_HIP(hipFuncSetAttribute((const void *)k_solve_persistent<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 512 * (int)sizeof(Fr)));
            ZK_HIP(hipFuncSetAttribute((const void *)k_solve_persistent<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 256 * (int)sizeof(Fr)));
        });
        if (nt_env == 256)
            hipLaunchKernelGGL(k_solve_persistent<256>, dim3((unsigned)buf.B), dim3(256), 3 * 256 * sizeof(Fr), s, cx, p.instr_off.p, p.level_instr.p, p.level_desc.p,
                               (uint32_t)first_level, (uint32_t)last_level);
        else if (nt_env == 512)
            hipLaunchKernelGGL(k_solve_persistent<512>, dim3((unsigned)buf.B), dim3(512), 3 * 512 * sizeof(Fr), s, cx, p.instr_off.p, p.level_instr.p, p.level_desc.p,
                               (uint32_t)first_level, (uint32_t)last_level);
        else
            hipLaunchKernelGGL(k_solve_persistent<1024>, dim3((unsigned)buf.B), dim3(1024), 3 * 1024 * sizeof(Fr), s, cx, p.instr_off.p, p.level_instr.p, p.level_desc.p,
                               (uint32_t)first_level, (uint32_t)last_level);
        ZK_HIP(hipGetLastError());
        return;
    }
#endif
    // batches of at most LP_MAX_B proofs (single Prove() calls, small ProveBatch arrays): one workgroup per instance walks the levels by itself (k_solve_lanes)
    static const size_t lanes_max_b = [] { const char *e = ZK_AB_ENV("ZKPROVE_SOLVER_LANES"); return e ? (size_t)atol(e) : LP_MAX_B; }();
    // (the AES programs keep the level launches for now: their wide levels -- up to 100 wavefront tasks -- want the whole device, and
    // the lane program measured 15.5 against 13.4 ms per call there; ZKPROVE_SOLVER_LANES_ALL=1 in the experiments build forces it)
    static const bool lanes_all = [] { const char *e = ZK_AB_ENV("ZKPROVE_SOLVER_LANES_ALL"); return e && e[0] == '1'; }();
    if (buf.B <= lanes_max_b && first_level < last_level && (lanes_all || p.levels.size() <= 256)) {
#ifdef ZK_EXPERIMENTS
        static const bool dbg_on = [] { const char *e = getenv("ZKPROVE_SOLVER_DBG"); return e && e[0] == '1'; }();
        static DevBuf<unsigned long long> dbg_buf;
        static int dbg_calls = 0;
        if (dbg_on) {
            dbg_buf.ensure(p.levels.size() * 16 * 4);
            ZK_HIP(hipMemsetAsync(dbg_buf.p, 0, p.levels.size() * 16 * 4 * 8, s));
            cx.dbg = dbg_buf.p;
        }
#endif
        static const int nt_env = [] { const char *e = ZK_AB_ENV("ZKPROVE_SOLVER_LANES_NT"); return e ? atoi(e) : 1024; }();
        if (nt_env == 512)
            hipLaunchKernelGGL((k_solve_lanes<512, 8>), dim3((unsigned)buf.B), dim3(512), 0, s, cx, p.lp_shapes.p, p.lp_tasks.p, p.lp_soa.p, p.lp_level.p, p.level_desc.p,
                               p.div_list.p, (uint32_t)first_level, (uint32_t)last_level);
        else
            hipLaunchKernelGGL((k_solve_lanes<1024, 4>), dim3((unsigned)buf.B), dim3(1024), 0, s, cx, p.lp_shapes.p, p.lp_tasks.p, p.lp_soa.p, p.lp_level.p, p.level_desc.p,
                               p.div_list.p, (uint32_t)first_level, (uint32_t)last_level);
        ZK_HIP(hipGetLastError());
#ifdef ZK_EXPERIMENTS
        if (dbg_on && ++dbg_calls == 8) {       // the 8th call: warm; one line per level (how long its fastest / slowest wavefront worked)
            ZK_HIP(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(p.levels.size() * 64);
            ZK_HIP(hipMemcpy(h.data(), dbg_buf.p, h.size() * 8, hipMemcpyDeviceToHost));
            for (size_t l = first_level; l < last_level; l++) {
                unsigned long long t0 = ~0ull, t1max = 0, t2max = 0, t3 = 0, t1min = ~0ull;
                for (int w = 0; w < 16; w++) {
                    const unsigned long long *q = &h[(l * 16 + w) * 4];
                    if (!q[0]) continue;
                    t0 = std::min(t0, q[0]); t1max = std::max(t1max, q[1]); t1min = std::min(t1min, q[1]); t2max = std::max(t2max, q[2]); t3 = std::max(t3, q[3]);
                }
                fprintf(stderr, "lvl %zu n %u wide %u: tasks first done %.2f last done %.2f, barrier reached %.2f, left %.2f us\n", l, p.levels[l].count, p.levels[l].wide,
                        (t1min - t0) * 0.01, (t1max - t0) * 0.01, (t2max - t0) * 0.01, (t3 - t0) * 0.01);
            }
        }
#endif
        return;
    }
    static const bool pack = [] { const char *e = ZK_AB_ENV("ZKPROVE_SOLVER_PACK"); return !(e && e[0] == '0'); }();
    static const int occ_env = [] { const char *e = ZK_AB_ENV("ZKPROVE_SOLVER_OCC"); return e ? atoi(e) : 0; }();
    static const bool fuse_runs = [] { const char *e = ZK_AB_ENV("ZKPROVE_SOLVER_RUNS"); return !(e && e[0] == '0'); }();
    // one-call batches of the long programs (B <= 8: a single AES Prove(), a few coalesced ones): the levels between two narrow runs as
    // ONE launch each (k_solve_wide_run: grid barrier with a bounded poll); SolverBuffers::allow_wide_run: only where the caller can repeat the call after a timeout
    // MEASURED AND LOST (profiles/r05g_*): AES-256 Prove() 14.7 ms with it against 10.7 ms without, same box.  The premise was wrong: a wide level
    // of these programs holds up to ~5 000 instructions, which a launch spreads over the whole device in ~20 us; a barrier that cannot hang
    // needs every workgroup resident, and at the <= 96 workgroups that is safe for, each of them works through ~50 instructions per level
    // one after the other (at 768+ workgroups a barrier costs as much as the launch it replaces: MI355X_MICROARCH.md, barrier-counter).
    // Off unless ZKPROVE_SOLVER_WIDE_RUN=1 in the experiments build, where the parity and forced-timeout tests keep it bit-exact.
    static const bool wide_runs = [] { const char *e = ZK_AB_ENV("ZKPROVE_SOLVER_WIDE_RUN"); return e && e[0] == '1'; }();
    static const uint32_t spin_limit = [] { const char *e = ZK_AB_ENV("ZKPROVE_SOLVER_SPIN_LIMIT"); return e ? (uint32_t)atol(e) : WIDE_RUN_SPIN_LIMIT; }();
    const bool use_wide = wide_runs && fuse_runs && buf.B <= 8 && buf.allow_wide_run && buf.sync.n >= 2;
    auto narrow_run_at = [&](size_t l) { return p.levels[l].narrow && l + 1 < last_level && l + 1 < p.levels.size() && p.levels[l + 1].narrow; };
    for (size_t l = first_level; l < last_level && l < p.levels.size(); l++) {
        const SolverLevel &lv = p.levels[l];
        if (use_wide && !narrow_run_at(l)) {
            size_t e = l;
            uint32_t maxc = 1, barriers = 0;
            while (e < last_level && e < p.levels.size() && !narrow_run_at(e)) {
                maxc = std::max(maxc, p.levels[e].count);
                barriers += 1 + (p.levels[e].div_count ? 1 : 0);
                e++;
            }
            if (e - l >= 2) {
                const uint32_t G = std::min(maxc, WIDE_RUN_MAX_G);
                hipLaunchKernelGGL((k_solve_wide_run<8, 32>), dim3(G), dim3(8, 32), 0, s, cx, p.instr_off.p, p.level_instr.p, p.level_desc.p, p.div_list.p, (uint32_t)l, (uint32_t)e,
                                   buf.sync.p, buf.sync_arrivals, spin_limit);
                buf.sync_arrivals += G * barriers;
                l = e - 1;
                continue;
            }
        }
        if (!lv.count) continue;
        if (fuse_runs && lv.narrow) {
            // a run of narrow levels: one launch (a single narrow level takes the ordinary launch)
            size_t e = l;
            while (e < last_level && e < p.levels.size() && p.levels[e].narrow) e++;
            static const bool chains = [] { const char *e2 = getenv("ZKPROVE_SOLVER_CHAINS"); return !(e2 && e2[0] == '0'); }();      // A/B knob; tests run both
            const SolverProgram::ChainRun *cr = nullptr;
            if (chains)
                for (const auto &r : p.chain_runs)
                    if (r.l0 == l && r.l1 == e) cr = &r;
            if (cr) {
                hipLaunchKernelGGL(k_solve_chains, dim3(ceil_div(buf.B, 64)), dim3(64, cr->waves), 0, s, cx, p.chain_recs.p, p.chain_wave_first.p + cr->first, p.chain_consts.p);
                l = e - 1;
                continue;
            }
            if (e - l >= 2) {
                static const int sl_env = [] { const char *e2 = ZK_AB_ENV("ZKPROVE_SOLVER_RUN_WAVES"); return e2 ? atoi(e2) : 16; }();
                if (sl_env == 4)
                    hipLaunchKernelGGL((k_solve_narrow_run<64, 4>), dim3(ceil_div(buf.B, 64)), dim3(64, 4), 0, s, cx, p.instr_off.p, p.level_instr.p, p.level_desc.p, p.div_list.p,
                                       (uint32_t)l, (uint32_t)e);
                else
                    hipLaunchKernelGGL((k_solve_narrow_run<64, 16>), dim3(ceil_div(buf.B, 64)), dim3(64, 16), 0, s, cx, p.instr_off.p, p.level_instr.p, p.level_desc.p, p.div_list.p,
                                       (uint32_t)l, (uint32_t)e);
                l = e - 1;
                continue;
            }
        }
        // the level's countHints first, G workgroups each (k_solve_counts); the level kernels below take the rest of the level
        static const uint32_t count_parts = [] { const char *e2 = getenv("ZKPROVE_SOLVER_COUNT_PARTS"); return e2 ? (uint32_t)atoi(e2) : 16u; }();      // 1: inside the level kernels, as before
        uint32_t skip = 0;
        if (count_parts > 1 && lv.n_count && (size_t)lv.n_count * count_parts <= 65535) {
            const dim3 gy(1, lv.n_count * count_parts);
            if (buf.B <= 8) hipLaunchKernelGGL((k_solve_counts<8, 32>), dim3(ceil_div(buf.B, 8), gy.y), dim3(8, 32), 0, s, cx, p.instr_off.p, p.level_instr.p + lv.first, count_parts);
            else if (buf.B <= 16) hipLaunchKernelGGL((k_solve_counts<16, 16>), dim3(ceil_div(buf.B, 16), gy.y), dim3(16, 16), 0, s, cx, p.instr_off.p, p.level_instr.p + lv.first, count_parts);
            else if (buf.B <= 128) hipLaunchKernelGGL((k_solve_counts<64, 16>), dim3(ceil_div(buf.B, 64), gy.y), dim3(64, 16), 0, s, cx, p.instr_off.p, p.level_instr.p + lv.first, count_parts);
            else hipLaunchKernelGGL((k_solve_counts<64, 4>), dim3(ceil_div(buf.B, 64), gy.y), dim3(64, 4), 0, s, cx, p.instr_off.p, p.level_instr.p + lv.first, count_parts);
            skip = lv.n_count;
        }
        const uint32_t l_first = lv.first + skip, l_count = lv.count - skip, l_wide = lv.wide - skip;
        const uint32_t packed_rows = l_wide + (l_count - l_wide + 3) / 4;
        const bool four = occ_env ? occ_env >= 4 : p.levels.size() <= 256;
        const uint32_t packed_rows16 = l_wide + (l_count - l_wide + 15) / 16;
        if (!l_count) {
            // (a level of countHints only)
        } else if (pack && !four && buf.B > 16 && buf.B <= 128 && packed_rows16 <= 65535) {
            // mid-size batches of the long programs (AES): one or two instance tiles cannot hide a level's longest instruction -- the ~3 600
            // queries of a countHint, a 130-term row -- behind other workgroups, so the instruction gets 16 slices instead of 4 (and the
            // short instructions sixteen to a workgroup).  AES-256, 64 proofs: the 118 wide levels 6.1 -> ... ms (profiles/r05*)
            hipLaunchKernelGGL((k_solve_level<64, 16, 1>), dim3(ceil_div(buf.B, 64), packed_rows16), dim3(64, 16), 0, s, cx, p.instr_off.p, p.level_instr.p + l_first, l_count, l_wide);
        } else if (pack && buf.B > 16 && packed_rows <= 65535) {
            if (four) hipLaunchKernelGGL((k_solve_level<64, 4, 4>), dim3(ceil_div(buf.B, 64), packed_rows), dim3(64, 4), 0, s, cx, p.instr_off.p, p.level_instr.p + l_first, l_count, l_wide);
            else hipLaunchKernelGGL((k_solve_level<64, 4, 1>), dim3(ceil_div(buf.B, 64), packed_rows), dim3(64, 4), 0, s, cx, p.instr_off.p, p.level_instr.p + l_first, l_count, l_wide);
        } else {
            for (uint32_t off = 0; off < l_count; off += 65535) {       // grid.y limit; every row cooperative (correct for any instruction)
                const uint32_t cnt = l_count - off < 65535 ? l_count - off : 65535;
                if (buf.B <= 8)
                    hipLaunchKernelGGL((k_solve_level<8, 32>), dim3(ceil_div(buf.B, 8), cnt), dim3(8, 32), 0, s, cx, p.instr_off.p, p.level_instr.p + l_first + off, cnt, cnt);
                else if (buf.B <= 16)
                    hipLaunchKernelGGL((k_solve_level<16, 16>), dim3(ceil_div(buf.B, 16), cnt), dim3(16, 16), 0, s, cx, p.instr_off.p, p.level_instr.p + l_first + off, cnt, cnt);
                else
                    hipLaunchKernelGGL((k_solve_level<64, 4>), dim3(ceil_div(buf.B, 64), cnt), dim3(64, 4), 0, s, cx, p.instr_off.p, p.level_instr.p + l_first + off, cnt, cnt);
            }
        }
        if (lv.div_count) {
            if (buf.B <= 16)
                hipLaunchKernelGGL(k_solve_divisions<2>, dim3(ceil_div(buf.B, 64), ceil_div(lv.div_count, 2)), dim3(64), 0, s, cx, p.div_list.p + 4 * (size_t)lv.div_first,
                                   lv.div_count);
            else
                hipLaunchKernelGGL(k_solve_divisions<16>, dim3(ceil_div(buf.B, 256), ceil_div(lv.div_count, 16)), dim3(256), 0, s, cx,
                                   p.div_list.p + 4 * (size_t)lv.div_first, lv.div_count);
        }
    }
    ZK_HIP(hipGetLastError());
}

}  // namespace zk
