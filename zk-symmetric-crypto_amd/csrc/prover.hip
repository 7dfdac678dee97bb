// Groth16 batch pipeline on one MI355X (see prover.h).
#include "prover.h"
#include "fixedbase.cuh"
#include "xmd.cuh"
#include "glv.cuh"
#include <string.h>
#include <algorithm>

namespace zk {

static constexpr int C_WIRE = 13;   // window bits for the wire-scalar MSMs (few wide scalars)
static constexpr int C_H = 16;      // window bits for the quotient MSM (all scalars full width)
// Up to this batch size the MSM chains run side by side on four streams (their kernels are latency-bound): Algorithm::side_max_b() -- 64 proofs,
// and 128 for the programs with a commitment (AES): their six wire MSMs end in ~15 fold chains of 1.4-3.8 ms, which the one-stream path of
// larger batches runs one after the other (65 proofs: 47.7 ms against 31.3 for 64, profiles/r05za_*); only while the wire combs exist
static constexpr size_t SIDE_G2_MAX_B = 64;
static constexpr size_t SIDE_G2_MAX_B_COMMIT = 128;
// Once the comb tables exist (a batch of >= comb_min_batch() proofs has been served) every batch from this size takes them: the walk's
// lanes are (plane, proof) pairs, so 16 proofs fill wavefronts with four planes each.  Below, the latency-tuned bucket pipeline of the
// one-proof calls (virtual instances, side streams) stays faster.
static constexpr size_t COMB_SMALL_MIN_B = 16;

// ------------------------------------------------------------------------------- pk parsing
namespace {
// Every failure names the section of gnark's ProvingKey.WriteTo layout it happened in and the byte offset, so that the first real
// pk.* that does not parse (the layout is restated from the published gnark source) gives an actionable message.
struct PkCursor {
    const uint8_t *p;
    size_t len, pos = 0;
    const char *section = "domain header";
    void need(size_t n) const {
        if (pos + n > len)
            throw ZkError(std::string("proving key: truncated in section '") + section + "' at offset " + std::to_string(pos) + " (needs " + std::to_string(n) +
                          " more bytes, file has " + std::to_string(len) + ")");
    }
    uint32_t u32be() { need(4); uint32_t v = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3]; pos += 4; return v; }
    uint64_t u64be() { uint64_t hi = u32be(); return (hi << 32) | u32be(); }
};
struct Slice { size_t off = 0, n = 0; bool raw = false; const char *name = ""; };

// one point: size from the flag bits of its first byte (gnark-crypto marshal.go)
static Slice take_points(PkCursor &c, size_t n, size_t comp_size, const char *name) {
    c.section = name;
    Slice s; s.off = c.pos; s.n = n; s.name = name;
    if (n == 0) return s;
    c.need(1);
    s.raw = (c.p[c.pos] & 0xC0) == 0;
    const size_t sz = s.raw ? 2 * comp_size : comp_size;
    c.need(n * sz);
    for (size_t i = 0; i < n; i++) {
        const bool raw_i = (c.p[c.pos + i * sz] & 0xC0) == 0;
        if (raw_i != s.raw)
            throw ZkError(std::string("proving key: section '") + name + "': point " + std::to_string(i) + " at offset " + std::to_string(c.pos + i * sz) +
                          " switches between compressed and uncompressed encoding");
    }
    c.pos += n * sz;
    return s;
}
static Slice take_slice(PkCursor &c, size_t comp_size, const char *name) { c.section = name; uint32_t n = c.u32be(); return take_points(c, n, comp_size, name); }

static void upload_g1(const uint8_t *base, const Slice &s, DevBuf<G1Affine> &out, hipStream_t st) {
    out.alloc(s.n ? s.n : 1);
    if (!s.n) return;
    const size_t bytes = s.n * (s.raw ? 64 : 32);
    DevBuf<uint8_t> raw(bytes);
    DevBuf<uint32_t> bad(1); bad.zero(st);
    ZK_HIP(hipMemcpyAsync(raw.p, base + s.off, bytes, hipMemcpyHostToDevice, st));
    if (s.raw) g1_from_be_bytes(out.p, raw.p, s.n, st);
    else g1_decompress(out.p, raw.p, s.n, bad.p, st);
    g1_validate(out.p, s.raw ? raw.p : nullptr, s.n, bad.p, st);
    uint32_t nb = 0; bad.download(&nb, 1, st);
    ZK_HIP(hipStreamSynchronize(st));
    if (nb)
        throw ZkError(std::string("proving key: section '") + s.name + "' (offset " + std::to_string(s.off) + ", " + std::to_string(s.n) + (s.raw ? " uncompressed" : " compressed") +
                      " G1 points): " + std::to_string(nb) + " invalid G1 point(s) (non-reduced coordinate, bad flag bits or not on the curve)");
}
static void upload_g2(const uint8_t *base, const Slice &s, DevBuf<G2Affine> &out, hipStream_t st) {
    out.alloc(s.n ? s.n : 1);
    if (!s.n) return;
    const size_t bytes = s.n * (s.raw ? 128 : 64);
    DevBuf<uint8_t> raw(bytes);
    DevBuf<uint32_t> bad(1); bad.zero(st);
    ZK_HIP(hipMemcpyAsync(raw.p, base + s.off, bytes, hipMemcpyHostToDevice, st));
    if (s.raw) g2_from_be_bytes(out.p, raw.p, s.n, st);
    else g2_decompress(out.p, raw.p, s.n, bad.p, st);
    g2_validate(out.p, s.raw ? raw.p : nullptr, s.n, bad.p, st);
    uint32_t nb = 0; bad.download(&nb, 1, st);
    ZK_HIP(hipStreamSynchronize(st));
    if (nb)
        throw ZkError(std::string("proving key: section '") + s.name + "' (offset " + std::to_string(s.off) + ", " + std::to_string(s.n) + (s.raw ? " uncompressed" : " compressed") +
                      " G2 points): " + std::to_string(nb) + " invalid G2 point(s) (not on the curve or not in the subgroup)");
}
static std::vector<uint8_t> take_bools(PkCursor &c, const char *name) {
    c.section = name;
    // gnark-crypto encoder []bool: u32 length, bits packed LSB-first  [gnark-mem, unpinned: SURVEY App. C]
    uint32_t n = c.u32be();
    size_t nbytes = (n + 7) / 8;
    c.need(nbytes);
    std::vector<uint8_t> out(n);
    for (uint32_t i = 0; i < n; i++) out[i] = (c.p[c.pos + i / 8] >> (i % 8)) & 1;
    c.pos += nbytes;
    return out;
}
static uint32_t bitrev32(uint32_t v, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) if (v & (1u << i)) r |= 1u << (bits - 1 - i);
    return r;
}
}  // namespace

// ---------------------------------------------------------------------------- device state
// Everything one batch's SOLVE stage writes and its MSM/NTT stage reads; two of them allow the (latency-bound) solver
// of batch k+1 to run on a second stream under the (ALU-bound) NTT + MSM stage of batch k.
struct SolveSet {
    SolverBuffers sb;
    DevBuf<G1XYZZ> m_com, m_pok;
    DevBuf<G1Affine> com_aff, pok_aff;
    DevBuf<U256> chal;
    DevBuf<U256> r, s, rnd;            // blinding scalars and the Randomize hint value of this batch (canonical)
    DevBuf<ProofPoints> out;           // proofs of this batch
    DevBuf<WideInput> wide;            // assignment elements wider than 32 bits (ZkpProveWitness; none for the request path)
    size_t n_wide = 0;
    std::vector<G1Affine> h_com;
    std::vector<U256> h_chal;
    hipEvent_t solved = nullptr, consumed = nullptr, uploaded = nullptr, t0 = nullptr, t1 = nullptr, t2 = nullptr, t3 = nullptr;
    hipEvent_t sorted = nullptr;       // ahead mode: the transforms and the quotient-MSM sort of this batch are complete
    hipEvent_t zdone = nullptr;        // the quotient MSM of this batch has read its sorted list (the slot may be refilled)
    bool ahead = false;                // ... were enqueued by enqueue_ntt_sort (enqueue_main then starts at the wire MSMs)
    int zslot = 0;                     // which copy of the sorted list / of the comb patterns they filled
    bool pat_pending = false;          // one pattern slot: the transforms are enqueued, the patterns wait for the walk in front (enqueue_patterns_late)
    bool comb = false;                 // ... as comb patterns (msm_comb.hip) rather than a sorted bucket list
    bool ready = false;
    void reserve(const SolverProgram &prog, size_t B, size_t c_B = (size_t)-1) {
        sb.reserve(prog, B, c_B);
        m_com.ensure(B); m_pok.ensure(B); com_aff.ensure(B); pok_aff.ensure(B); chal.ensure(B);
        r.ensure(B); s.ensure(B); rnd.ensure(B); out.ensure(B);
        h_com.resize(B); h_chal.resize(B);
        if (!solved) {
            ZK_HIP(hipEventCreate(&solved)); ZK_HIP(hipEventCreate(&consumed)); ZK_HIP(hipEventCreate(&uploaded));
            ZK_HIP(hipEventCreate(&t0)); ZK_HIP(hipEventCreate(&t1)); ZK_HIP(hipEventCreate(&t2)); ZK_HIP(hipEventCreate(&t3));
            ZK_HIP(hipEventCreate(&sorted)); ZK_HIP(hipEventCreate(&zdone));
        }
        ready = true;
    }
};

struct Algorithm::Resident {
    size_t B = 0, cap = 0;
    SolveSet set[3];                // two working sets for the pipelined entries, a third when batches are prepared two ahead (resident_prove_pipelined)
    MsmScratch ms_b2;               // the G2 MSM of LARGE batches on the copy stream, beside the G1 MSMs of the main stream (enqueue_main, ahead mode)
    MsmScratch ms, ms2, ms_g2, ms_k, ms_a;      // ms2: the commitment MSMs of the solve stage when it runs on the second stream; ms_g2: the G2 MSM of
                                    // SMALL batches, which runs on its own stream beside the latency-bound G1 MSMs; ms_k, ms_a: likewise the K and the A / B1 MSMs
    DevBuf<G1XYZZ> m_ar, m_bs1, m_k, m_z;
    DevBuf<G2XYZZ> m_bs2;
    DevBuf<G1Affine> ar_aff, bs1_aff;
    DevBuf<G1XYZZ> f_t, f_u, f_v;
    CombScratch comb;               // bit-plane patterns (two copies) and range accumulators of the comb quotient MSM
    size_t comb_cap = 0, dense_cap = 0;      // batch sizes the comb scratch / the dense bucket-sort buffers are reserved for
    size_t side_cap = 0;                     // ... and the side scratches ms_a / ms_k / ms_g2 (at most side_max_b() instances)
    bool used_comb = false;         // the last enqueue_main took the comb path (debug_fetch then forms the products f itself)
    int last_set = 0;               // the set that holds the most recent batch
    hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr, copy_stream = nullptr;
    bool last_g2_side = false;  // ... with the G2 MSM on the copy stream (phase times: msm_g1 is then one uninterrupted main-stream span)
    bool last_ahead = false;    // the last enqueue_main ran in ahead mode (phase times: the transforms are bracketed by ev[19], ev[20])
    hipEvent_t ev[23] = {};     // [21], [22]: split fold of a mid-size batch (comb_msm_split)     // [19], [20] transforms in ahead mode  [17] products f ready  [18] dense list sorted     // [10] MSM sums of the current batch complete  [11] tail of the previous batch done  [12] A and B1 sums complete  [13] end of the G1 MSMs
};

Algorithm::Algorithm() {}
Algorithm::~Algorithm() {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    (void)hipSetDevice(device);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    if (res) {
        for (auto &e : res->ev) if (e) (void)hipEventDestroy(e);
        for (auto &st : res->set) for (hipEvent_t e : {st.solved, st.consumed, st.uploaded, st.t0, st.t1, st.t2, st.t3, st.sorted, st.zdone}) if (e) (void)hipEventDestroy(e);
        if (res->stream) (void)hipStreamDestroy(res->stream);
        if (res->stream2) (void)hipStreamDestroy(res->stream2);
        if (res->stream3) (void)hipStreamDestroy(res->stream3);
        if (res->copy_stream) (void)hipStreamDestroy(res->copy_stream);
    }
}

void Algorithm::load(const uint8_t *pkb, size_t pk_len, const uint8_t *r1cs_bytes, size_t r1cs_len) {
    DeviceScope on_device(device);
    r1cs_parse(r1cs_bytes, r1cs_len, cs);
    hipStream_t st = 0;
    prog.build(cs, st);
    has_commit = prog.has_commit;
    commit_level = 0;
    for (size_t l = 0; l < prog.levels.size(); l++) if (prog.levels[l].has_commit) commit_level = l;
    const int logn = cs.log_domain();
    dom.init(logn, st);
    static const bool prio_off = [] { const char *e = getenv("ZKPROVE_NTT_PRIO"); return e && e[0] == '0'; }();      // A/B knob
    static const bool prio_all = [] { const char *e = getenv("ZKPROVE_NTT_PRIO"); return e && e[0] == '2'; }();      // A/B knob: every program
    dom.hi_prio = prio_all || (!prio_off && prog.levels.size() <= 256);      // short programs: the copy stream's chain transforms -> patterns -> walk bounds the window between two walks (ntt.hip, ZK_NTT_PRIO)
    const size_t N = (size_t)1 << logn, nw = cs.nb_wires();

    PkCursor c{pkb, pk_len};
    pk.domain = c.u64be();
    if (pk.domain != N)
        throw ZkError("proving key: domain cardinality " + std::to_string(pk.domain) + " at offset 0 does not match the constraint system (" + std::to_string(N) + ")");
    c.need(5 * 32 + 1);
    c.pos += 5 * 32 + 1;                                  // CardinalityInv, Generator, GeneratorInv, FrMultiplicativeGen(+Inv), withPrecompute
    Slice misc1 = take_points(c, 3, 32, "G1.Alpha,Beta,Delta");
    Slice sA = take_slice(c, 32, "G1.A"), sB = take_slice(c, 32, "G1.B"), sZ = take_slice(c, 32, "G1.Z"), sK = take_slice(c, 32, "G1.K");
    Slice misc2 = take_points(c, 2, 64, "G2.Beta,Delta");
    Slice sB2 = take_slice(c, 64, "G2.B");
    c.section = "NbWires / NbInfinityA / NbInfinityB";
    uint64_t nwires = c.u64be(), ninfA = c.u64be(), ninfB = c.u64be();
    std::vector<uint8_t> infA = take_bools(c, "InfinityA"), infB = take_bools(c, "InfinityB");
    c.section = "CommitmentKeys";
    uint32_t nkeys = c.u32be();
    Slice sPed, sPedS;
    if (nkeys > 1) throw ZkError("proving key: " + std::to_string(nkeys) + " commitment keys at offset " + std::to_string(c.pos - 4) + " (the circuits have at most one)");
    if (nkeys == 1) { sPed = take_slice(c, 32, "CommitmentKeys[0].Basis"); sPedS = take_slice(c, 32, "CommitmentKeys[0].BasisExpSigma"); }
    if (c.pos != pk_len) throw ZkError("proving key: " + std::to_string(pk_len - c.pos) + " trailing bytes after the last section (offset " + std::to_string(c.pos) + ")");
    if (nwires != nw || infA.size() != nw || infB.size() != nw)
        throw ZkError("proving key: NbWires " + std::to_string(nwires) + " / len(InfinityA) " + std::to_string(infA.size()) + " / len(InfinityB) " + std::to_string(infB.size()) +
                      " do not match the constraint system's " + std::to_string(nw) + " wires");
    if ((nkeys == 1) != has_commit) throw ZkError("proving key: commitment keys do not match the constraint system");
    pk.n_wires = nw;

    // scalar rows
    std::vector<uint32_t> rowsA, rowsB, rowsK, rowsZ, rowsPed;
    std::vector<uint8_t> skip(nw, 0);
    for (uint32_t w : prog.private_committed) skip[w] = 1;
    if (has_commit) skip[prog.commit_wire] = 1;
    for (size_t w = 0; w < nw; w++) {
        if (!infA[w]) rowsA.push_back((uint32_t)w);
        if (!infB[w]) rowsB.push_back((uint32_t)w);
        if (w >= cs.nb_public && !skip[w]) rowsK.push_back((uint32_t)w);
    }
    if (rowsA.size() != sA.n || rowsB.size() != sB.n || rowsB.size() != sB2.n || rowsK.size() != sK.n || ninfA != nw - sA.n || ninfB != nw - sB.n)
        throw ZkError("proving key: point counts do not match the constraint system (G1.A " + std::to_string(sA.n) + " vs " + std::to_string(rowsA.size()) + " finite, G1.B " +
                      std::to_string(sB.n) + " / G2.B " + std::to_string(sB2.n) + " vs " + std::to_string(rowsB.size()) + ", G1.K " + std::to_string(sK.n) + " vs " +
                      std::to_string(rowsK.size()) + ", NbInfinityA " + std::to_string(ninfA) + ", NbInfinityB " + std::to_string(ninfB) + ")");
    if (sZ.n + 1 < N) throw ZkError("proving key: too few Z points");
    sZ.n = N - 1;
    if (has_commit) {
        if (sPed.n != prog.private_committed.size() || sPedS.n != sPed.n) throw ZkError("proving key: pedersen basis size mismatch");
        rowsPed = prog.private_committed;
    }
    DevBuf<G1Affine> pts;
    upload_g1(pkb, misc1, pk.g1_misc, st);
    upload_g2(pkb, misc2, pk.g2_misc, st);
    pk.delta_tab.alloc(32 * 255); pk.delta2_tab.alloc(32 * 255);
    hipLaunchKernelGGL(k_fixed_table<Fp>, dim3(ceil_div(32 * 255, 64)), dim3(64), 0, st, pk.g1_misc.p + 2, pk.delta_tab.p);
    hipLaunchKernelGGL(k_fixed_table<Fp2>, dim3(ceil_div(32 * 255, 64)), dim3(64), 0, st, pk.g2_misc.p + 1, pk.delta2_tab.p);
    // split mode: slice [lo, hi) of a base set of n points for this rank (whole set when split_world == 1)
    auto slice = [&](size_t n, size_t &lo, size_t &hi) {
        const size_t base = n / (size_t)split_world, rem = n % (size_t)split_world, r = (size_t)split_rank;
        lo = r * base + (r < rem ? r : rem);
        hi = lo + base + (r < rem ? 1 : 0);
    };
    size_t lo, hi;
    upload_g1(pkb, sA, pts, st); slice(sA.n, lo, hi); msm_bases_build(pk.A, pts.p + lo, hi - lo, rowsA.data() + lo, C_WIRE, st);
    upload_g1(pkb, sB, pts, st); slice(sB.n, lo, hi); msm_bases_build(pk.B1, pts.p + lo, hi - lo, rowsB.data() + lo, C_WIRE, st);
    {   // quotient bases in evaluation form (keyfold.hip): Z becomes one base per coset point (dense MSM over A*B on the coset),
        // and the part of the quotient that is linear in the wires (through c = O w) moves into the bases of K.
        upload_g1(pkb, sZ, pts, st);
        const size_t ncons = cs.nb_constraints;
        DevBuf<G1Affine> ze(N), zc(ncons);
        fold_quotient_bases(pts.p, dom, ncons, ze.p, zc.p, st);
        slice(N, lo, hi);
        msm_bases_build(pk.Z, ze.p + lo, hi - lo, nullptr, C_H, st);
        pk.Z.dense_stride = N;
        z_offset = lo;
        pk.ze.alloc(hi - lo);
        ZK_HIP(hipMemcpyAsync(pk.ze.p, ze.p + lo, (hi - lo) * sizeof(G1Affine), hipMemcpyDeviceToDevice, st));
        std::vector<OutputTerm> terms;
        for (size_t i = 0; i < cs.nb_instructions(); i++) {
            if (cs.bp_kind[cs.blueprint_id[i]] != BP_R1C) continue;
            const uint32_t *cd = cs.calldata.data() + cs.start_calldata[i];
            const uint32_t *p = cd + 4 + 2 * (size_t)(cd[1] + cd[2]);
            for (uint32_t k = 0; k < cd[3]; k++, p += 2) terms.push_back(OutputTerm{p[1] == WIRE_CONST ? 0u : p[1], cs.constraint_offset[i], p[0]});
        }
        DevBuf<Fr> coeffs(cs.coeffs.size() + 1);
        ZK_HIP(hipMemcpy(coeffs.p, cs.coeffs.data(), cs.coeffs.size() * sizeof(Fr), hipMemcpyHostToDevice));
        upload_g1(pkb, sK, pts, st);
        DevBuf<G1Affine> kf;
        std::vector<uint32_t> rowsKf;
        fold_output_matrix(zc.p, pts.p, rowsK, std::move(terms), coeffs.p, kf, rowsKf, st);
        slice(rowsKf.size(), lo, hi);
        msm_bases_build(pk.K, kf.p + lo, hi - lo, rowsKf.data() + lo, C_WIRE, st);
    }
    pk.n_ped = 0;
    if (has_commit) {
        upload_g1(pkb, sPed, pts, st); msm_bases_build(pk.ped, pts.p, sPed.n, rowsPed.data(), C_WIRE, st);
        upload_g1(pkb, sPedS, pts, st); msm_bases_build(pk.ped_sigma, pts.p, sPedS.n, rowsPed.data(), C_WIRE, st);
        pk.n_ped = sPed.n;
    }
    DevBuf<G2Affine> pts2;
    upload_g2(pkb, sB2, pts2, st); slice(sB2.n, lo, hi); msm_bases_build(pk.B2, pts2.p + lo, hi - lo, rowsB.data() + lo, C_WIRE, st);
    ZK_HIP(hipDeviceSynchronize());
}

// ------------------------------------------------------------------------------ finalize
__global__ __launch_bounds__(64) void k_u256_to_mont(const U256 *in, Fr *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v;
#pragma unroll
    for (int k = 0; k < 8; k++) v.l[k] = in[i].l[k];
    out[i] = v.to_mont();
}
// wires row (index + 1) of instance `instance` <- value (the u32 assignment of solver_begin wrote a placeholder there)
__global__ __launch_bounds__(64) void k_wide_inputs(const WideInput *__restrict__ wide, size_t n, Fr *__restrict__ wires, size_t B) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v;
#pragma unroll
    for (int k = 0; k < 8; k++) v.l[k] = wide[i].value.l[k];
    wires[((size_t)wide[i].index + 1) * B + wide[i].instance] = v.to_mont();
}
__device__ __forceinline__ void store_fp(U256 &dst, const Fp &v) {
#pragma unroll
    for (int k = 0; k < 8; k++) dst.l[k] = v.l[k];
}
// Stage 1 (blockIdx.y = task, lanes = instances): 0: Ar = A + alpha + r*delta   1: Bs1 = B + beta + s*delta   2: T = (-rs)*delta
__global__ __launch_bounds__(64) void k_finalize_stage1(const G1XYZZ *__restrict__ m_ar, const G1XYZZ *__restrict__ m_bs1, const G1Affine *__restrict__ misc,
                                                        const G1Affine *__restrict__ dtab, const U256 *__restrict__ r, const U256 *__restrict__ s,
                                                        G1Affine *__restrict__ ar_aff, G1Affine *__restrict__ bs1_aff, G1XYZZ *__restrict__ t_out, size_t B) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int task = blockIdx.y;
    Fr rm, sm;
#pragma unroll
    for (int k = 0; k < 8; k++) { rm.l[k] = r[b].l[k]; sm.l[k] = s[b].l[k]; }
    if (task == 0) {
        G1XYZZ ar = m_ar[b];
        ar.madd(misc[0]);
        ar.add(fixed_mul(dtab, rm.l));
        ar_aff[b] = ar.to_affine();
    } else if (task == 1) {
        G1XYZZ bs1 = m_bs1[b];
        bs1.madd(misc[1]);
        bs1.add(fixed_mul(dtab, sm.l));
        bs1_aff[b] = bs1.to_affine();
    } else {
        const Fr kr = (rm.to_mont() * sm.to_mont()).neg().from_mont();
        t_out[b] = fixed_mul(dtab, kr.l);
    }
}
// Stage 1 for small batches: one 16-lane workgroup per (instance, task), the fixed-base product as a tree (fixedbase.cuh)
__global__ __launch_bounds__(16) void k_finalize_stage1_small(const G1XYZZ *__restrict__ m_ar, const G1XYZZ *__restrict__ m_bs1, const G1Affine *__restrict__ misc,
                                                              const G1Affine *__restrict__ dtab, const U256 *__restrict__ r, const U256 *__restrict__ s,
                                                              G1Affine *__restrict__ ar_aff, G1Affine *__restrict__ bs1_aff, G1XYZZ *__restrict__ t_out, size_t B) {
    __shared__ G1XYZZ lds[16];
    ZK_FOLD_SETPRIO();
    const size_t b = blockIdx.x;
    const int task = blockIdx.y, lane = threadIdx.x;
    Fr rm, sm;
#pragma unroll
    for (int k = 0; k < 8; k++) { rm.l[k] = r[b].l[k]; sm.l[k] = s[b].l[k]; }
    Fr kk = task == 0 ? rm : sm;
    if (task == 2) kk = (rm.to_mont() * sm.to_mont()).neg().from_mont();
    const G1XYZZ f = fixed_mul_tree(dtab, kk.l, lane, lds);
    if (lane != 0) return;
    if (task == 2) { t_out[b] = f; return; }
    G1XYZZ acc = task == 0 ? m_ar[b] : m_bs1[b];
    acc.madd(misc[task]);
    acc.add(f);
    if (task == 0) ar_aff[b] = acc.to_affine();
    else bs1_aff[b] = acc.to_affine();
}
// Bs for small batches (the tail of the G2 chain, the longest chain of a one-proof AES call)
__global__ __launch_bounds__(16) void k_finalize_g2_small(const G2XYZZ *__restrict__ m_bs2, const G2Affine *__restrict__ misc, const G2Affine *__restrict__ dtab,
                                                          const U256 *__restrict__ s, ProofPoints *__restrict__ out, size_t B) {
    __shared__ G2XYZZ lds[16];
    ZK_FOLD_SETPRIO();
    const size_t b = blockIdx.x;
    const int lane = threadIdx.x;
    uint32_t sl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) sl[k] = s[b].l[k];
    const G2XYZZ f = fixed_mul_tree(dtab, sl, lane, lds);
    if (lane != 0) return;
    G2XYZZ bs = m_bs2[b];
    bs.madd(misc[0]);
    bs.add(f);
    const G2Affine a = bs.to_affine();
    store_fp(out[b].bs[0], a.x.a0.from_mont()); store_fp(out[b].bs[1], a.x.a1.from_mont());
    store_fp(out[b].bs[2], a.y.a0.from_mont()); store_fp(out[b].bs[3], a.y.a1.from_mont());
}
// Stage 2 (blockIdx.y = task): 0: U = s*Ar   1: V = r*Bs1   (variable base)
__global__ __launch_bounds__(64) void k_finalize_stage2(const G1Affine *__restrict__ ar_aff, const G1Affine *__restrict__ bs1_aff, const U256 *__restrict__ r,
                                                        const U256 *__restrict__ s, G1XYZZ *__restrict__ u_out, G1XYZZ *__restrict__ v_out, size_t B) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // through the curve's endomorphism (glv.cuh): one chain of 127 doublings instead of 254 -- these two multiplications are a
    // dependent chain on one thread each, i.e. latency of every Prove() call whatever the batch size
    uint32_t k[8];
    if (blockIdx.y == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = s[b].l[i];
        u_out[b] = scalar_mul_glv(ar_aff[b], k);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = r[b].l[i];
        v_out[b] = scalar_mul_glv(bs1_aff[b], k);
    }
}
// The same two multiplications for batches of at most 128 proofs, 32 to a workgroup (single Prove() calls: this chain IS the end of the call's critical
// path), as a pipeline inside one workgroup per multiplication.  k P = k1 P + k2 phi(P) right to left: wavefront 0 runs the ONE chain
// that cannot be shortened -- the doublings D_i = 2^i P, i < 128 -- and hands every D_i over through a ring in LDS; four more wavefronts
// add the D_i whose bit is set (two per half-scalar, even and odd positions: an addition takes ~1.5 doublings and half the bits are
// set, so each keeps up with the doubler), and three additions join the four partial sums with the endomorphism and the signs at the
// end.  128 doublings + 2 additions on the critical path instead of 128 doublings + ~96 additions.  Lanes = instances (lockstep).
// Same group element as scalar_mul_glv, hence the same proof bytes.
template <int RING>
__global__ __launch_bounds__(320) void k_finalize_stage2_pipe(const G1Affine *__restrict__ ar_aff, const G1Affine *__restrict__ bs1_aff, const U256 *__restrict__ r,
                                                             const U256 *__restrict__ s, G1XYZZ *__restrict__ u_out, G1XYZZ *__restrict__ v_out, size_t B) {
    static_assert(RING % 2 == 0, "even and odd positions keep their slots");
    __shared__ G1XYZZ ring[RING][32];      // lanes 0..31 hold instances (64 KB of static LDS would not fit both arrays at 64)
    __shared__ G1XYZZ part[4][32];
    ZK_FOLD_SETPRIO();
    __shared__ volatile int produced, consumed[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t b_raw = (size_t)blockIdx.x * 32 + lane;
    const bool live = lane < 32 && b_raw < B;
    const size_t b = live ? b_raw : B - 1;
    const int ll = lane & 31;
    if (threadIdx.x < 4) consumed[threadIdx.x] = 0;
    if (threadIdx.x == 4) produced = 0;
    __syncthreads();
    const G1Affine P = blockIdx.y == 0 ? ar_aff[b] : bs1_aff[b];
    uint32_t k[8], k1[4], k2[4];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = blockIdx.y == 0 ? s[b].l[i] : r[b].l[i];
    bool n1, n2;
    const bool split = glv::decompose(k, k1, n1, k2, n2);      // (always true; an instance it failed for takes the plain ladder below)
    if (wave == 0) {
        G1XYZZ D = G1XYZZ::from_affine(P);
        for (int i = 0; i < 128; i++) {
            if (i >= RING) {
                const int a0 = i & 1;                          // the adders of this slot's parity
                while (consumed[a0] < i - RING + 1 || consumed[2 + a0] < i - RING + 1) __builtin_amdgcn_s_sleep(1);
                __threadfence_block();                         // acquire: the slot is overwritten only after the adders' reads of it
            }
            if (live) ring[i % RING][ll] = D;
            __threadfence_block();
            if (lane == 0) produced = i + 1;
            D = D.dbl();
        }
    } else {
        const int a = wave - 1, half = a >> 1, par = a & 1;    // consumed[] index: half * 2 + parity -> match the doubler's view
        const int slot = par + 2 * half;
        G1XYZZ acc = G1XYZZ::inf();
        for (int i = par; i < 128; i += 2) {
            while (produced <= i) __builtin_amdgcn_s_sleep(1);
            __threadfence_block();                             // acquire: the ring read below may not move above the poll of `produced`
            const uint32_t bit = ((half ? k2[i >> 5] : k1[i >> 5]) >> (i & 31)) & 1u;
            if (bit && live && split) {
                const G1XYZZ D = ring[i % RING][ll];
                acc.add(D);
            }
            __threadfence_block();
            if (lane == 0) consumed[slot] = i + 1;
        }
        if (live) part[slot][ll] = acc;
    }
    __syncthreads();
    // R1 = +-(even + odd of k1), R2 = +-phi(even + odd of k2)
    if (wave < 2 && live) {
        G1XYZZ t = part[2 * wave][ll];
        t.add(part[2 * wave + 1][ll]);
        if (wave == 1) {
            Fp beta;
            const uint32_t BETA[8] = {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu, 0x0u, 0x0u};
#pragma unroll
            for (int i = 0; i < 8; i++) beta.l[i] = BETA[i];
            t.X = t.X * beta.to_mont();
        }
        if (wave == 0 ? n1 : n2) t = t.neg();
        part[2 * wave][ll] = t;
    }
    __syncthreads();
    if (wave == 0 && live) {
        G1XYZZ t = part[0][ll];
        t.add(part[2][ll]);
        if (!split) t = scalar_mul(P, k, 256);
        if (blockIdx.y == 0) u_out[b] = t;
        else v_out[b] = t;
    }
}
// Stage 3: Krs = K + Z + T + U + V ; canonical outputs
__global__ __launch_bounds__(64) void k_finalize_stage3(const G1XYZZ *__restrict__ m_k, const G1XYZZ *__restrict__ m_z, const G1XYZZ *__restrict__ t_in,
                                                        const G1XYZZ *__restrict__ u_in, const G1XYZZ *__restrict__ v_in, const G1Affine *__restrict__ ar_aff,
                                                        ProofPoints *__restrict__ out, size_t B) {
    ZK_FOLD_SETPRIO();
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    G1XYZZ krs = m_k[b];
    krs.add(m_z[b]);
    krs.add(t_in[b]);
    krs.add(u_in[b]);
    krs.add(v_in[b]);
    const G1Affine krs_a = krs.to_affine(), ar_a = ar_aff[b];
    store_fp(out[b].ar[0], ar_a.x.from_mont()); store_fp(out[b].ar[1], ar_a.y.from_mont());
    store_fp(out[b].krs[0], krs_a.x.from_mont()); store_fp(out[b].krs[1], krs_a.y.from_mont());
}
// Bs = B2 + beta2 + s*delta2
__global__ __launch_bounds__(64) void k_finalize_g2(const G2XYZZ *__restrict__ m_bs2, const G2Affine *__restrict__ misc, const G2Affine *__restrict__ dtab,
                                                    const U256 *__restrict__ s, ProofPoints *__restrict__ out, size_t B) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t sl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) sl[k] = s[b].l[k];
    G2XYZZ bs = m_bs2[b];
    bs.madd(misc[0]);
    bs.add(fixed_mul(dtab, sl));
    const G2Affine a = bs.to_affine();
    store_fp(out[b].bs[0], a.x.a0.from_mont()); store_fp(out[b].bs[1], a.x.a1.from_mont());
    store_fp(out[b].bs[2], a.y.a0.from_mont()); store_fp(out[b].bs[3], a.y.a1.from_mont());
}
__global__ void k_store_commit(const G1Affine *com, const G1Affine *pok, ProofPoints *out, size_t B, int has) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (has) {
        store_fp(out[b].commitment[0], com[b].x); store_fp(out[b].commitment[1], com[b].y);
        store_fp(out[b].pok[0], pok[b].x); store_fp(out[b].pok[1], pok[b].y);
    } else {
        for (int k = 0; k < 8; k++) { out[b].commitment[0].l[k] = 0; out[b].commitment[1].l[k] = 0; out[b].pok[0].l[k] = 0; out[b].pok[1].l[k] = 0; }
    }
}


static constexpr size_t TAIL_TREE_MAX_B = 128;      // batches up to this size: fixed-base products as 16-lane trees
static void launch_finalize_g2(const G2XYZZ *m_bs2, const G2Affine *misc, const G2Affine *dtab, const U256 *s, ProofPoints *out, size_t B, hipStream_t st) {
    if (B <= TAIL_TREE_MAX_B) hipLaunchKernelGGL(k_finalize_g2_small, dim3((unsigned)B), dim3(16), 0, st, m_bs2, misc, dtab, s, out, B);
    else hipLaunchKernelGGL(k_finalize_g2, dim3(ceil_div(B, 64)), dim3(64), 0, st, m_bs2, misc, dtab, s, out, B);
}

// ------------------------------------------------------------------------- BSB22 challenge on the device
// challenge = fr.Hash(commitment.Marshal(), "bsb22-commitment", 1)[0]  (gnark backend/groth16/bn254/prove.go; gnark-crypto
// field/hash: RFC 9380 expand_message_xmd with SHA-256, 48 bytes, big-endian, reduced mod r).  One thread per proof: five
// SHA-256 compressions and one 384-bit reduction; keeps the solve stage stream-ordered (no device-host round trip in the
// middle of the witness solve).  The expander is the general device function of xmd.cuh, which the parity tests run against the
// RFC's published vectors; the bytes equal hostutil.cpp::hash_to_fr's (tests: proofs with commitments == oracle).
// com: canonical affine commitments (little-endian limbs; infinity = zeros, serialised as zero bytes like the host code);
// challenge: Montgomery form, as the solver reads it
__global__ __launch_bounds__(64) void k_bsb22_challenge(const G1Affine *__restrict__ com, Fr *__restrict__ challenge, size_t B) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint8_t DST[16] = {'b', 's', 'b', '2', '2', '-', 'c', 'o', 'm', 'm', 'i', 't', 'm', 'e', 'n', 't'};
    // msg = G1Affine.Marshal() = X || Y, 32 bytes big-endian each  [gnark-mem: uncompressed form, no public committed values]
    uint8_t m[64], uni[48];
    for (int i = 0; i < 8; i++) {
        const uint32_t x = com[b].x.l[7 - i], y = com[b].y.l[7 - i];
        m[4 * i] = (uint8_t)(x >> 24); m[4 * i + 1] = (uint8_t)(x >> 16); m[4 * i + 2] = (uint8_t)(x >> 8); m[4 * i + 3] = (uint8_t)x;
        m[32 + 4 * i] = (uint8_t)(y >> 24); m[32 + 4 * i + 1] = (uint8_t)(y >> 16); m[32 + 4 * i + 2] = (uint8_t)(y >> 8); m[32 + 4 * i + 3] = (uint8_t)y;
    }
    expand_message_xmd_dev(m, 64, DST, 16, uni, 48);
    // uniform bytes as a 384-bit big-endian integer = hi * 2^256 + lo, hi = first 16 bytes
    Fr hi = Fr::zero(), lo;
    for (int i = 0; i < 4; i++) hi.l[3 - i] = ((uint32_t)uni[4 * i] << 24) | ((uint32_t)uni[4 * i + 1] << 16) | ((uint32_t)uni[4 * i + 2] << 8) | uni[4 * i + 3];
    for (int i = 0; i < 8; i++) lo.l[7 - i] = ((uint32_t)uni[16 + 4 * i] << 24) | ((uint32_t)uni[16 + 4 * i + 1] << 16) | ((uint32_t)uni[16 + 4 * i + 2] << 8) | uni[16 + 4 * i + 3];
    // Montgomery form of (hi R + lo) mod r  (R = 2^256; to_mont accepts any 256-bit operand)
    challenge[b] = hi.to_mont().to_mont() + lo.to_mont();
}

// ---------------------------------------------------------------------------- comb quotient MSM (msm_comb.hip)
// Table build is triggered by the first batch of >= comb_min_batch() proofs (128: from there on the comb beats the bucket pipeline --
// 6 465 vs 5 694 proofs/s at B = 128, 8 343 vs 6 734 at 256, 9 819 vs 7 474 at 512, equal at 96; scripts/debug/batch_sweep.sh); once the tables exist every batch that fills whole
// wavefronts (B > SIDE_G2_MAX_B) takes the comb path.  ZKPROVE_COMB=0 keeps the bucket pipeline (A/B runs), ZKPROVE_COMB_K fixes the
// clump size (default: the largest k <= 19 whose table fits 30 % of the free device memory), ZKPROVE_COMB_MIN_BATCH the trigger.
static bool comb_enabled() {
    static const bool on = [] { const char *e = getenv("ZKPROVE_COMB"); return !(e && e[0] == '0'); }();
    return on;
}
static size_t comb_min_batch() {
    static const size_t v = [] { const char *e = getenv("ZKPROVE_COMB_MIN_BATCH"); return e ? (size_t)atol(e) : (size_t)128; }();
    return v;
}
// The clump shape whose table fits a byte budget: the largest k <= 20 whose one-size table fits, and -- when `mixed` -- as many leading
// clumps of k + 1 bases as the rest of the budget holds (CombShape::n1; k = 20 stays one size: a clump of 21 doubles a 100 GB table).
struct CombPick { int k = 0; size_t n1 = 0; size_t bytes = 0; size_t clumps = 0; };
static CombPick comb_pick_shape(size_t n, double budget, bool mixed) {
    CombPick p;
    for (int k = 20; k >= 12 && !p.k; k--)
        if ((double)comb_table_bytes(n, k) <= budget) p.k = k;
    if (!p.k) return p;
    if (mixed && p.k < 20) p.n1 = comb_mixed_fit(n, p.k, (size_t)budget);
    p.bytes = comb_mixed_bytes(n, p.k, p.n1);
    p.clumps = comb_mixed_clumps(n, p.k, p.n1);
    return p;
}
// ZKPROVE_COMB_FILL=0: the round-4 budget only -- a share of the device, one clump size, no top-up (same-box A/B of the device-filling budget)
// Copies of the bit-plane patterns of the quotient comb (nclumps x 254 x B words each: 3.4 GB ChaCha20, 7.4 GB AES-256 at B = 1024).  TWO let the
// next batch's patterns be written while this batch is walked -- but nothing runs under the walk (its single-wave workgroups take every
// wave slot), so the extraction ends up in the window behind it anyway; with ONE (default since round 5) it is enqueued behind the
// walk in front and the second copy's 10.7 GB go into the tables.  ZKPROVE_PAT_SLOTS=2: the old double buffer (same-box A/B: DESIGN.md section 9).
static int pat_slots() {
    static const int v = [] { const char *e = getenv("ZKPROVE_PAT_SLOTS"); return e && e[0] == '2' ? 2 : 1; }();
    return v;
}
static bool comb_fill_enabled() {
    static const bool on = [] { const char *e = getenv("ZKPROVE_COMB_FILL"); return !(e && e[0] == '0'); }();
    return on;
}
static int comb_forced_k() {
    static const int forced = [] { const char *e = getenv("ZKPROVE_COMB_K"); return e ? atoi(e) : 0; }();
    return forced;
}
static std::mutex g_comb_budget_lock[64];      // per device: replicas on different GPUs decide and build in parallel
static bool comb_memlog() {
    static const bool log = getenv("ZKPROVE_MEMLOG") != nullptr;
    return log;
}
size_t Algorithm::resident_batch() const { return res ? res->B : 0; }
void Algorithm::debug_leftovers(long *out3) {
    DeviceScope on_device(device);
    Resident &R = *res;
    std::vector<uint32_t> h(R.B);
    ZK_HIP(hipMemcpy(h.data(), R.ms.gen_count.p, R.B * sizeof(uint32_t), hipMemcpyDeviceToHost));
    out3[0] = out3[1] = out3[2] = 0;
    for (uint32_t v : h) { out3[0] += v; out3[1] = std::max<long>(out3[1], v); out3[2] += v != 0; }
}
bool Algorithm::comb_active() const {
    return pk.Zcomb.ok && res && split_world == 1 && res->B >= COMB_SMALL_MIN_B && res->comb_cap >= res->B;
}
// What this replica will still allocate at its reserved capacity: the second and third working set of the pipelined entries, the scratch
// of the commitment MSMs on the solve stream.  (A peer's budget decision sets this aside; caller holds mu.)
size_t Algorithm::pending_bytes() const {
    if (!res) return 0;
    const Resident &R = *res;
    const size_t B = R.cap, N = (size_t)1 << prog.log_domain;
    const size_t per_set = B * (prog.n_wires + 2 * N + (pk.Zcomb.ok ? 0 : N)) * sizeof(Fr) + ((size_t)64 << 20);      // (with comb tables the further sets keep no c rows of their own)
    size_t need = ((R.set[1].ready ? 0 : 1) + (R.set[2].ready ? 0 : 1)) * per_set;
    if (has_commit && R.ms2.entries.n == 0)
        need += B * (pk.ped.n * (size_t)pk.ped.W * 4 + ((size_t)4 << (pk.ped.c - 1)) + msm_part_stride(pk.ped.n, pk.ped.c) * sizeof(G1XYZZ) + ((size_t)144 << (pk.ped.c - 1)) + pk.ped.n * 36);
    if (has_commit && !wire_comb_tried && comb_enabled()) need += (size_t)6 << 30;      // wire-comb tables (4 GiB at clumps of 16) and their pattern / accumulator scratch
    return need;
}
// A peer as this replica's budget decision sees it.  settled: it has reserved the working buffers of a large batch and decided its own
// tables (or will never build any) -- what it still allocates is `pending`.  can_top / weight: it holds tables that may still grow; the
// bytes a whole clump size more would cost it (the free memory is shared in proportion: equal relative gain per byte).
Algorithm::CombPeer Algorithm::comb_peer_view() {
    CombPeer v;
    if (!res || res->cap < comb_min_batch() || !pk.comb_tried) return v;      // has not served a large batch yet: its needs are unknown
    v.settled = true;
    v.pending = pending_bytes();
    if (pk.Zcomb.ok && !comb_topped && pk.Zcomb.k < 20) {
        v.can_top = true;
        v.weight = zk::comb_table_bytes(pk.Zcomb.n, pk.Zcomb.k + 1) - zk::comb_table_bytes(pk.Zcomb.n, pk.Zcomb.k);
    }
    return v;
}
// Table build is triggered by the first batch of >= comb_min_batch() proofs.  Budget, in two regimes:
//  * some circuit loaded on this device has not served a large batch yet (its working sets and tables are still to come): a share of
//    the device -- 0.42 of its memory divided by the replicas loaded on it -- and at most 75 % of what is free beyond this replica's own
//    further needs; one clump size throughout (k = 19 for a 2^16 domain, 18 for 2^17 with two circuits loaded);
//  * every loaded circuit has settled (or this one is alone): what is really free, less every replica's pending allocations and the
//    headroom, goes into the tables -- mixed clump sizes fill it to the last GiB -- divided between the circuits whose tables can still
//    grow in proportion to the bytes a clump size more costs each (equal relative gain per byte: a 2^17 domain at k = 18 and a 2^16
//    domain at k = 19 both buy 5 % fewer additions for ~50 GB).  A circuit that decided under the first regime is topped up later, at
//    its next large batch after the device settled (comb_top_up: one rebuild).
// Decisions are serialised per device so that each one sees the tables of the replicas before it on the same GPU.
// kept free beyond every known allocation: the 4 GiB every large allocation must leave (require_headroom, common.h) + room for what the
// estimates do not list (verifier images, pinned-copy staging, the host application's own buffers)
static constexpr size_t COMB_HEADROOM = (size_t)9 << 30;
void Algorithm::comb_prepare(size_t B) {
    if (!comb_enabled() || split_world != 1 || B < COMB_SMALL_MIN_B) return;
    Resident &R = *res;
    if (!pk.comb_tried && B >= comb_min_batch()) {
        pk.comb_tried = true;
        std::lock_guard<std::mutex> budget(g_comb_budget_lock[device & 63]);
        // still to come for this replica: the second and third working set of the pipelined entries, the two pattern copies (~ the
        // table's clump count x 254 x B words: bounded by the k = 16 figure), range accumulators, and slack for the allocator
        const size_t still_needed = pending_bytes() + (size_t)pat_slots() * ((pk.ze.n / 16 + 1) * 254 * B * sizeof(uint32_t)) + COMB_HEADROOM;
        const int loaded = loaded_on_device ? loaded_on_device(device) : comb_share;      // algorithm replicas on this GPU right now, this one included
        const int share = std::max(std::max(loaded, comb_share), 1);
        size_t free_b = 0, total_b = 0;
        ZK_HIP(hipMemGetInfo(&free_b, &total_b));
        // the neighbours: all settled?  what will they still allocate, and how much of the free memory is theirs to grow into?
        bool settled = true;
        size_t peers_pending = 0, peers_weight = 0;
        if (peers_visit) {
            const bool complete = peers_visit(this, [&](Algorithm *p) {
                std::unique_lock<std::mutex> lk(p->mu, std::try_to_lock);
                if (!lk) { settled = false; return; }                      // proving right now: ask again another time
                const CombPeer v = p->comb_peer_view();
                if (!v.settled) { settled = false; return; }
                peers_pending += v.pending;
                if (v.can_top) peers_weight += v.weight;
            });
            settled = settled && complete;
        } else settled = share == 1;
        const double by_share = 0.42 * (double)total_b / (double)share;
        const double by_free = free_b > still_needed ? 0.75 * (double)(free_b - still_needed) : 0.0;
        CombPick pick = comb_pick_shape(pk.ze.n, std::min(by_share, by_free), false);
        if (comb_forced_k()) {      // ZKPROVE_COMB_K (and ZKPROVE_COMB_N1: leading clumps of k + 1 bases): a fixed shape, e.g. the two-circuit shape for a one-circuit profiling run
            static const size_t forced_n1 = [] { const char *e = getenv("ZKPROVE_COMB_N1"); return e ? (size_t)atol(e) : (size_t)0; }();
            pick = CombPick();
            pick.k = comb_forced_k();
            pick.n1 = std::min(forced_n1, pk.ze.n / (size_t)(pick.k + 1));
            pick.bytes = comb_mixed_bytes(pk.ze.n, pick.k, pick.n1);
            pick.clumps = comb_mixed_clumps(pk.ze.n, pick.k, pick.n1);
        }
        else if (settled && pick.k && comb_fill_enabled()) {
            // everything the neighbours need is known: fill the device
            const double avail = (double)free_b - (double)still_needed - (double)peers_pending;
            const size_t my_weight = pick.k < 20 ? zk::comb_table_bytes(pk.ze.n, pick.k + 1) - zk::comb_table_bytes(pk.ze.n, pick.k) : 0;
            const double frac = (my_weight + peers_weight) ? (double)my_weight / (double)(my_weight + peers_weight) : 1.0;
            const double extra = avail - (double)pick.bytes;
            if (extra > 0) {
                // (the patterns of a finer table are fewer, never more: still_needed stays an upper bound)
                const CombPick full = comb_pick_shape(pk.ze.n, std::min((double)pick.bytes + 0.97 * extra * frac, 0.42 * (double)total_b), true);
                if (full.k && full.bytes >= pick.bytes) pick = full;
            }
            comb_topped = true;
        }
        if (comb_memlog())
            fprintf(stderr, "zkprove: comb budget: free %.1f GiB of %.1f, still needed %.1f, peers pending %.1f, share 1/%d, device %s -> k = %d, %zu clumps of %d (%zu clumps, %.1f GiB)\n",
                    free_b / 1073741824.0, total_b / 1073741824.0, still_needed / 1073741824.0, peers_pending / 1073741824.0, share, settled ? "settled" : "not settled",
                    pick.k, pick.n1, pick.k + 1, pick.clumps, pick.bytes / 1073741824.0);
        bool ok = false;
        std::string why = "not enough free device memory";
        if (pick.k) {
            try {
                ok = comb_build(pk.Zcomb, pk.ze.p, pk.ze.n, pick.k, R.stream, pick.n1);
                pk.Zcomb.hi_prio = dom.hi_prio;
                if (!ok) why = "degenerate bases";
            } catch (const ZkError &e) {
                pk.Zcomb.table.release();
                (void)hipGetLastError();
                why = e.what();
            }
        }
        if (!ok) fprintf(stderr, "zkprove: comb tables of the quotient MSM not built (%s): the bucket pipeline is kept\n", why.c_str());
    }
    if (pk.Zcomb.ok && R.comb_cap < B) {
        R.comb.reserve(pk.Zcomb, B, pat_slots());
        R.comb_cap = B;
    }
}
// Top-up (proving entries, before anything is enqueued: nothing of this replica is in flight).  The tables were sized while a
// neighbour's needs were unknown; now that every circuit on the device has settled, this replica's share of what is still free goes
// into a prefix of larger clumps.  The old table is released first (both do not fit), so a failed build falls back to the old shape.
void Algorithm::comb_top_up() {
    if (!res || !pk.Zcomb.ok || comb_topped || comb_forced_k() || !comb_fill_enabled() || split_world != 1 || res->B < comb_min_batch() || !peers_visit) return;
    Resident &R = *res;
    std::lock_guard<std::mutex> budget(g_comb_budget_lock[device & 63]);
    bool settled = true;
    size_t peers_pending = 0, peers_weight = 0;
    const bool complete = peers_visit(this, [&](Algorithm *p) {
        std::unique_lock<std::mutex> lk(p->mu, std::try_to_lock);
        if (!lk) { settled = false; return; }
        const CombPeer v = p->comb_peer_view();
        if (!v.settled) { settled = false; return; }
        peers_pending += v.pending;
        if (v.can_top) peers_weight += v.weight;
    });
    if (!settled || !complete) return;                      // not yet: the next large batch asks again
    comb_topped = true;
    const CombBases &old = pk.Zcomb;
    if (old.k >= 20) return;
    size_t free_b = 0, total_b = 0;
    ZK_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t old_bytes = old.table.bytes(), n = old.n;
    const int old_k = old.k;
    const size_t old_n1 = old.n1, old_clumps = old.nclumps;
    const double avail = (double)free_b - (double)pending_bytes() - (double)peers_pending - (double)COMB_HEADROOM;
    const size_t my_weight = zk::comb_table_bytes(n, old_k + 1) - zk::comb_table_bytes(n, old_k);
    const double frac = (double)my_weight / (double)(my_weight + peers_weight);
    if (avail <= 0) return;
    const CombPick pick = comb_pick_shape(n, std::min((double)old_bytes + 0.97 * avail * frac, 0.42 * (double)total_b), true);      // (one replica's tables: at most 0.42 of the device)
    if (comb_memlog())
        fprintf(stderr, "zkprove: comb top-up: free %.1f GiB, own pending %.1f, peers pending %.1f, share %.2f -> k = %d, %zu clumps of %d (%zu clumps, %.1f GiB; was %zu clumps, %.1f GiB)\n",
                free_b / 1073741824.0, pending_bytes() / 1073741824.0, peers_pending / 1073741824.0, frac, pick.k, pick.n1, pick.k + 1, pick.clumps, pick.bytes / 1073741824.0,
                old_clumps, old_bytes / 1073741824.0);
    if (!pick.k || (double)pick.clumps > 0.993 * (double)old_clumps) return;      // less than 0.7 % fewer additions: not worth a rebuild
    ZK_HIP(hipDeviceSynchronize());
    pk.Zcomb = CombBases();
    bool ok = false;
    try {
        ok = comb_build(pk.Zcomb, pk.ze.p, n, pick.k, R.stream, pick.n1);
        pk.Zcomb.hi_prio = dom.hi_prio;
    } catch (const ZkError &e) {
        pk.Zcomb = CombBases();
        (void)hipGetLastError();
        fprintf(stderr, "zkprove: comb top-up failed (%s): rebuilding the previous tables\n", e.what());
    }
    if (!ok) {
        try { ok = comb_build(pk.Zcomb, pk.ze.p, n, old_k, R.stream, old_n1); pk.Zcomb.hi_prio = dom.hi_prio; } catch (const ZkError &) { pk.Zcomb = CombBases(); (void)hipGetLastError(); }
        if (!ok) { fprintf(stderr, "zkprove: comb tables lost: this algorithm continues on the bucket pipeline\n"); comb_release(); return; }
    }
    R.comb.reserve(pk.Zcomb, R.cap, pat_slots());
}

void bsb22_challenge_device(const G1Affine *com_canonical, Fr *challenge_mont, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_bsb22_challenge, dim3(ceil_div(n, 64)), dim3(64), 0, s, com_canonical, challenge_mont, n);
}

// ---------------------------------------------------------------------------- wire comb (calibration)
// width class of the values of one wire position over the calibration batch: 0 = 0 / +-1 only, 1 = < 2^8, 2 = < 2^32, 3 = anything
__global__ __launch_bounds__(256) void k_wire_class(const Fr *__restrict__ wires, size_t nw, size_t B, uint8_t *__restrict__ cls) {
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const Fr one = Fr::one(), mone = Fr::one().neg();
    uint32_t c = 0;
    for (size_t b = 0; b < B; b++) {
        const Fr v = wires[w * B + b];
        if (v.is_zero() || v == one || v == mone) continue;
        const Fr s = v.from_mont();
        uint32_t hi = 0;
        for (int k = 1; k < 8; k++) hi |= s.l[k];
        const uint32_t k = hi ? 3u : (s.l[0] >> 8 ? 2u : 1u);
        c = k > c ? k : c;
    }
    cls[w] = (uint8_t)c;
}
static int wire_comb_k() {
    static const int v = [] { const char *e = getenv("ZKPROVE_WCOMB_K"); return e ? atoi(e) : 16; }();
    return v;
}
void Algorithm::wire_comb_prepare(size_t B) {
    if (!comb_enabled() || wire_comb_tried || split_world != 1 || B < comb_min_batch() || calib_inputs.empty() || !pk.A.f9 || !pk.B2.f9) return;
    wire_comb_tried = true;
    Resident &R = *res;
    const size_t ni = prog.n_inputs, Bc = std::min(calib_inputs.size() / ni, R.cap), nw = prog.n_wires;
    if (!Bc) return;
    SolveSet &S = R.set[0];
    // 1. calibration solve (set 0 is free: nothing has been uploaded for the batch being reserved yet).  The batch size of the set is
    // borrowed for it and put back on every way out (an allocation or HIP error below reaches resident_reserve's retry loop).
    struct RestoreB {
        Resident &R; SolveSet &S; size_t B, sbB, msB; bool armed = true;
        void now() { if (armed) { R.B = B; S.sb.B = sbB; R.ms.B = msB; armed = false; } }
        ~RestoreB() { now(); }
    } restore{R, S, R.B, S.sb.B, R.ms.B};
    R.B = Bc; S.sb.B = Bc; R.ms.B = Bc;
    std::vector<U256> ones(Bc);
    for (auto &v : ones) { memset(v.l, 0, 32); v.l[0] = 1; }
    upload_set(S, Bc, calib_inputs.data(), ones.data(), ones.data(), calib_rnd.data(), R.stream);
    enqueue_solve(S, R.ms, R.stream);
    std::vector<uint32_t> errs(Bc);
    ZK_HIP(hipMemcpyAsync(errs.data(), S.sb.err.p, Bc * sizeof(uint32_t), hipMemcpyDeviceToHost, R.stream));
    DevBuf<uint8_t> dcls(nw);
    hipLaunchKernelGGL(k_wire_class, dim3(ceil_div(nw, 256)), dim3(256), 0, R.stream, S.sb.wires.p, nw, Bc, dcls.p);
    std::vector<uint8_t> cls(nw);
    ZK_HIP(hipMemcpyAsync(cls.data(), dcls.p, nw, hipMemcpyDeviceToHost, R.stream));
    ZK_HIP(hipStreamSynchronize(R.stream));
    ZK_HIP(hipStreamSynchronize(R.copy_stream));
    restore.now();
    for (uint32_t e : errs)
        if (e) { fprintf(stderr, "zkprove: wire-comb calibration: the synthetic witnesses did not solve (error word %u): wire MSMs keep the bucket pipeline\n", e); return; }
    size_t nwide = 0;
    for (uint8_t c : cls) nwide += c != 0;
    if (!nwide) return;                    // every wire is 0 / +-1 (the ChaCha20 circuit): nothing to tabulate
    // 2. tables per base set
    auto attach = [&](auto &mb) -> bool {
        if (!mb.n) return true;
        std::vector<uint32_t> rows(mb.n);
        ZK_HIP(hipMemcpy(rows.data(), mb.scalar_row.p, mb.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::vector<uint8_t> c(mb.n);
        for (size_t j = 0; j < mb.n; j++) c[j] = cls[rows[j]];
        return msm_attach_wire_comb(mb, c.data(), wire_comb_k(), R.stream);
    };
    auto detach_all = [&] {
        for (MsmBasesG1 *mb : {&pk.A, &pk.B1, &pk.K, &pk.ped, &pk.ped_sigma}) { mb->wcomb.clear(); mb->wide_mask.release(); }
        pk.B2.wcomb.clear(); pk.B2.wide_mask.release();
    };
    bool ok = false;
    try {
        ok = attach(pk.A) && attach(pk.B1) && attach(pk.K) && attach(pk.B2);
        if (ok && has_commit) ok = attach(pk.ped) && attach(pk.ped_sigma);
    } catch (...) {
        detach_all();          // a table allocation failed half way (memory pressure): nothing half-attached stays behind, unused and unfreed
        throw;
    }
    if (!ok) {
        detach_all();
        fprintf(stderr, "zkprove: wire-comb tables not built (degenerate bases): wire MSMs keep the bucket pipeline\n");
        return;
    }
    wire_comb_on = true;
}
// scratch of the wire combs that run on `sc`: the main-stream MSMs (A, B1, K, B2; and the Pedersen pair when it shares that scratch)
// or the solve-stream ones (the Pedersen pair of the pipelined entries)
void Algorithm::wire_comb_reserve(MsmScratch &sc, bool main_set, size_t B) {
    if (!wire_comb_on) return;
    size_t pw = 0, ab = 0, rb = 0;
    auto need = [&](const auto &mb) {
        size_t p = 0, a = 0, r = 0;
        msm_wire_comb_needs(mb, B, p, a, r);
        pw = std::max(pw, p); ab = std::max(ab, a); rb = std::max(rb, r);
    };
    if (main_set) { need(pk.A); need(pk.B1); need(pk.K); need(pk.B2); }
    if (has_commit) { need(pk.ped); need(pk.ped_sigma); }
    sc.reserve_wire_comb(pw, ab, rb);
}

// ------------------------------------------------------------------------------- pipeline
static std::atomic<uint64_t> g_use_clock{0};      // Algorithm::last_use stamps (least recently used algorithm gives its tables back first)
void Algorithm::resident_reserve(size_t B) {
    DeviceScope on_device(device);
    if (!res) {
        res.reset(new Resident());
        // the solve stage and the proof tail are chains of small latency-bound kernels: on high-priority streams they slip in
        // between the wavefronts of the large NTT / MSM kernels instead of queueing behind them
        int prio_lo = 0, prio_hi = 0;
        ZK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        // non-blocking: a synchronous copy on the null stream must not drain the pipeline
        ZK_HIP(hipStreamCreateWithPriority(&res->stream, hipStreamNonBlocking, prio_lo));
        ZK_HIP(hipStreamCreateWithPriority(&res->stream2, hipStreamNonBlocking, prio_hi));
        ZK_HIP(hipStreamCreateWithPriority(&res->stream3, hipStreamNonBlocking, prio_hi));
        // exactly four streams: a fifth one no longer gets a hardware queue of its own (HIP multiplexes streams over 4 queues) and
        // doubled the latency of the 160-460 dependent solver launches.  The copy stream doubles as the side stream of small batches.
        // Its priority decides who gets the wave slots the transforms (copy stream) and the next batch's solve stage (stream2) compete
        // for.  A long solve stage (AES: 430-460 levels, a commitment MSM in the middle) must win, or it is not through when the
        // quotient walk ends and the transforms of ITS batch start late: AES-256 218.5 ms per step with the copy stream low, 232.2 high.
        // A short one (ChaCha20: 163 levels) finishes anyway and only delays the transforms, which are the critical chain there:
        // 88.1 ms high, 88.9 low.  ZKPROVE_COPY_PRIO=hi|lo overrides.
        static const int copy_prio = [] { const char *e = ZK_AB_ENV("ZKPROVE_COPY_PRIO"); return !e ? -1 : !strcmp(e, "hi") ? 1 : 0; }();
        const bool copy_hi = copy_prio < 0 ? prog.levels.size() <= 256 : copy_prio == 1;
        ZK_HIP(hipStreamCreateWithPriority(&res->copy_stream, hipStreamNonBlocking, copy_hi ? prio_hi : prio_lo));
        for (auto &e : res->ev) ZK_HIP(hipEventCreate(&e));
    }
    Resident &R = *res;
    for (int attempt = 0;; attempt++) {
        try {
            resident_reserve_buffers(B);
            break;
        } catch (const ZkError &) {
            (void)hipGetLastError();
            // DevBuf::alloc releases before it allocates: the buffer whose growth failed is EMPTY now, so no capacity recorded before
            // the failure may survive it -- neither for the retry nor, when the retries run out, for the next (smaller) call, which
            // would otherwise skip its ensure() calls and launch kernels on a null pointer.  Every reservation is "grow if smaller",
            // so starting from zero costs nothing for the buffers that are still in place.
            R.cap = 0; R.dense_cap = 0; R.comb_cap = 0; R.side_cap = 0; R.B = 0;
            if (attempt >= 5 || !relieve_pressure(false)) throw;      // memory-pressure valve, see comb_release
        }
    }
}
// Instances the c rows of the second / third working set are sized for: with comb tables only the products buffer of the batches too
// small for the comb (the solver writes its output column into the first set's rows, SolverBuffers::c_shared); without tables the
// bucket pipeline keeps every batch's products there.
size_t Algorithm::side_max_b() const { return has_commit && wire_comb_on ? SIDE_G2_MAX_B_COMMIT : SIDE_G2_MAX_B; }
size_t Algorithm::own_c_rows(size_t B) const { return pk.Zcomb.ok ? std::min(B, side_max_b()) : B; }
void Algorithm::resident_reserve_buffers(size_t B) {
    Resident &R = *res;
    R.B = B;
    // (tables given back under memory pressure: the further sets need full c rows again)
    for (int i = 1; i < 3; i++)
        if (R.set[i].ready && R.cap) R.set[i].sb.c.ensure(((size_t)1 << prog.log_domain) * own_c_rows(R.cap));
    if (B > R.cap) {
        R.cap = B;
        R.set[0].reserve(prog, B);
        if (R.set[1].ready) R.set[1].reserve(prog, B, own_c_rows(B));
        if (R.set[2].ready) R.set[2].reserve(prog, B, own_c_rows(B));
        size_t max_entries = 0, max_nb = 0, max_part = 0;
        auto acc = [&](size_t n, int c, int W) {
            max_entries = std::max(max_entries, n * (size_t)W);
            max_nb = std::max(max_nb, (size_t)1 << (c - 1));
            max_part = std::max(max_part, msm_part_stride(n, c));
        };
        acc(pk.A.n, pk.A.c, pk.A.W); acc(pk.B1.n, pk.B1.c, pk.B1.W); acc(pk.K.n, pk.K.c, pk.K.W);      // (the quotient MSM's share: below)
        acc(pk.B2.n, pk.B2.c, pk.B2.W);
        if (has_commit) acc(pk.ped.n, pk.ped.c, pk.ped.W);
        R.ms.reserve(B, max_entries, max_nb, max_part);
        R.ms.reserve_compact(B, std::max({pk.A.n, pk.B1.n, pk.K.n, pk.B2.n, has_commit ? pk.ped.n : (size_t)1}));
        static const bool g2_side_on = [] { const char *e = ZK_AB_ENV("ZKPROVE_G2_SIDE"); return e && e[0] == '1'; }();
        if (g2_side_on && B > side_max_b() && split_world == 1) {          // opt-in: the G2 MSM of large batches beside the G1 MSMs (enqueue_main)
            R.ms_b2.reserve(B, pk.B2.n * (size_t)pk.B2.W, (size_t)1 << (pk.B2.c - 1), msm_part_stride(pk.B2.n, pk.B2.c));
            R.ms_b2.reserve_compact(B, pk.B2.n);
            R.ms_b2.reserve_sums(B, 1, (size_t)1 << (pk.B2.c - 1));
        }
        {
            size_t nb1 = (size_t)1 << (std::max({pk.A.c, pk.B1.c, pk.K.c, has_commit ? pk.ped.c : 1}) - 1);
            R.ms.reserve_sums(B, nb1, (size_t)1 << (pk.B2.c - 1));
        }
        R.m_ar.alloc(B); R.m_bs1.alloc(B); R.m_k.alloc(B); R.m_z.alloc(B); R.m_bs2.alloc(B);
        R.ar_aff.alloc(B); R.bs1_aff.alloc(B); R.f_t.alloc(B); R.f_u.alloc(B); R.f_v.alloc(B);
    }
    // the side scratches of the batches whose MSM chains run side by side (side_max_b: 64 proofs; 128 once an AES program has its wire
    // combs): before the table decision -- the budget looks at what is left -- and again when the wire combs have raised the limit
    auto reserve_side = [&]() {
        const size_t Bs = std::min(R.cap, side_max_b());
        if (R.side_cap >= Bs) return;
        const size_t n_side = std::max(pk.B2.n, has_commit ? pk.ped_sigma.n : (size_t)1);
        const int c_side = std::max(pk.B2.c, has_commit ? pk.ped_sigma.c : 1);
        R.ms_g2.reserve(Bs, n_side * (size_t)std::max(pk.B2.W, has_commit ? pk.ped_sigma.W : 1), (size_t)1 << (c_side - 1),
                        std::max(msm_part_stride(pk.B2.n, pk.B2.c), has_commit ? msm_part_stride(pk.ped_sigma.n, pk.ped_sigma.c) : (size_t)1));
        const size_t na = std::max(pk.A.n, pk.B1.n);
        R.ms_a.reserve(Bs, na * (size_t)pk.A.W, (size_t)1 << (pk.A.c - 1), std::max(msm_part_stride(pk.A.n, pk.A.c), msm_part_stride(pk.B1.n, pk.B1.c)));
        R.ms_a.reserve_compact(Bs, na);
        R.ms_a.reserve_sums(Bs, (size_t)1 << (pk.A.c - 1), 1);
        R.ms_k.reserve(Bs, pk.K.n * (size_t)pk.K.W, (size_t)1 << (pk.K.c - 1), msm_part_stride(pk.K.n, pk.K.c));
        R.ms_k.reserve_compact(Bs, pk.K.n);
        R.ms_k.reserve_sums(Bs, (size_t)1 << (pk.K.c - 1), 1);
        R.ms_g2.reserve_compact(Bs, n_side);
        R.ms_g2.reserve_sums(Bs, has_commit ? (size_t)1 << (pk.ped_sigma.c - 1) : 1, (size_t)1 << (pk.B2.c - 1));
        R.side_cap = Bs;
    };
    reserve_side();
    // the comb tables of the quotient MSM are decided once the working buffers above exist (the budget looks at what is left)
    comb_prepare(B);
    wire_comb_prepare(B);
    reserve_side();
    wire_comb_reserve(R.ms, true, R.cap);
    if (R.ms_b2.entries.n) wire_comb_reserve(R.ms_b2, true, R.cap);
    if (R.ms2.entries.n) wire_comb_reserve(R.ms2, false, R.cap);
    if (wire_comb_on) {
        // the side scratches of batches <= 64 (one MSM chain per stream): room for the wire combs of the base sets that run on them
        const size_t Bs = std::min(R.cap, side_max_b());
        auto side_room = [&](MsmScratch &sc, std::initializer_list<const MsmBasesG1 *> g1, const MsmBasesG2 *g2) {
            size_t pw = 0, ab = 0, rb = 0, p = 0, a = 0, r = 0;
            for (const MsmBasesG1 *mb : g1) { msm_wire_comb_needs(*mb, Bs, p, a, r); pw = std::max(pw, p); ab = std::max(ab, a); rb = std::max(rb, r); }
            if (g2) { msm_wire_comb_needs(*g2, Bs, p, a, r); pw = std::max(pw, p); ab = std::max(ab, a); rb = std::max(rb, r); }
            sc.reserve_wire_comb(pw, ab, rb);
        };
        side_room(R.ms_a, {&pk.A, &pk.B1}, nullptr);
        side_room(R.ms_k, {&pk.K}, nullptr);
        if (has_commit) side_room(R.ms_g2, {&pk.ped_sigma}, &pk.B2); else side_room(R.ms_g2, {}, &pk.B2);
    }
    // bucket pipeline of the quotient MSM: for every batch when there are no comb tables, else only for the batches too small for the comb
    const size_t nbz = (size_t)1 << (pk.Z.c - 1), ez = pk.Z.n * (size_t)pk.Z.W;
    const size_t Bd = pk.Zcomb.ok ? std::min(B, side_max_b()) : B;
    if (R.dense_cap < Bd) {
        R.ms.reserve(Bd, ez, nbz, msm_part_stride(pk.Z.n, pk.Z.c));
        R.ms.reserve_dense(Bd, ez, nbz);
        if (Bd > side_max_b()) R.ms.reserve_dense_second(Bd, ez, nbz);      // ahead mode, enqueue_ntt_sort
        R.ms.reserve_tree(Bd, ez, nbz);
        R.ms.reserve_sums(Bd, nbz, 1);
        R.dense_cap = Bd;               // only now: an allocation above may throw, and the retry (resident_reserve) must come back here
    }
    for (auto &st : R.set) st.sb.B = B;
    R.ms.B = B; R.ms2.B = B;
}

// inputs [B][n_inputs] (instance-major, as the host builds them), r / s / randomize [B] canonical -> set S, on stream st
void Algorithm::upload_set(SolveSet &S, size_t B, const uint32_t *inputs, const U256 *r, const U256 *s, const U256 *randomize, hipStream_t st, const uint8_t *raw,
                           int raw_alg) {
    S.n_wide = 0;
    S.sb.raw_alg = raw ? raw_alg : -1;
    if (raw) ZK_HIP(hipMemcpyAsync(S.sb.raw.p, raw, SOLVER_RAW_STRIDE * B, hipMemcpyHostToDevice, st));
    else ZK_HIP(hipMemcpyAsync(S.sb.inputs.p, inputs, prog.n_inputs * B * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(S.r.p, r, B * sizeof(U256), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(S.s.p, s, B * sizeof(U256), hipMemcpyHostToDevice, st));
    if (randomize) ZK_HIP(hipMemcpyAsync(S.rnd.p, randomize, B * sizeof(U256), hipMemcpyHostToDevice, st));
    else ZK_HIP(hipMemsetAsync(S.rnd.p, 0, B * sizeof(U256), st));
}

void Algorithm::resident_upload(size_t B, const uint32_t *inputs, const U256 *r, const U256 *s, const U256 *randomize, const WideInput *wide, size_t n_wide,
                                const uint8_t *raw, int raw_alg) {
    resident_reserve(B);
    Resident &R = *res;
    for (auto &S : R.set)
        if (S.ready) {
            upload_set(S, B, inputs, r, s, randomize, R.stream, raw, raw_alg);
            if (n_wide) {
                for (size_t i = 0; i < n_wide; i++)
                    if (wide[i].instance >= B || wide[i].index >= prog.n_inputs) throw ZkError("wide witness element out of range");
                S.wide.ensure(n_wide);
                ZK_HIP(hipMemcpyAsync(S.wide.p, wide, n_wide * sizeof(WideInput), hipMemcpyHostToDevice, R.stream));
                S.n_wide = n_wide;
            }
        }
    ZK_HIP(hipStreamSynchronize(R.stream));
}

// SOLVE stage of one batch (witness solve incl. the BSB22 commitment round trip) on stream st.  Events t0..t3 of the set
// bracket: [t0 levels-before-commit t1 commitment t2 remaining-levels t3].
void Algorithm::enqueue_solve(SolveSet &S, MsmScratch &commit_scratch, hipStream_t st) {
    Resident &R = *res;
    const size_t B = R.B;
    ZK_HIP(hipEventRecord(S.t0, st));
    {   // output column: own rows when they hold this batch, else the first set's (own_c_rows; solve stages share one stream)
        const size_t need = ((size_t)1 << prog.log_domain) * B;
        S.sb.c_shared = S.sb.c.n >= need ? nullptr : R.set[0].sb.c.p;
        if (S.sb.c_shared && (R.set[0].sb.c.n < need || !comb_active())) throw ZkError("internal: working set without rows for the output column");
    }
    hipLaunchKernelGGL(k_u256_to_mont, dim3(ceil_div(B, 64)), dim3(64), 0, st, S.rnd.p, S.sb.randomize.p, B);
    solver_begin(prog, S.sb, st);
    if (S.n_wide) hipLaunchKernelGGL(k_wide_inputs, dim3(ceil_div(S.n_wide, 64)), dim3(64), 0, st, S.wide.p, S.n_wide, S.sb.wires.p, B);
    if (has_commit) {
        solver_run_levels(prog, S.sb, 0, commit_level, st);
        ZK_HIP(hipEventRecord(S.t1, st));
        // BSB22 commitment: C = sum v_i * Basis_i ; challenge = H2F(C.Marshal(), "bsb22-commitment") ; PoK = sum v_i * BasisExpSigma_i
        if (B <= side_max_b()) {
            // small batch: the two Pedersen MSMs are latency-bound chains -- the proof-of-knowledge one runs on the side stream
            // (scratch of the side G2 MSM, idle until the MSM stage) beside the commitment
            ZK_HIP(hipEventRecord(R.ev[14], st));
            ZK_HIP(hipStreamWaitEvent(R.copy_stream, R.ev[14], 0));
            msm_batch(pk.ped_sigma, S.sb.wires.p, B, R.ms_g2, S.m_pok.p, R.copy_stream);
            g1_to_affine_canonical(S.m_pok.p, S.pok_aff.p, B, R.copy_stream);
            ZK_HIP(hipEventRecord(R.ev[14], R.copy_stream));
            msm_batch(pk.ped, S.sb.wires.p, B, commit_scratch, S.m_com.p, st);
            g1_to_affine_canonical(S.m_com.p, S.com_aff.p, B, st);
            ZK_HIP(hipStreamWaitEvent(st, R.ev[14], 0));
        } else {
            msm_batch(pk.ped, S.sb.wires.p, B, commit_scratch, S.m_com.p, st);
            msm_batch(pk.ped_sigma, S.sb.wires.p, B, commit_scratch, S.m_pok.p, st);
            g1_to_affine_canonical(S.m_com.p, S.com_aff.p, B, st);
            g1_to_affine_canonical(S.m_pok.p, S.pok_aff.p, B, st);
        }
        // challenge = H2F(C.Marshal(), "bsb22-commitment") on the device: the solve stage stays stream-ordered
        hipLaunchKernelGGL(k_bsb22_challenge, dim3(ceil_div(B, 64)), dim3(64), 0, st, S.com_aff.p, S.sb.challenge.p, B);
        ZK_HIP(hipEventRecord(S.t2, st));
        solver_run_levels(prog, S.sb, commit_level, prog.levels.size(), st);
    } else {
        ZK_HIP(hipEventRecord(S.t1, st));
        ZK_HIP(hipEventRecord(S.t2, st));
        solver_run_levels(prog, S.sb, 0, prog.levels.size(), st);
    }
    ZK_HIP(hipEventRecord(S.t3, st));
}

// the bucket pipeline keeps the instance-major products of a batch in the set's own c rows (own_c_rows sizes them for exactly the batches that take it)
static void products_rows_check(const SolverBuffers &sb, size_t N, size_t B) {
    if (sb.c.n < N * B) throw ZkError("internal: working set without rows for the quotient products");
}
// Ahead mode (pipelined entries, large batches): the transforms and the quotient MSM's bucket sort of a batch need nothing but its solved
// wires, so they are enqueued right behind its solve stage -- on the copy stream (side_chain_stream), i.e. while the main stream is still
// busy with an EARLIER batch.  (Nothing runs under the quotient walk itself: its single-wave workgroups hold every wave slot; the ahead
// work fills the window between two walks, beside the wire MSMs -- DESIGN.md section 6.)  The sort / pattern extraction then no longer
// stands on the critical path between the wire MSMs and the walk; the sorted list is double-buffered (slot = batch
// parity).  ZKPROVE_SORT_AHEAD=0 keeps everything in enqueue_main.
bool Algorithm::ahead_eligible() const {
    static const bool on = [] { const char *e = ZK_AB_ENV("ZKPROVE_SORT_AHEAD"); return !(e && e[0] == '0'); }();
    const Resident &R = *res;
    const size_t max_entries = pk.Z.n * (size_t)pk.Z.W, nb = (size_t)1 << (pk.Z.c - 1);
    if (!(on && R.B > side_max_b() && split_world == 1)) return false;
    if (comb_active()) return true;                      // the comb path prepares its bit-plane patterns ahead instead of a sorted list
    // the complete predicate of msm_dense_presort: a batch that does not meet it takes the in-line sort of enqueue_main
    return pk.Z.dense && pk.Z.n != 0 && msm_dense_presort_applicable(pk.Z, R.B, R.ms) && R.ms.entries_zb.n >= R.B * max_entries + 1 &&
           R.ms.bucket_start_zb.n >= R.B * (nb + 1);
}
// The stream the transforms of an ahead-prepared batch go to: the COPY stream, behind the batch's solve stage (event), so that the
// solve stage of the batch after it -- a chain of 163-457 short, latency-bound launches -- runs BESIDE these transforms instead of in
// front of them.  (On one stream the chain solve -> transforms was serial, and since the quotient walk's single-wave workgroups take
// every wave slot for its whole duration, the solve stage ended up alone on the device between the walk and the transforms:
// 10.6 ms of 96 per ChaCha20 step, 28 of 239 for AES-256 -- scripts/debug/timeline_step.py.)  ZKPROVE_NTT_STREAM=solve: the old order.
hipStream_t Algorithm::side_chain_stream(hipStream_t solve_stream, SolveSet &S) {
    static const bool on_solve = [] { const char *e = ZK_AB_ENV("ZKPROVE_NTT_STREAM"); return e && !strcmp(e, "solve"); }();
    if (on_solve) return solve_stream;
    ZK_HIP(hipStreamWaitEvent(res->copy_stream, S.solved, 0));
    return res->copy_stream;
}
void Algorithm::enqueue_ntt_sort(SolveSet &S, hipStream_t sn, int slot, hipEvent_t slot_free) {
    Resident &R = *res;
    const size_t B = R.B, N = (size_t)1 << prog.log_domain;
    ZK_HIP(hipEventRecord(R.ev[19], sn));
    dom.to_coset(S.sb.a.p, S.sb.b.p, B, sn);
    ZK_HIP(hipEventRecord(R.ev[20], sn));
    S.comb = comb_active();
    if (!S.comb) { products_rows_check(S.sb, N, B); fr_transpose_mul(S.sb.a.p, S.sb.b.p, S.sb.c.p, N, B, N, sn); }
    ZK_HIP(hipEventRecord(R.ev[17], sn));
    ZK_HIP(hipStreamWaitEvent(R.copy_stream, R.ev[17], 0));
    if (slot_free) ZK_HIP(hipStreamWaitEvent(R.copy_stream, slot_free, 0));      // the quotient MSM that walked this copy of the list is done
    if (S.comb && pat_slots() == 1) {
        // one pattern copy: written once the walk in front has read it -- enqueue_patterns_late, called by the pipelined entries right
        // after that walk has been enqueued (its event must be the one of THAT batch)
        S.pat_pending = true;
        S.ahead = true;
        S.zslot = 0;
        return;
    }
    if (S.comb) {
        // comb path: the products f = A*B are formed on the fly and only their bit planes (table indices) are written
        comb_patterns(pk.Zcomb, S.sb.a.p, S.sb.b.p, B, R.comb.pat[slot].p, R.copy_stream);
    } else {
        if (!msm_dense_presort(pk.Z, S.sb.c.p + z_offset, B, R.ms, R.copy_stream, slot)) throw ZkError("ahead sort: dense list buffers missing");
        R.ms.presorted = false;                 // the flag belongs to the msm_batch call that consumes the list; enqueue_main sets it
    }
    ZK_HIP(hipEventRecord(S.sorted, R.copy_stream));
    S.ahead = true;
    S.zslot = slot;
}
// One pattern slot: the patterns of an ahead-prepared batch, behind the walk of the batch in front of it (walk_done: that batch's zdone,
// nullptr for the first batch of a call).  Copy stream: behind this batch's transforms, which were enqueued there (ev[17] of its own).
void Algorithm::enqueue_patterns_late(SolveSet &S, hipEvent_t walk_done) {
    if (!S.pat_pending) return;
    Resident &R = *res;
    if (walk_done) ZK_HIP(hipStreamWaitEvent(R.copy_stream, walk_done, 0));
    comb_patterns(pk.Zcomb, S.sb.a.p, S.sb.b.p, R.B, R.comb.pat[0].p, R.copy_stream);
    ZK_HIP(hipEventRecord(S.sorted, R.copy_stream));
    S.pat_pending = false;
}
// NTT + MSM of one solved batch on stream st.  R.ev[3..9] bracket the phases.  before_msm (optional): an event the MSMs must
// wait for -- they overwrite the per-batch sums that the previous batch's tail (on another stream) still reads.
void Algorithm::enqueue_main(SolveSet &S, hipStream_t st, hipEvent_t before_msm, hipStream_t tail) {
    Resident &R = *res;
    const size_t B = R.B;
    ZK_HIP(hipEventRecord(R.ev[3], st));
    const size_t N = (size_t)1 << prog.log_domain;
    R.last_ahead = S.ahead;
    R.last_g2_side = false;
    if (S.ahead) {
        // transforms, products and sort are under way (enqueue_ntt_sort): wire MSMs now, the quotient MSM as soon as its list is sorted
        S.ahead = false;
        ZK_HIP(hipEventRecord(R.ev[4], st));
        if (before_msm) ZK_HIP(hipStreamWaitEvent(st, before_msm, 0));
        msm_batch(pk.A, S.sb.wires.p, B, R.ms, R.m_ar.p, st);
        msm_batch(pk.B1, S.sb.wires.p, B, R.ms, R.m_bs1.p, st);
        ZK_HIP(hipEventRecord(R.ev[12], st));
        ZK_HIP(hipStreamWaitEvent(tail, R.ev[12], 0));
        enqueue_tail_early(S, tail);
        ZK_HIP(hipEventRecord(R.ev[5], st));                 // (re-recorded on the copy stream below when the G2 MSM runs there)
        // The G2 MSM needs the wires only and feeds the tail only.  With wire combs its chain is long and partly latency-bound (two walks,
        // four folds): it CAN run on the copy stream with scratch of its own, beside the G1 chain of the main stream.
        // (Measured, same box: AES-256 4 113 proofs/s with it, 4 199 without -- the step is bound by VALU work, not by the main stream's
        // length, and the extra stream only adds contention.  Kept as an opt-in: ZKPROVE_G2_SIDE=1.)
        static const bool g2_side = [] { const char *e = ZK_AB_ENV("ZKPROVE_G2_SIDE"); return e && e[0] == '1'; }();
        const bool b2_side = g2_side && R.ms_b2.entries.n != 0;
        R.last_g2_side = b2_side;
        hipStream_t g2s = b2_side ? R.copy_stream : st;
        if (b2_side) {
            ZK_HIP(hipStreamWaitEvent(g2s, R.ev[3], 0));
            if (before_msm) ZK_HIP(hipStreamWaitEvent(g2s, before_msm, 0));
            ZK_HIP(hipEventRecord(R.ev[5], g2s));
        }
        msm_batch(pk.B2, S.sb.wires.p, B, b2_side ? R.ms_b2 : R.ms, R.m_bs2.p, g2s);
        ZK_HIP(hipEventRecord(R.ev[6], g2s));
        ZK_HIP(hipStreamWaitEvent(tail, R.ev[6], 0));
        launch_finalize_g2(R.m_bs2.p, pk.g2_misc.p, pk.delta2_tab.p, S.s.p, S.out.p, B, tail);
        msm_batch(pk.K, S.sb.wires.p, B, R.ms, R.m_k.p, st);
        ZK_HIP(hipStreamWaitEvent(st, S.sorted, 0));
        R.used_comb = S.comb;
        if (S.comb) {
            // the walk here; its fold (a latency-bound chain of doublings per proof) on the tail stream, in front of the tail's last
            // stage, which is its only reader -- the main stream goes straight on to the next batch's MSMs
            static const bool fold_on_tail = [] { const char *e = ZK_AB_ENV("ZKPROVE_COMB_FOLD_TAIL"); return !(e && e[0] == '0'); }();
            comb_walk(pk.Zcomb, R.comb.pat[S.zslot].p, B, R.comb, st, &R.ev[8]);
            if (fold_on_tail) {
                ZK_HIP(hipEventRecord(S.zdone, st));
                ZK_HIP(hipStreamWaitEvent(tail, S.zdone, 0));
                comb_fold(pk.Zcomb, B, R.comb, R.m_z.p, tail);
            } else comb_fold(pk.Zcomb, B, R.comb, R.m_z.p, st);
        } else {
            R.ms.presorted = true;
            R.ms.zslot = S.zslot;
            msm_batch(pk.Z, S.sb.c.p + z_offset, B, R.ms, R.m_z.p, st, &R.ev[8]);
        }
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipEventRecord(R.ev[13], st));
        ZK_HIP(hipEventRecord(S.zdone, st));
        return;
    }
    // A small batch is a set of latency-bound kernel chains: independent chains go to the otherwise idle streams (the transforms of b
    // and the K MSM to the solve stream, the G2 MSM to the copy stream) -- never to new ones, see resident_reserve
    const bool side = B <= side_max_b();      // (with tables a batch of >= 16 proofs takes the comb for its quotient MSM and the wire combs inside these chains)
    hipStream_t sk = R.stream2;
    if (side) {
        // the G2 chain first: MSM, blinding s*delta2, Fp2 inversion -- the longest of the tail's chains (5 ms for one AES proof), and on
        // a one-proof batch the HOST needs a millisecond to submit the launches of the other chains
        {
            hipStream_t g2s = R.copy_stream;      // idle in single-chunk calls
            ZK_HIP(hipStreamWaitEvent(g2s, R.ev[3], 0));
            if (before_msm) ZK_HIP(hipStreamWaitEvent(g2s, before_msm, 0));
            ZK_HIP(hipEventRecord(R.ev[5], g2s));
            msm_batch(pk.B2, S.sb.wires.p, B, R.ms_g2, R.m_bs2.p, g2s);
            if (split_world == 1)
                launch_finalize_g2(R.m_bs2.p, pk.g2_misc.p, pk.delta2_tab.p, S.s.p, S.out.p, B, g2s);
            ZK_HIP(hipEventRecord(R.ev[6], g2s));               // (the tail stream joins it below, behind its own chain)
            R.last_g2_side = true;                // phase times: msm_g1 = the main stream from its first MSM to the end (collect_times)
        }
        ZK_HIP(hipStreamWaitEvent(sk, R.ev[3], 0));
        dom.dif(S.sb.b.p, B, true, nullptr, sk);
        dom.dit(S.sb.b.p, B, false, dom.coset_pre.p, sk);
        ZK_HIP(hipEventRecord(R.ev[15], sk));
        if (before_msm) ZK_HIP(hipStreamWaitEvent(sk, before_msm, 0));
        msm_batch(pk.K, S.sb.wires.p, B, R.ms_k, R.m_k.p, sk);
        ZK_HIP(hipEventRecord(R.ev[16], sk));
        // Ar and Bs1 need the wires only: their MSMs and the 4 ms of dependent doublings of s*Ar, r*Bs1 start on the tail stream
        // at once, beside the transforms
        ZK_HIP(hipStreamWaitEvent(tail, R.ev[3], 0));
        msm_batch(pk.A, S.sb.wires.p, B, R.ms_a, R.m_ar.p, tail);
        if (comb_active() && !has_commit) {
            // with tables the main stream carries the call's longest chain -- transforms of a, patterns, quotient walk, fold -- and starts it
            // at once; the B1 sum (conditional sums only for ChaCha20) follows A on the tail stream: 64 proofs 10.97 -> 10.60 ms, 16: 5.98 ->
            // 5.59.  NOT for the AES programs: their wire MSMs are chains of ~15 short kernels each, and whatever of them is still queued when
            // the walk starts waits for wave slots under it (the walk's single-wave workgroups take them all): 31.4 -> 33.0 ms at 64 proofs
            // with the walk first (profiles/r05g_timeline_aes_64.txt) -- there the chains go first and the walk last
            if (before_msm) ZK_HIP(hipStreamWaitEvent(tail, before_msm, 0));
            msm_batch(pk.B1, S.sb.wires.p, B, R.ms, R.m_bs1.p, tail);
            ZK_HIP(hipEventRecord(R.ev[12], tail));
        } else {
            if (before_msm) ZK_HIP(hipStreamWaitEvent(st, before_msm, 0));
            msm_batch(pk.B1, S.sb.wires.p, B, R.ms, R.m_bs1.p, st);            // beside the A MSM; the main stream's own chain is the shortest
            ZK_HIP(hipEventRecord(R.ev[12], st));
            ZK_HIP(hipStreamWaitEvent(tail, R.ev[12], 0));
        }
        if (split_world == 1) enqueue_tail_early(S, tail);
        dom.dif(S.sb.a.p, B, true, nullptr, st);
        dom.dit(S.sb.a.p, B, false, dom.coset_pre.p, st);
    }
    const bool comb = comb_active();
    // A mid-size batch on the comb path is one call's latency: its transforms (a third of the chain in front of the walk) go to the copy
    // stream with the pattern extraction behind them, BESIDE the wire MSMs of the main stream, which only need the wires (ChaCha20, 128
    // proofs: 17.8 -> ... ms per call).  Large batches keep the transforms in front (VALU-bound either way, and the single-stream pass
    // of bench.py reads the transforms' own duration from ev[3] .. ev[4]).
    const bool ntt_beside = comb && !side && B <= 256;
    if (!side) {
        if (ntt_beside) {
            ZK_HIP(hipStreamWaitEvent(R.copy_stream, R.ev[3], 0));
            dom.to_coset(S.sb.a.p, S.sb.b.p, B, R.copy_stream);
        } else
            dom.to_coset(S.sb.a.p, S.sb.b.p, B, st);
    }
    ZK_HIP(hipEventRecord(R.ev[4], st));
    if (before_msm) ZK_HIP(hipStreamWaitEvent(st, before_msm, 0));
    bool presorted = false;
    R.used_comb = comb;
    if (!side) {
        // the quotient MSM's scalars f = A*B are complete: its bucket sort (LDS / memory bound) -- or, on the comb path, the extraction
        // of its bit-plane patterns -- runs on the copy stream beside the wire MSMs (multiplier bound) and is joined just before the walk
        if (!comb) { products_rows_check(S.sb, N, B); fr_transpose_mul(S.sb.a.p, S.sb.b.p, S.sb.c.p, N, B, N, st); }
        ZK_HIP(hipEventRecord(R.ev[17], st));
        if (!ntt_beside) ZK_HIP(hipStreamWaitEvent(R.copy_stream, R.ev[17], 0));
        if (comb) { comb_patterns(pk.Zcomb, S.sb.a.p, S.sb.b.p, B, R.comb.pat[0].p, R.copy_stream); presorted = true; }
        else presorted = msm_dense_presort(pk.Z, S.sb.c.p + z_offset, B, R.ms, R.copy_stream);
        if (presorted) ZK_HIP(hipEventRecord(R.ev[18], R.copy_stream));
        msm_batch(pk.A, S.sb.wires.p, B, R.ms, R.m_ar.p, st);
        msm_batch(pk.B1, S.sb.wires.p, B, R.ms, R.m_bs1.p, st);
        // Ar and Bs1 only need these two sums: their blinding and the two variable-base multiplications s*Ar, r*Bs1 (4 ms of
        // dependent doublings whatever the batch size) start now on the tail stream, under the remaining MSMs
        ZK_HIP(hipEventRecord(R.ev[12], st));
        ZK_HIP(hipStreamWaitEvent(tail, R.ev[12], 0));
        if (split_world == 1) enqueue_tail_early(S, tail);
    }
    // the G2 sum next, so that its blinding and affine conversion also run under the two largest MSMs.  (A small batch is a set of
    // latency-bound chains: its G2 chain -- MSM and tail, which used to queue behind the G1 tail's two variable-base multiplications:
    // one Prove() call 7.4 -> 6.5 ms -- went to the copy stream at the top.)
    if (!side) {
        ZK_HIP(hipEventRecord(R.ev[5], st));
        msm_batch(pk.B2, S.sb.wires.p, B, R.ms, R.m_bs2.p, st);
        ZK_HIP(hipEventRecord(R.ev[6], st));
        ZK_HIP(hipStreamWaitEvent(tail, R.ev[6], 0));
        if (split_world == 1)
            launch_finalize_g2(R.m_bs2.p, pk.g2_misc.p, pk.delta2_tab.p, S.s.p, S.out.p, B, tail);
    } else {
        ZK_HIP(hipStreamWaitEvent(tail, R.ev[6], 0));
    }
    if (!side) msm_batch(pk.K, S.sb.wires.p, B, R.ms, R.m_k.p, st);
    // sum_k h_k Z_k = sum_i (a_i b_i) Ze_i + (a part linear in the wires that sits in K); c only serves as the buffer of the
    // instance-major products
    if (side) {
        ZK_HIP(hipStreamWaitEvent(st, R.ev[15], 0));      // b on the coset comes from the side stream
        if (comb) comb_patterns(pk.Zcomb, S.sb.a.p, S.sb.b.p, B, R.comb.pat[0].p, st);
        else {
            products_rows_check(S.sb, N, B);
            fr_transpose_mul(S.sb.a.p, S.sb.b.p, S.sb.c.p, N, B, N, st);
        }
    }
    if (presorted) ZK_HIP(hipStreamWaitEvent(st, R.ev[18], 0));
    // (65 .. 256 proofs with tables: the copy stream, which carried transforms and patterns, is idle now and takes the upper planes' fold
    // beside the lower planes' walk -- ChaCha20, 128 proofs: 16.4 -> ... ms per call)
    // (16 .. 64 / 128 proofs, chains side by side: the solve stream -- K chain and the transforms of b, long done when the walk starts -- takes it)
    if (comb && side) comb_msm_split(pk.Zcomb, R.comb.pat[0].p, B, R.comb, R.m_z.p, st, sk, &R.ev[21], &R.ev[8]);
    else if (comb && ntt_beside) comb_msm_split(pk.Zcomb, R.comb.pat[0].p, B, R.comb, R.m_z.p, st, R.copy_stream, &R.ev[21], &R.ev[8]);
    else if (comb) comb_msm(pk.Zcomb, R.comb.pat[0].p, B, R.comb, R.m_z.p, st, &R.ev[8]);
    else msm_batch(pk.Z, S.sb.c.p + z_offset, B, R.ms, R.m_z.p, st, &R.ev[8]);
    ZK_HIP(hipGetLastError());
    if (side) {                                                  // the batch's sums are complete only when the side streams are done too
        ZK_HIP(hipStreamWaitEvent(st, R.ev[6], 0));
        ZK_HIP(hipStreamWaitEvent(st, R.ev[16], 0));
    }
    ZK_HIP(hipEventRecord(R.ev[13], st));
}

// Tail of one batch, first half: Ar, Bs1 (blinded, affine), -rs*delta, s*Ar, r*Bs1.  1024-thread kernels with long dependent
// chains (two variable-base scalar multiplications per proof): latency-bound, so they run on their own stream.
void Algorithm::enqueue_tail_early(SolveSet &S, hipStream_t st) {
    Resident &R = *res;
    const size_t B = R.B;
    if (B <= TAIL_TREE_MAX_B)
        hipLaunchKernelGGL(k_finalize_stage1_small, dim3((unsigned)B, 3), dim3(16), 0, st, R.m_ar.p, R.m_bs1.p, pk.g1_misc.p, pk.delta_tab.p, S.r.p, S.s.p, R.ar_aff.p,
                           R.bs1_aff.p, R.f_t.p, B);
    else
        hipLaunchKernelGGL(k_finalize_stage1, dim3(ceil_div(B, 64), 3), dim3(64), 0, st, R.m_ar.p, R.m_bs1.p, pk.g1_misc.p, pk.delta_tab.p, S.r.p, S.s.p,
                           R.ar_aff.p, R.bs1_aff.p, R.f_t.p, B);
    static const bool stage2_pipe = [] { const char *e = ZK_AB_ENV("ZKPROVE_STAGE2_PIPE"); return !(e && e[0] == '0'); }();
    if (B <= 128 && stage2_pipe)
        hipLaunchKernelGGL(k_finalize_stage2_pipe<4>, dim3(ceil_div(B, 32), 2), dim3(320), 0, st, R.ar_aff.p, R.bs1_aff.p, S.r.p, S.s.p, R.f_u.p, R.f_v.p, B);
    else
        hipLaunchKernelGGL(k_finalize_stage2, dim3(ceil_div(B, 64), 2), dim3(64), 0, st, R.ar_aff.p, R.bs1_aff.p, S.r.p, S.s.p, R.f_u.p, R.f_v.p, B);
    ZK_HIP(hipGetLastError());
}
// last part, once every sum of the batch is complete: Krs, canonical outputs
void Algorithm::enqueue_finalize(SolveSet &S, hipStream_t st) {
    Resident &R = *res;
    const size_t B = R.B;
    hipLaunchKernelGGL(k_finalize_stage3, dim3(ceil_div(B, 64)), dim3(64), 0, st, R.m_k.p, R.m_z.p, R.f_t.p, R.f_u.p, R.f_v.p, R.ar_aff.p, S.out.p, B);
    hipLaunchKernelGGL(k_store_commit, dim3(ceil_div(B, 64)), dim3(64), 0, st, S.com_aff.p, S.pok_aff.p, S.out.p, B, has_commit ? 1 : 0);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipEventRecord(R.ev[7], st));
}

void Algorithm::collect_times(SolveSet &S, PhaseTimes *times) {
    Resident &R = *res;
    float t01, t12, t23, t34, t45, t56, t67;
    ZK_HIP(hipEventElapsedTime(&t01, S.t0, S.t1)); ZK_HIP(hipEventElapsedTime(&t12, S.t1, S.t2)); ZK_HIP(hipEventElapsedTime(&t23, S.t2, S.t3));
    ZK_HIP(hipEventElapsedTime(&t34, R.ev[3], R.ev[4])); ZK_HIP(hipEventElapsedTime(&t45, R.ev[4], R.ev[5]));
    float t6e;
    ZK_HIP(hipEventElapsedTime(&t56, R.ev[5], R.ev[6])); ZK_HIP(hipEventElapsedTime(&t6e, R.ev[6], R.ev[13])); ZK_HIP(hipEventElapsedTime(&t67, R.ev[13], R.ev[7]));
    t45 += t6e;
    if (R.last_g2_side) ZK_HIP(hipEventElapsedTime(&t45, R.ev[4], R.ev[13]));      // the G2 MSM ran beside the G1 chain: msm_g1 = the main stream from its first MSM to the end of the walk
    ZK_HIP(hipEventElapsedTime(&times->z_segments, R.ev[8], R.ev[9]));
    if (R.last_ahead) ZK_HIP(hipEventElapsedTime(&t34, R.ev[19], R.ev[20]));      // the transforms ran on the copy stream (side_chain_stream): start to end, INCLUDING the time they wait for wave slots under a quotient walk
    times->solve = t01 + t23; times->commit = t12; times->ntt = t34; times->msm_g1 = t45; times->msm_g2 = t56; times->finalize = t67;
}

void Algorithm::resident_prove(PhaseTimes *times) {
    DeviceScope on_device(device);
    comb_top_up();
    Resident &R = *res;
    hipStream_t st = R.stream;
    R.last_set = 0;
    SolveSet &S0 = R.set[0];
    // one-call batches of the long programs solve runs of wide levels in one launch each, behind a grid barrier with a bounded poll
    // (solver.hip, k_solve_wide_run); three timeouts in an algorithm's life and it keeps the per-level launches
    S0.sb.allow_wide_run = wide_run_strikes < 3;
    for (int attempt = 0;; attempt++) {
        enqueue_solve(S0, R.ms, st);
        enqueue_main(S0, st, nullptr, R.stream3);
        ZK_HIP(hipEventRecord(R.ev[10], st));
        ZK_HIP(hipStreamWaitEvent(R.stream3, R.ev[10], 0));
        enqueue_finalize(S0, R.stream3);
        ZK_HIP(hipStreamSynchronize(st));
        ZK_HIP(hipStreamSynchronize(R.stream3));
        if (S0.sb.sync_arrivals == 0 || attempt) break;                 // no wide-run launch in this call / this was the repeat
        uint32_t herr[8] = {0};
        ZK_HIP(hipMemcpy(herr, S0.sb.err.p, std::min<size_t>(R.B, 8) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        bool retry = false;
        for (size_t i = 0; i < R.B && i < 8; i++) retry = retry || herr[i] == SERR_RETRY;
        if (!retry) break;
        wide_run_strikes++;
        fprintf(stderr, "zkprove: solver grid barrier timed out (device busy): the call is repeated with per-level launches\n");
        ZK_HIP(hipStreamSynchronize(R.stream2));
        ZK_HIP(hipStreamSynchronize(R.copy_stream));
        S0.sb.allow_wide_run = false;
    }
    S0.sb.allow_wide_run = false;
    if (times) {
        collect_times(R.set[0], times);
        ZK_HIP(hipEventElapsedTime(&times->total, R.set[0].t0, R.ev[7]));
    }
}

// Memory-pressure valve: the comb tables are an optimisation, the working sets are not.  When a working-set allocation fails while this
// algorithm holds comb tables (a third circuit with large batches on the same GPU), the tables and their scratch are given back and the
// quotient / wire MSMs of this algorithm return to the table-free bucket pipeline.
bool Algorithm::comb_drop_tables() {
    if (comb_table_bytes() == 0) return false;
    Resident &R = *res;
    ZK_HIP(hipDeviceSynchronize());
    (void)hipGetLastError();
    const size_t mib = pk.Zcomb.table.bytes() >> 20;
    pk.Zcomb = CombBases();
    R.comb = CombScratch();
    R.comb_cap = 0;
    for (MsmBasesG1 *mb : {&pk.A, &pk.B1, &pk.K, &pk.ped, &pk.ped_sigma}) { mb->wcomb.clear(); mb->wide_mask.release(); }
    pk.B2.wcomb.clear(); pk.B2.wide_mask.release();
    wire_comb_on = false;
    R.ms_a = MsmScratch(); R.ms_k = MsmScratch(); R.ms_g2 = MsmScratch();      // (sized for side_max_b() instances: 64 again without the wire combs)
    R.side_cap = 0;
    comb_dropped = true;
    fprintf(stderr, "zkprove: out of device memory: comb tables (%zu MiB) released, this algorithm continues on the bucket pipeline\n", mib);
    return true;
}
// Tables that went under pressure are not gone for good: when this algorithm is next asked to prove -- the pressure has passed, every
// circuit's working sets exist -- the budget (comb_prepare: what is free beyond this algorithm's own needs) is consulted once more and
// builds what fits, possibly a smaller clump size.  Twice per algorithm at most.
void Algorithm::comb_second_chance() {
    if (!comb_dropped || comb_rebuilds >= 2) return;
    comb_dropped = false;
    comb_rebuilds++;
    pk.comb_tried = false;
    comb_topped = false;
    wire_comb_tried = false;
}
bool Algorithm::comb_release() {
    if (!comb_drop_tables()) return false;
    resident_reserve(res->cap);                    // the bucket pipeline's buffers for the reserved batch size
    return true;
}
// A third circuit with large batches on a GPU whose memory two circuits' tables and working sets already fill: its working sets do not
// fit and it has no tables of its own to give back.  The tables of the algorithms that are idle at that moment go instead (they continue
// on the bucket pipeline; the log line says so) -- a request must not fail for the sake of another circuit's optimisation.
bool (*Algorithm::neighbours_release)(Algorithm *self) = nullptr;
bool (*Algorithm::peers_visit)(Algorithm *self, const std::function<void(Algorithm *)> &fn) = nullptr;
int (*Algorithm::loaded_on_device)(int device) = nullptr;
bool Algorithm::release_comb_if_idle() {
    std::unique_lock<std::mutex> lk(mu, std::try_to_lock);
    if (!lk || !res) return false;
    DeviceScope on_device(device);
    try {
        return comb_release();
    } catch (...) {
        (void)hipGetLastError();
        return false;
    }
}
size_t Algorithm::comb_table_bytes() const {
    size_t b = pk.Zcomb.table.bytes();
    for (const MsmBasesG1 *mb : {&pk.A, &pk.B1, &pk.K, &pk.ped, &pk.ped_sigma})
        for (const auto &c : mb->wcomb) b += c->table.bytes();
    for (const auto &c : pk.B2.wcomb) b += c->table.bytes();
    return b;
}
bool Algorithm::relieve_pressure(bool rereserve) {
    if (rereserve ? comb_release() : comb_drop_tables()) return true;
    return neighbours_release && neighbours_release(this);
}

// the second working set (and the separate scratch of its commitment MSMs) exists only once a pipelined entry is used
void Algorithm::ensure_second_set(int nsets) {
    // every allocation below is "grow if smaller": after a failure the memory-pressure valve (own comb tables, then one idle
    // neighbour's per round) makes room and the whole reservation is simply repeated
    for (int attempt = 0;; attempt++) {
        try {
            ensure_second_set_once(nsets);
            return;
        } catch (const ZkError &) {
            (void)hipGetLastError();
            if (attempt >= 5 || !relieve_pressure(true)) throw;
        }
    }
}
void Algorithm::ensure_second_set_once(int nsets) {
    Resident &R = *res;
    for (int i = 1; i < nsets; i++) {
        if (R.set[i].ready) continue;
        R.set[i].reserve(prog, R.cap, own_c_rows(R.cap));
        R.set[i].sb.B = R.B;
        SolveSet &a = R.set[0], &b = R.set[i];
        ZK_HIP(hipMemcpy(b.sb.inputs.p, a.sb.inputs.p, prog.n_inputs * R.B * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        ZK_HIP(hipMemcpy(b.sb.raw.p, a.sb.raw.p, SOLVER_RAW_STRIDE * R.B, hipMemcpyDeviceToDevice));
        b.sb.raw_alg = a.sb.raw_alg;
        ZK_HIP(hipMemcpy(b.r.p, a.r.p, R.B * sizeof(U256), hipMemcpyDeviceToDevice));
        ZK_HIP(hipMemcpy(b.s.p, a.s.p, R.B * sizeof(U256), hipMemcpyDeviceToDevice));
        ZK_HIP(hipMemcpy(b.rnd.p, a.rnd.p, R.B * sizeof(U256), hipMemcpyDeviceToDevice));
    }
    if (has_commit && (R.ms2.entries.n == 0 || R.ms2.B < R.cap)) {      // (again when the capacity grew: a small batch may come first)
        R.ms2.reserve(R.cap, pk.ped.n * (size_t)pk.ped.W, (size_t)1 << (pk.ped.c - 1), msm_part_stride(pk.ped.n, pk.ped.c));
        R.ms2.reserve_compact(R.cap, pk.ped.n);
        R.ms2.reserve_sums(R.cap, (size_t)1 << (pk.ped.c - 1), 1);
        wire_comb_reserve(R.ms2, false, R.cap);
    }
}

// nsteps batches back to back (the same resident inputs are re-proved each time), the SOLVE stage of batch k+1 overlapped
// with the NTT + MSM stage of batch k.  times (optional) = phase times of the LAST batch; total = whole call / nsteps.
void Algorithm::resident_prove_pipelined(int nsteps, PhaseTimes *times) {
    DeviceScope on_device(device);
    comb_top_up();
    Resident &R = *res;
    if (nsteps <= 0) return;
    hipStream_t M = R.stream, S = R.stream2, F = R.stream3;
    const bool ahead = ahead_eligible();
    // Ahead mode prepares batches TWO ahead (three working sets): solve stage, transforms and products of batch k+2 run while batch k is
    // on the main stream, its sort as soon as batch k's quotient MSM has released the list copy -- the side chain solve -> transforms ->
    // sort only gets the issue slots the MSMs leave, and with a single batch of lead it finished after the walk it should hide under
    // (the main stream waited ~20 ms per step).  ZKPROVE_PIPE_DEPTH=1 keeps one batch of lead.
    static const int depth_env = [] { const char *e = getenv("ZKPROVE_PIPE_DEPTH"); return e ? atoi(e) : 2; }();
    const int D = (ahead && depth_env >= 2) ? 2 : 1, NS = D + 1;
    ensure_second_set(NS);
    R.last_set = (nsteps - 1) % NS;
    ZK_HIP(hipEventRecord(R.ev[0], M));
    ZK_HIP(hipStreamWaitEvent(S, R.ev[0], 0));
    ZK_HIP(hipStreamWaitEvent(F, R.ev[0], 0));
    auto prepare = [&](int j, hipEvent_t set_free, hipEvent_t slot_free) {      // solve stage of batch j on the solve stream, its ahead work (transforms, patterns / sort) on the copy stream
        SolveSet &nx = R.set[j % NS];
        nx.pat_pending = false;
        if (set_free) ZK_HIP(hipStreamWaitEvent(S, set_free, 0));
        enqueue_solve(nx, has_commit ? R.ms2 : R.ms, S);
        ZK_HIP(hipEventRecord(nx.solved, S));
        if (ahead) enqueue_ntt_sort(nx, side_chain_stream(S, nx), j & 1, slot_free);
    };
    for (int j = 0; j < D && j < nsteps; j++) prepare(j, nullptr, nullptr);
    enqueue_patterns_late(R.set[0], nullptr);                  // (one pattern slot: batch 0's patterns have no walk in front of them)
    for (int k = 0; k < nsteps; k++) {
        SolveSet &cur = R.set[k % NS];
        ZK_HIP(hipStreamWaitEvent(M, cur.solved, 0));
        enqueue_main(cur, M, k ? R.ev[11] : nullptr, F);
        ZK_HIP(hipEventRecord(R.ev[10], M));
        ZK_HIP(hipStreamWaitEvent(F, R.ev[10], 0));
        enqueue_finalize(cur, F);
        ZK_HIP(hipEventRecord(R.ev[11], F));
        ZK_HIP(hipEventRecord(cur.consumed, F));
        // one pattern slot: batch k + 1's patterns behind THIS walk (its event is recorded now), in front of batch k + 2's transforms on
        // the copy stream; with one batch of lead batch k + 1 is only prepared below, and its patterns follow there
        if (k + 1 < nsteps) enqueue_patterns_late(R.set[(k + 1) % NS], cur.zdone);
        const int j = k + D;
        if (j < nsteps) {
            // set j % NS last held batch j - NS (= k - 1 for D = 2, k for D = 1 ... no: k - 1 in both): free once that batch's tail is done;
            // list copy j & 1 was last walked by batch j - 2
            SolveSet &nx = R.set[j % NS];
            const bool reused = j >= NS;
            hipEvent_t slot_free = (ahead && j >= 2) ? R.set[(j - 2) % NS].zdone : nullptr;
            prepare(j, reused ? nx.consumed : nullptr, slot_free);
        }
        if (k + 1 < nsteps) enqueue_patterns_late(R.set[(k + 1) % NS], cur.zdone);
    }
    ZK_HIP(hipStreamSynchronize(S));
    ZK_HIP(hipStreamSynchronize(R.copy_stream));
    ZK_HIP(hipStreamSynchronize(M));
    ZK_HIP(hipStreamSynchronize(F));
    if (times) {
        collect_times(R.set[(nsteps - 1) % NS], times);
        float tot;
        ZK_HIP(hipEventElapsedTime(&tot, R.ev[0], R.ev[7]));
        times->total = tot / nsteps;
    }
}

// ---- split mode -----------------------------------------------------------------------------------------------------
// record per proof: Ar | Bs1 | K | Z partial sums as G1 affine (Montgomery, 64 B each), Bs2 partial as G2 affine (128 B)
struct SplitRecord { G1Affine ar, bs1, k, z; G2Affine bs2; };
static_assert(sizeof(SplitRecord) == 384, "split record layout");
__global__ __launch_bounds__(64) void k_split_pack(const G1XYZZ *__restrict__ m_ar, const G1XYZZ *__restrict__ m_bs1, const G1XYZZ *__restrict__ m_k,
                                                   const G1XYZZ *__restrict__ m_z, const G2XYZZ *__restrict__ m_bs2, SplitRecord *__restrict__ out, size_t B) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int task = blockIdx.y;
    if (task == 0) out[b].ar = m_ar[b].to_affine();
    else if (task == 1) out[b].bs1 = m_bs1[b].to_affine();
    else if (task == 2) out[b].k = m_k[b].to_affine();
    else if (task == 3) out[b].z = m_z[b].to_affine();
    else out[b].bs2 = m_bs2[b].to_affine();
}
// sums of the gathered partials: in [world][B] records -> the five sums of the ordinary pipeline
__global__ __launch_bounds__(64) void k_split_sum(const SplitRecord *__restrict__ in, int world, G1XYZZ *__restrict__ m_ar, G1XYZZ *__restrict__ m_bs1,
                                                  G1XYZZ *__restrict__ m_k, G1XYZZ *__restrict__ m_z, G2XYZZ *__restrict__ m_bs2, size_t B) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int task = blockIdx.y;
    if (task < 4) {
        G1XYZZ acc = G1XYZZ::inf();
        for (int r = 0; r < world; r++) {
            const SplitRecord &rec = in[(size_t)r * B + b];
            acc.madd(task == 0 ? rec.ar : task == 1 ? rec.bs1 : task == 2 ? rec.k : rec.z);
        }
        (task == 0 ? m_ar : task == 1 ? m_bs1 : task == 2 ? m_k : m_z)[b] = acc;
    } else {
        G2XYZZ acc = G2XYZZ::inf();
        for (int r = 0; r < world; r++) acc.madd(in[(size_t)r * B + b].bs2);
        m_bs2[b] = acc;
    }
}
void Algorithm::split_partials(uint8_t *out384) {
    std::lock_guard<std::mutex> lk(mu);
    DeviceScope on_device(device);
    if (!res) throw ZkError("split_partials: nothing uploaded");
    Resident &R = *res;
    const size_t B = R.B;
    hipStream_t st = R.stream;
    R.last_set = 0;
    enqueue_solve(R.set[0], R.ms, st);
    enqueue_main(R.set[0], st, nullptr, R.stream3);          // split_world > 1: the tail is not started
    // the sums are complete once the main stream AND the tail stream (which carries the A MSM of a small batch) are through
    ZK_HIP(hipEventRecord(R.ev[10], st));
    ZK_HIP(hipStreamWaitEvent(R.stream3, R.ev[10], 0));
    DevBuf<SplitRecord> rec(B);
    hipLaunchKernelGGL(k_split_pack, dim3(ceil_div(B, 64), 5), dim3(64), 0, R.stream3, R.m_ar.p, R.m_bs1.p, R.m_k.p, R.m_z.p, R.m_bs2.p, rec.p, B);
    ZK_HIP(hipMemcpyAsync(out384, rec.p, B * sizeof(SplitRecord), hipMemcpyDeviceToHost, R.stream3));
    ZK_HIP(hipStreamSynchronize(R.stream3));
    ZK_HIP(hipStreamSynchronize(st));
}
void Algorithm::split_finish(int world, const uint8_t *gathered) {
    std::lock_guard<std::mutex> lk(mu);
    DeviceScope on_device(device);
    if (!res) throw ZkError("split_finish: nothing uploaded");
    Resident &R = *res;
    const size_t B = R.B;
    hipStream_t st = R.stream3;
    DevBuf<SplitRecord> rec((size_t)world * B);
    ZK_HIP(hipMemcpyAsync(rec.p, gathered, (size_t)world * B * sizeof(SplitRecord), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_split_sum, dim3(ceil_div(B, 64), 5), dim3(64), 0, st, rec.p, world, R.m_ar.p, R.m_bs1.p, R.m_k.p, R.m_z.p, R.m_bs2.p, B);
    SolveSet &S = R.set[0];
    enqueue_tail_early(S, st);
    launch_finalize_g2(R.m_bs2.p, pk.g2_misc.p, pk.delta2_tab.p, S.s.p, S.out.p, B, st);
    enqueue_finalize(S, st);
    ZK_HIP(hipStreamSynchronize(st));
}

void Algorithm::resident_download(ProofPoints *out, uint32_t *errs) {
    DeviceScope on_device(device);
    Resident &R = *res;
    SolveSet &S = R.set[R.last_set];
    ZK_HIP(hipMemcpy(out, S.out.p, R.B * sizeof(ProofPoints), hipMemcpyDeviceToHost));
    if (errs) ZK_HIP(hipMemcpy(errs, S.sb.err.p, R.B * sizeof(uint32_t), hipMemcpyDeviceToHost));
}

// batch proofs, chunked to max_chunk instances per device pass.  More than one chunk: the chunks are software-pipelined exactly as
// resident_prove_pipelined does with repeated passes (solve of chunk k+1 and tail of chunk k-1 under the NTT + MSM stage of chunk k),
// with the uploads / downloads on a copy stream, so callers of ProveBatch see the pipelined rate.
void Algorithm::prove_batch_raw(size_t batch, const uint8_t *raw, int raw_alg, const U256 *r, const U256 *s, const U256 *randomize, ProofPoints *out, uint32_t *errs) {
    if (!raw || raw_alg < 0) throw ZkError("prove_batch_raw: no records");
    prove_batch_impl(batch, nullptr, raw, raw_alg, r, s, randomize, out, errs, nullptr, nullptr, 0);
}
void Algorithm::prove_batch(size_t batch, const uint32_t *inputs, const U256 *r, const U256 *s, const U256 *randomize, ProofPoints *out,
                            uint32_t *errs, PhaseTimes *times, const WideInput *wide, size_t n_wide) {
    prove_batch_impl(batch, inputs, nullptr, -1, r, s, randomize, out, errs, times, wide, n_wide);
}
void Algorithm::prove_batch_impl(size_t batch, const uint32_t *inputs, const uint8_t *raw, int raw_alg, const U256 *r, const U256 *s, const U256 *randomize,
                                 ProofPoints *out, uint32_t *errs, PhaseTimes *times, const WideInput *wide, size_t n_wide) {
    std::unique_lock<std::mutex> lk(mu);
    last_use = ++g_use_clock;
    comb_second_chance();
    DeviceScope on_device(device);
    const size_t ni = prog.n_inputs;
    const size_t nchunks = (batch + max_chunk - 1) / max_chunk;
    if (n_wide && nchunks > 1) throw ZkError("prove_batch: wide witness elements are supported for single-chunk batches only");
    if (nchunks <= 1) {
        resident_upload(batch, inputs, r, s, randomize, wide, n_wide, raw, raw_alg);
        PhaseTimes t;
        resident_prove(&t);
        resident_download(out, errs);
        if (times) *times = t;
        return;
    }
    // more than one chunk: stream them from the caller's arrays
    struct ArrayFeed : ChunkFeed {
        size_t batch, chunk, ni, k = 0;
        const uint32_t *inputs; const U256 *r, *s, *rnd; ProofPoints *out; uint32_t *errs;
        const uint8_t *raw = nullptr; int raw_alg = -1;
        bool next(Chunk &c) override {
            const size_t off = k * chunk;
            if (off >= batch) return false;
            c.id = k++; c.count = std::min(chunk, batch - off);
            c.inputs = inputs ? inputs + off * ni : nullptr; c.r = r + off; c.s = s + off; c.rnd = rnd ? rnd + off : nullptr;
            c.raw = raw ? raw + off * SOLVER_RAW_STRIDE : nullptr; c.raw_alg = raw_alg;
            c.out = out + off; c.errs = errs ? errs + off : nullptr;
            return true;
        }
        void done(const Chunk &) override {}
    } feed;
    feed.batch = batch; feed.chunk = max_chunk; feed.ni = ni;
    feed.inputs = inputs; feed.r = r; feed.s = s; feed.rnd = randomize; feed.out = out; feed.errs = errs;
    feed.raw = raw; feed.raw_alg = raw_alg;
    lk.unlock();
    prove_stream(feed, times);
}

// Chunks of at most max_chunk instances, software-pipelined exactly as resident_prove_pipelined does with repeated passes (solve stage,
// transforms and the quotient MSM's ahead-prepared data of chunk k+1 and the tail of chunk k-1 under the MSM stage of chunk k), uploads
// and downloads on the copy stream.  The feed is asked for chunk k only after chunk k-NS has been handed back.
void Algorithm::prove_stream(ChunkFeed &feed, PhaseTimes *times) {
    std::lock_guard<std::mutex> lk(mu);
    last_use = ++g_use_clock;
    comb_second_chance();
    DeviceScope on_device(device);
    resident_reserve(max_chunk);
    comb_top_up();
    Resident &R = *res;
    // two working sets; three when chunks are prepared ahead (as in resident_prove_pipelined: the side chain of chunk k then runs beside
    // the MSMs of chunks k-2 and k-1)
    static const int depth_env = [] { const char *e = getenv("ZKPROVE_PIPE_DEPTH"); return e ? atoi(e) : 2; }();
    const size_t NS = (ahead_eligible() && depth_env >= 2) ? 3 : 2;
    ensure_second_set((int)NS);
    hipStream_t M = R.stream, S = R.stream2, F = R.stream3, C = R.copy_stream;
    ChunkFeed::Chunk held[3];
    // chunk k lives in set k % NS.  The host waits for ITS tail only (event), then copies on the null stream: none of the prover's
    // (non-blocking) streams is synchronised or gets the copies queued behind the ahead work of later chunks.
    auto download = [&](size_t k) {
        SolveSet &cur = R.set[k % NS];
        const ChunkFeed::Chunk &c = held[k % NS];
        ZK_HIP(hipEventSynchronize(cur.consumed));
        ZK_HIP(hipMemcpy(c.out, cur.out.p, c.count * sizeof(ProofPoints), hipMemcpyDeviceToHost));
        if (c.errs) ZK_HIP(hipMemcpy(c.errs, cur.sb.err.p, c.count * sizeof(uint32_t), hipMemcpyDeviceToHost));
        feed.done(c);
    };
    ZK_HIP(hipEventRecord(R.ev[0], M));
    // Submission order = resident_prove_pipelined's: main and tail of chunk k, THEN the solve stage of chunk k + D.  HIP multiplexes
    // the streams of all algorithms of the process over a few hardware queues per priority; when the solve stream and the tail stream
    // of this algorithm share one (they did with a second circuit loaded), whatever is submitted behind a chunk's tail waits for that
    // chunk's walk.  Submitted chunk by chunk (solve k, main k, tail k, solve k+1, ...) every solve stage then ran alone after the walk
    // before it: AES-256 ProveBatch at 91 % of the resident rate.  In this order a shared queue only moves the solve stage
    // from one window between walks to the next, which is still a window early.
    const size_t D = NS - 1;
    size_t fetched = 0;                     // chunks taken from the feed
    bool drained = false;
    auto set_batch = [&](size_t Bk) { R.B = Bk; R.ms.B = Bk; R.ms2.B = Bk; };
    auto prepare = [&]() -> bool {          // next chunk of the feed -> its set: upload, solve stage, ahead work
        ChunkFeed::Chunk c;
        if (drained || !feed.next(c)) { drained = true; return false; }
        if (c.count == 0 || c.count > max_chunk) throw ZkError("prove_stream: bad chunk size");
        const size_t j = fetched++;
        SolveSet &nx = R.set[j % NS];
        held[j % NS] = c;
        nx.pat_pending = false;
        set_batch(c.count);
        nx.sb.B = c.count;
        // inputs go up on the solve stream itself (the set is free: its previous chunk has been downloaded); the copy stream carries
        // the ahead-prepared transforms / patterns / sort of earlier chunks, which wait for other streams
        upload_set(nx, c.count, c.inputs, c.r, c.s, c.rnd, S, c.raw, c.raw_alg);
        enqueue_solve(nx, has_commit ? R.ms2 : R.ms, S);
        ZK_HIP(hipEventRecord(nx.solved, S));
        if (ahead_eligible()) enqueue_ntt_sort(nx, side_chain_stream(S, nx), (int)(j & 1), j >= 2 ? R.set[(j - 2) % NS].zdone : nullptr);
        return true;
    };
    // one pattern slot: the patterns of chunk n go behind the walk of chunk n - 1 (enqueue_patterns_late), with chunk n's own batch size
    auto late_patterns = [&](size_t n, hipEvent_t walk_done) {
        if (n >= fetched || !R.set[n % NS].pat_pending) return;
        set_batch(held[n % NS].count);
        enqueue_patterns_late(R.set[n % NS], walk_done);
    };
    size_t k = 0;
    try {
    for (size_t j = 0; j < D; j++)
        if (!prepare()) break;
    late_patterns(0, nullptr);
    for (; k < fetched; k++) {
        SolveSet &cur = R.set[k % NS];
        set_batch(held[k % NS].count);
        ZK_HIP(hipStreamWaitEvent(M, cur.solved, 0));
        enqueue_main(cur, M, k ? R.ev[11] : nullptr, F);
        ZK_HIP(hipEventRecord(cur.zdone, M));
        ZK_HIP(hipEventRecord(R.ev[10], M));
        ZK_HIP(hipStreamWaitEvent(F, R.ev[10], 0));
        enqueue_finalize(cur, F);
        ZK_HIP(hipEventRecord(R.ev[11], F));
        ZK_HIP(hipEventRecord(cur.consumed, F));
        late_patterns(k + 1, cur.zdone);        // (before the host blocks in download; in front of chunk k + 2's transforms on the copy stream)
        if (k >= 1) download(k - 1);            // frees set (k - 1) % NS == (k + D) % NS; at most NS chunks are away from the feed
        prepare();                              // chunk k + D
        late_patterns(k + 1, cur.zdone);        // (one chunk of lead: chunk k + 1 was prepared just now)
    }
    } catch (...) {
        // nothing may still read the caller's pinned chunk buffers or write this replica's sets when the error reaches the caller
        (void)hipStreamSynchronize(S); (void)hipStreamSynchronize(C); (void)hipStreamSynchronize(M); (void)hipStreamSynchronize(F);
        throw;
    }
    // k chunks were enqueued; all but the last are already back
    if (k) download(k - 1);
    ZK_HIP(hipStreamSynchronize(S));
    ZK_HIP(hipStreamSynchronize(C));
    ZK_HIP(hipStreamSynchronize(M));
    ZK_HIP(hipStreamSynchronize(F));
    if (k) R.last_set = (int)((k - 1) % NS);
    if (times && k) {
        PhaseTimes t;
        collect_times(R.set[(k - 1) % NS], &t);
        float tot;
        ZK_HIP(hipEventElapsedTime(&tot, R.ev[0], R.ev[7]));
        t.total = tot;
        *times = t;
    }
}

// parity-test hook: solved wires and quotient-stage products (A*B on the coset g<w>, natural order) of the last resident
// batch, canonical form; instances == nullptr: all B of them, else the n listed ones (wires_out [n][n_wires][32], h_out [n][N][32])
__global__ __launch_bounds__(256) void k_debug_gather(const Fr *__restrict__ src, size_t rows, size_t B, const uint32_t *__restrict__ inst, size_t n, Fr *__restrict__ out,
                                                      const Fr *__restrict__ mul) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const size_t k = i / rows, row = i % rows;
    const size_t b = inst ? inst[k] : k;
    Fr v = src[row * B + b];
    if (mul) v = v * mul[row * B + b];
    out[k * rows + row] = v.from_mont();
}
void Algorithm::debug_fetch(const uint32_t *instances, size_t n, uint8_t *wires_out, uint8_t *h_out) {
    DeviceScope on_device(device);
    Resident &R = *res;
    const size_t B = R.B, nw = prog.n_wires, N = (size_t)1 << prog.log_domain;
    if (!instances) n = B;
    DevBuf<uint32_t> dinst(n ? n : 1);
    if (instances) {
        for (size_t i = 0; i < n; i++) if (instances[i] >= B) throw ZkError("debug_fetch: instance out of range");
        ZK_HIP(hipMemcpy(dinst.p, instances, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    const SolverBuffers &sb = R.set[R.last_set].sb;
    auto fetch = [&](const Fr *src, const Fr *mul, size_t rows, uint8_t *out) {
        DevBuf<Fr> d(rows * n);
        hipLaunchKernelGGL(k_debug_gather, dim3(ceil_div(rows * n, 256)), dim3(256), 0, 0, src, rows, B, instances ? dinst.p : (const uint32_t *)nullptr, n, d.p, mul);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipMemcpy(out, d.p, rows * n * sizeof(Fr), hipMemcpyDeviceToHost));
    };
    if (wires_out) fetch(sb.wires.p, nullptr, nw, wires_out);
    // the products f = A*B on the coset: a and b hold the coset evaluations after every path (the bucket path's instance-major copy in
    // c is the same numbers transposed)
    if (h_out) fetch(sb.a.p, sb.b.p, N, h_out);
}

// -------------------------------------------------------------------------- serialisation
static void put_g1(std::vector<uint8_t> &o, const U256 &x, const U256 &y) {
    uint8_t b[32];
    static const U256 Z = {{0}};
    if (u256_cmp(x, Z) == 0 && u256_cmp(y, Z) == 0) {
        memset(b, 0, 32); b[0] = 0x40;
    } else {
        u256_to_be(x, b);
        b[0] |= u256_cmp(y, FP_HALF) > 0 ? 0xC0 : 0x80;
    }
    o.insert(o.end(), b, b + 32);
}
std::vector<uint8_t> proof_serialize(const ProofPoints &p, bool has_commit) {
    std::vector<uint8_t> o;
    put_g1(o, p.ar[0], p.ar[1]);
    {   // G2: X.A1 | X.A0, flag from y (A1 first unless zero)
        uint8_t b[64];
        static const U256 Z = {{0}};
        bool inf = true;
        for (int k = 0; k < 4; k++) if (u256_cmp(p.bs[k], Z)) inf = false;
        if (inf) { memset(b, 0, 64); b[0] = 0x40; }
        else {
            u256_to_be(p.bs[1], b); u256_to_be(p.bs[0], b + 32);
            const bool largest = u256_cmp(p.bs[3], Z) == 0 ? u256_cmp(p.bs[2], FP_HALF) > 0 : u256_cmp(p.bs[3], FP_HALF) > 0;
            b[0] |= largest ? 0xC0 : 0x80;
        }
        o.insert(o.end(), b, b + 64);
    }
    put_g1(o, p.krs[0], p.krs[1]);
    const uint32_t n = has_commit ? 1 : 0;
    o.push_back(0); o.push_back(0); o.push_back(0); o.push_back((uint8_t)n);
    if (has_commit) put_g1(o, p.commitment[0], p.commitment[1]);
    put_g1(o, p.pok[0], p.pok[1]);
    return o;
}

}  // namespace zk
