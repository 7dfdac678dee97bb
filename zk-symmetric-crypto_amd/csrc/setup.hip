// Groth16 key generation on the GPU + gnark ProvingKey / VerifyingKey writers.
//
// Replaces `groth16.Setup(r1cs)` + `pk.WriteTo` / `vk.WriteTo` as used by gnark/keygen/keygen.go:94-132
// (gnark v0.14 backend/groth16/bn254/setup.go, not in the mount).  Needed because the reference's pk.* files
// are absent: tests and bench.py prove against keys produced here.  Single-party setup, toxic waste supplied
// by the caller (alpha, beta, gamma, delta, tau, sigma).
//   A_i = [A_i(tau)]_1, B_i = [B_i(tau)]_{1,2}, K_j = [(beta A_j + alpha B_j + C_j)(tau) / delta]_1 (private),
//   vk.K_j = [... / gamma]_1 (public + commitment wires), Pedersen basis = [... / gamma]_1 (committed wires),
//   Z_k = [tau^k (tau^N - 1) / delta]_1, Lagrange basis over <omega_N>; infinity points are dropped from A/B.
#include <string.h>
#include "../../include/zkprove.h"
#include "fixedbase.cuh"
#include "hostutil.h"
#include "r1cs_reader.h"

namespace zk {
namespace {

__device__ __forceinline__ Fr pow_u64(Fr base, uint64_t e) {
    Fr acc = Fr::one();
    while (e) {
        if (e & 1) acc = acc * base;
        base = base.sqr();
        e >>= 1;
    }
    return acc;
}
// tox: 6 canonical Fr -> Montgomery, plus derived constants
// out: [0..5] alpha beta gamma delta tau sigma (mont) [6] gamma^-1 [7] delta^-1 [8] zt = tau^N - 1 [9] zt / N [10] omega [11] zt/delta
// hdr (canonical): [0] 1/N [1] omega [2] omega^-1 [3] 5 [4] 1/5
__global__ void k_setup_consts(const U256 *tox, int logn, Fr *out, Fr *hdr) {
    if (threadIdx.x || blockIdx.x) return;
    for (int i = 0; i < 6; i++) {
        Fr v;
        for (int k = 0; k < 8; k++) v.l[k] = tox[i].l[k];
        out[i] = v.to_mont();
    }
    out[6] = out[2].inv();
    out[7] = out[3].inv();
    Fr tn = out[4];
    for (int s = 0; s < logn; s++) tn = tn.sqr();
    out[8] = tn - Fr::one();
    Fr n = Fr::from_u32(1u << logn);
    Fr ninv = n.inv();
    out[9] = out[8] * ninv;
    static const uint32_t RM1[8] = {0xf0000000u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = RM1[i];
    for (int s = 0; s < logn; s++)
        for (int i = 0; i < 8; i++) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0u);
    Fr g = Fr::from_u32(5);
    Fr w = g.pow(e, 256);
    out[10] = w;
    out[11] = out[8] * out[7];
    hdr[0] = ninv.from_mont(); hdr[1] = w.from_mont(); hdr[2] = w.inv().from_mont(); hdr[3] = g.from_mont(); hdr[4] = g.inv().from_mont();
}
// L_i(tau) = (tau^N - 1)/N * w^i / (tau - w^i)
__global__ void k_lagrange(const Fr *c, size_t N, Fr *L) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Fr wi = pow_u64(c[10], i);
    L[i] = c[9] * wi * (c[4] - wi).inv();
}
// per wire sums: which = blockIdx.y (0 A, 1 B, 2 C); CSR by wire: off[which*(nw+1) + w], terms (cid, cons)
__global__ void k_wire_sums(const uint32_t *off, const uint32_t *terms, const Fr *coeffs, const Fr *L, size_t nw, Fr *out) {
    size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const int which = blockIdx.y;
    const uint32_t lo = off[which * (nw + 1) + w], hi = off[which * (nw + 1) + w + 1];
    Fr acc = Fr::zero();
    for (uint32_t t = lo; t < hi; t++) acc = acc + coeffs[terms[2 * t]] * L[terms[2 * t + 1]];
    out[which * nw + w] = acc;
}
// canonical scalars: sA, sB (wire order), sKg = (beta A + alpha B + C)/gamma, sKd = .../delta, sKgs = sKg * sigma
__global__ void k_wire_scalars(const Fr *abc, const Fr *c, size_t nw, Fr *sA, Fr *sB, Fr *sKg, Fr *sKd, Fr *sKgs) {
    size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const Fr a = abc[w], b = abc[nw + w], cc = abc[2 * nw + w];
    const Fr t = a * c[1] + b * c[0] + cc;
    const Fr kg = t * c[6];
    sA[w] = a.from_mont(); sB[w] = b.from_mont();
    sKg[w] = kg.from_mont(); sKd[w] = (t * c[7]).from_mont(); sKgs[w] = (kg * c[5]).from_mont();
}
__global__ void k_z_scalars(const Fr *c, size_t n, Fr *sZ) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    sZ[k] = (c[11] * pow_u64(c[4], k)).from_mont();
}
__global__ void k_misc_scalars(const Fr *c, Fr *out) {     // alpha beta delta | beta delta gamma -sigma
    if (threadIdx.x || blockIdx.x) return;
    out[0] = c[0].from_mont(); out[1] = c[1].from_mont(); out[2] = c[3].from_mont();
    out[3] = c[1].from_mont(); out[4] = c[3].from_mont(); out[5] = c[2].from_mont(); out[6] = c[5].neg().from_mont();
}
template <class F>
__global__ __launch_bounds__(64) void k_fixed_mul_batch(const Affine<F> *__restrict__ tab, const Fr *__restrict__ scalars, size_t n, Affine<F> *__restrict__ out);
template <>
__global__ __launch_bounds__(64) void k_fixed_mul_batch<Fp>(const G1Affine *__restrict__ tab, const Fr *__restrict__ scalars, size_t n, G1Affine *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a = fixed_mul(tab, scalars[i].l).to_affine();
    out[i] = G1Affine{a.x.from_mont(), a.y.from_mont()};
}
template <>
__global__ __launch_bounds__(64) void k_fixed_mul_batch<Fp2>(const G2Affine *__restrict__ tab, const Fr *__restrict__ scalars, size_t n, G2Affine *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G2Affine a = fixed_mul(tab, scalars[i].l).to_affine();
    out[i] = G2Affine{a.x.from_mont(), a.y.from_mont()};
}
__global__ void k_generators(G1Affine *g1, G2Affine *g2) {
    if (threadIdx.x || blockIdx.x) return;
    g1->x = Fp::from_u32(1); g1->y = Fp::from_u32(2);
    // standard BN254 G2 generator (gnark-crypto bn254.go), canonical limbs -> Montgomery
#include "g2_generator.inc"
    Fp t;
    for (int k = 0; k < 8; k++) t.l[k] = G2_GEN_X0[k]; g2->x.a0 = t.to_mont();
    for (int k = 0; k < 8; k++) t.l[k] = G2_GEN_X1[k]; g2->x.a1 = t.to_mont();
    for (int k = 0; k < 8; k++) t.l[k] = G2_GEN_Y0[k]; g2->y.a0 = t.to_mont();
    for (int k = 0; k < 8; k++) t.l[k] = G2_GEN_Y1[k]; g2->y.a1 = t.to_mont();
}

// ---- byte writers (gnark-crypto marshal.go conventions; see oracle/gnark_format.py for the decoded layout)
struct Bytes {
    std::vector<uint8_t> v;
    void u32be(uint32_t x) { for (int i = 3; i >= 0; i--) v.push_back((uint8_t)(x >> (8 * i))); }
    void u64be(uint64_t x) { for (int i = 7; i >= 0; i--) v.push_back((uint8_t)(x >> (8 * i))); }
    void fe(const uint32_t *l) { for (int i = 7; i >= 0; i--) for (int k = 3; k >= 0; k--) v.push_back((uint8_t)(l[i] >> (8 * k))); }
};
static bool lex_largest(const uint32_t *y) {
    U256 t;
    memcpy(t.l, y, 32);
    return u256_cmp(t, FP_HALF) > 0;
}
static bool is_zero8(const uint32_t *l) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= l[i]; return o == 0; }
static void put_g1(Bytes &b, const G1Affine &p, bool raw) {
    if (raw) { b.fe(p.x.l); b.fe(p.y.l); return; }
    size_t at = b.v.size();
    if (is_zero8(p.x.l) && is_zero8(p.y.l)) { b.v.insert(b.v.end(), 32, 0); b.v[at] = 0x40; return; }
    b.fe(p.x.l);
    b.v[at] |= lex_largest(p.y.l) ? 0xC0 : 0x80;
}
static void put_g2(Bytes &b, const G2Affine &p, bool raw) {
    if (raw) { b.fe(p.x.a1.l); b.fe(p.x.a0.l); b.fe(p.y.a1.l); b.fe(p.y.a0.l); return; }
    size_t at = b.v.size();
    if (is_zero8(p.x.a0.l) && is_zero8(p.x.a1.l) && is_zero8(p.y.a0.l) && is_zero8(p.y.a1.l)) { b.v.insert(b.v.end(), 64, 0); b.v[at] = 0x40; return; }
    b.fe(p.x.a1.l); b.fe(p.x.a0.l);
    const bool largest = is_zero8(p.y.a1.l) ? lex_largest(p.y.a0.l) : lex_largest(p.y.a1.l);
    b.v[at] |= largest ? 0xC0 : 0x80;
}
static void put_bools(Bytes &b, const std::vector<uint8_t> &f) {     // u32 length, bits LSB-first  [gnark-mem, unpinned]
    b.u32be((uint32_t)f.size());
    size_t at = b.v.size();
    b.v.insert(b.v.end(), (f.size() + 7) / 8, 0);
    for (size_t i = 0; i < f.size(); i++) if (f[i]) b.v[at + i / 8] |= (uint8_t)(1u << (i % 8));
}

}  // namespace

void groth16_setup(const uint8_t *r1cs_bytes, size_t r1cs_len, const uint8_t *toxic192, bool raw, std::vector<uint8_t> &pk_out, std::vector<uint8_t> &vk_out) {
    R1cs cs;
    r1cs_parse(r1cs_bytes, r1cs_len, cs);
    if (cs.commitments.size() > 1) throw ZkError("setup: more than one commitment is not supported");
    const bool has_commit = !cs.commitments.empty();
    const int logn = cs.log_domain();
    const size_t N = (size_t)1 << logn, nw = cs.nb_wires(), npub = cs.nb_public;
    hipStream_t st = 0;
    // CSR by wire of the three constraint matrices
    std::vector<uint32_t> cnt(3 * (nw + 1), 0);
    auto walk = [&](auto &&fn) {
        for (size_t i = 0; i < cs.nb_instructions(); i++) {
            if (cs.bp_kind[cs.blueprint_id[i]] != BP_R1C) continue;
            const uint32_t *cd = cs.calldata.data() + cs.start_calldata[i];
            const uint32_t n[3] = {cd[1], cd[2], cd[3]};
            const uint32_t *p = cd + 4;
            for (int which = 0; which < 3; which++)
                for (uint32_t k = 0; k < n[which]; k++, p += 2) {
                    if (p[1] >= nw) throw ZkError("setup: wire id out of range");
                    fn(which, p[0], p[1], cs.constraint_offset[i]);
                }
        }
    };
    walk([&](int which, uint32_t, uint32_t w, uint32_t) { cnt[which * (nw + 1) + w + 1]++; });
    std::vector<uint32_t> off(3 * (nw + 1), 0);
    uint32_t run = 0;
    for (int which = 0; which < 3; which++) {
        for (size_t w = 0; w <= nw; w++) {
            if (w > 0) run += cnt[which * (nw + 1) + w];
            off[which * (nw + 1) + w] = run;
        }
    }
    std::vector<uint32_t> terms(2 * (size_t)run), cur(off);
    walk([&](int which, uint32_t cid, uint32_t w, uint32_t cons) {
        uint32_t &c = cur[which * (nw + 1) + w];
        terms[2 * (size_t)c] = cid; terms[2 * (size_t)c + 1] = cons;
        c++;
    });
    DevBuf<U256> d_tox(6);
    ZK_HIP(hipMemcpy(d_tox.p, toxic192, 192, hipMemcpyHostToDevice));
    DevBuf<Fr> d_c(16), d_hdr(8), d_L(N), d_coeffs(cs.coeffs.size()), d_abc(3 * nw), d_sA(nw), d_sB(nw), d_sKg(nw), d_sKd(nw), d_sKgs(nw), d_sZ(N), d_misc(8);
    DevBuf<uint32_t> d_off(off.size()), d_terms(terms.size() ? terms.size() : 1);
    d_off.upload(off.data(), off.size(), st);
    d_terms.upload(terms.data(), terms.size(), st);
    ZK_HIP(hipMemcpyAsync(d_coeffs.p, cs.coeffs.data(), cs.coeffs.size() * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_setup_consts, dim3(1), dim3(64), 0, st, d_tox.p, logn, d_c.p, d_hdr.p);
    hipLaunchKernelGGL(k_lagrange, dim3(ceil_div(N, 128)), dim3(128), 0, st, d_c.p, N, d_L.p);
    hipLaunchKernelGGL(k_wire_sums, dim3(ceil_div(nw, 128), 3), dim3(128), 0, st, d_off.p, d_terms.p, d_coeffs.p, d_L.p, nw, d_abc.p);
    hipLaunchKernelGGL(k_wire_scalars, dim3(ceil_div(nw, 128)), dim3(128), 0, st, d_abc.p, d_c.p, nw, d_sA.p, d_sB.p, d_sKg.p, d_sKd.p, d_sKgs.p);
    hipLaunchKernelGGL(k_z_scalars, dim3(ceil_div(N - 1, 128)), dim3(128), 0, st, d_c.p, N - 1, d_sZ.p);
    hipLaunchKernelGGL(k_misc_scalars, dim3(1), dim3(64), 0, st, d_c.p, d_misc.p);
    ZK_HIP(hipGetLastError());
    std::vector<Fr> sA(nw), sB(nw), sKg(nw), sKd(nw), sKgs(nw), sZ(N - 1), misc(7), hdr(5);
    d_sA.download(sA.data(), nw, st); d_sB.download(sB.data(), nw, st); d_sKg.download(sKg.data(), nw, st);
    d_sKd.download(sKd.data(), nw, st); d_sKgs.download(sKgs.data(), nw, st); d_sZ.download(sZ.data(), N - 1, st);
    d_misc.download(misc.data(), 7, st); d_hdr.download(hdr.data(), 5, st);
    ZK_HIP(hipStreamSynchronize(st));
    // classify wires and build the ordered scalar lists
    std::vector<uint8_t> committed(nw, 0), infA(nw, 0), infB(nw, 0);
    uint32_t commit_wire = 0xffffffffu;
    if (has_commit) {
        for (uint32_t w : cs.commitments[0].private_committed) { if (w >= nw) throw ZkError("setup: committed wire out of range"); committed[w] = 1; }
        commit_wire = cs.commitments[0].commitment_wire;
    }
    std::vector<Fr> g1s, g2s;
    g1s.push_back(misc[0]); g1s.push_back(misc[1]); g1s.push_back(misc[2]);
    size_t nA = 0, nB = 0, nK = 0, nVK = 0, nPed = 0;
    for (size_t w = 0; w < nw; w++) { if (is_zero8(sA[w].l)) infA[w] = 1; else { g1s.push_back(sA[w]); nA++; } }
    for (size_t w = 0; w < nw; w++) { if (is_zero8(sB[w].l)) infB[w] = 1; else { g1s.push_back(sB[w]); nB++; } }
    for (size_t k = 0; k + 1 < N; k++) g1s.push_back(sZ[k]);
    auto is_vk = [&](size_t w) { return w < npub || w == commit_wire; };
    for (size_t w = 0; w < nw; w++) if (!is_vk(w) && !committed[w]) { g1s.push_back(sKd[w]); nK++; }
    for (size_t w = 0; w < nw; w++) if (is_vk(w) && !committed[w]) { g1s.push_back(sKg[w]); nVK++; }
    for (size_t w = 0; w < nw; w++) if (committed[w]) { g1s.push_back(sKg[w]); nPed++; }
    for (size_t w = 0; w < nw; w++) if (committed[w]) g1s.push_back(sKgs[w]);
    g2s.push_back(misc[3]); g2s.push_back(misc[4]); g2s.push_back(misc[5]); g2s.push_back(misc[6]);
    for (size_t w = 0; w < nw; w++) if (!infB[w]) g2s.push_back(sB[w]);
    // fixed-base multiplications
    DevBuf<G1Affine> d_g1(1), d_t1(32 * 255), d_o1(g1s.size());
    DevBuf<G2Affine> d_g2(1), d_t2(32 * 255), d_o2(g2s.size());
    DevBuf<Fr> d_s1(g1s.size()), d_s2(g2s.size());
    d_s1.upload(g1s.data(), g1s.size(), st); d_s2.upload(g2s.data(), g2s.size(), st);
    hipLaunchKernelGGL(k_generators, dim3(1), dim3(64), 0, st, d_g1.p, d_g2.p);
    hipLaunchKernelGGL(k_fixed_table<Fp>, dim3(ceil_div(32 * 255, 64)), dim3(64), 0, st, d_g1.p, d_t1.p);
    hipLaunchKernelGGL(k_fixed_table<Fp2>, dim3(ceil_div(32 * 255, 64)), dim3(64), 0, st, d_g2.p, d_t2.p);
    hipLaunchKernelGGL(k_fixed_mul_batch<Fp>, dim3(ceil_div(g1s.size(), 64)), dim3(64), 0, st, d_t1.p, d_s1.p, g1s.size(), d_o1.p);
    hipLaunchKernelGGL(k_fixed_mul_batch<Fp2>, dim3(ceil_div(g2s.size(), 64)), dim3(64), 0, st, d_t2.p, d_s2.p, g2s.size(), d_o2.p);
    ZK_HIP(hipGetLastError());
    std::vector<G1Affine> o1(g1s.size());
    std::vector<G2Affine> o2(g2s.size()), gen2(1);
    d_o1.download(o1.data(), o1.size(), st); d_o2.download(o2.data(), o2.size(), st);
    ZK_HIP(hipStreamSynchronize(st));
    {   // pedersen G = G2 generator (canonical)
        DevBuf<G2Affine> tmp(1);
        DevBuf<Fr> one(1);
        Fr o = Fr::zero(); o.l[0] = 1;
        one.upload(&o, 1, st);
        hipLaunchKernelGGL(k_fixed_mul_batch<Fp2>, dim3(1), dim3(64), 0, st, d_t2.p, one.p, (size_t)1, tmp.p);
        tmp.download(gen2.data(), 1, st);
        ZK_HIP(hipStreamSynchronize(st));
    }
    // ---- proving key (ProvingKey.WriteTo / WriteRawTo)
    Bytes pk;
    pk.u64be(N);
    for (int i = 0; i < 5; i++) pk.fe(hdr[i].l);
    pk.v.push_back(0);
    size_t o = 0;
    put_g1(pk, o1[0], raw); put_g1(pk, o1[1], raw); put_g1(pk, o1[2], raw);
    o = 3;
    auto slice1 = [&](Bytes &b, size_t n) { b.u32be((uint32_t)n); for (size_t i = 0; i < n; i++) put_g1(b, o1[o + i], raw); o += n; };
    slice1(pk, nA); slice1(pk, nB); slice1(pk, N - 1); slice1(pk, nK);
    const size_t o_vk = o;
    o += nVK;
    const size_t o_ped = o;
    put_g2(pk, o2[0], raw); put_g2(pk, o2[1], raw);
    pk.u32be((uint32_t)nB);
    for (size_t i = 0; i < nB; i++) put_g2(pk, o2[4 + i], raw);
    pk.u64be(nw); pk.u64be(nw - nA); pk.u64be(nw - nB);
    put_bools(pk, infA); put_bools(pk, infB);
    pk.u32be(has_commit ? 1 : 0);
    if (has_commit) {
        o = o_ped;
        slice1(pk, nPed); slice1(pk, nPed);
    }
    pk_out.swap(pk.v);
    // ---- verifying key: alpha1 beta1 beta2 gamma2 delta1 delta2, K, PublicAndCommitmentCommitted, commitment keys (compressed)
    Bytes vk;
    put_g1(vk, o1[0], false); put_g1(vk, o1[1], false); put_g2(vk, o2[0], false); put_g2(vk, o2[2], false); put_g1(vk, o1[2], false); put_g2(vk, o2[1], false);
    vk.u32be((uint32_t)nVK);
    for (size_t i = 0; i < nVK; i++) put_g1(vk, o1[o_vk + i], false);
    vk.u32be(has_commit ? 1 : 0);
    if (has_commit) vk.u32be(0);
    vk.u32be(has_commit ? 1 : 0);
    if (has_commit) { put_g2(vk, gen2[0], false); put_g2(vk, o2[3], false); }
    vk_out.swap(vk.v);
}

}  // namespace zk

extern "C" int ZkpSetup(GoSlice r1cs, const uint8_t *toxic192, int raw, void **pk_out, long long *pk_len, void **vk_out, long long *vk_len) {
    try {
        zk::bind_default_device();
        std::vector<uint8_t> pk, vk;
        zk::groth16_setup((const uint8_t *)r1cs.data, (size_t)r1cs.len, toxic192, raw != 0, pk, vk);
        *pk_out = malloc(pk.size()); memcpy(*pk_out, pk.data(), pk.size()); *pk_len = (long long)pk.size();
        *vk_out = malloc(vk.size()); memcpy(*vk_out, vk.data(), vk.size()); *vk_len = (long long)vk.size();
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "zkprove setup: %s\n", e.what());
        return -1;
    }
}
