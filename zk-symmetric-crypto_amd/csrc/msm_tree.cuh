// Bucket accumulation with AFFINE additions that share one field inversion ("batch-affine"), as a pairwise reduction tree.
//
// Replaces gnark-crypto v0.19.2 ecc/bn254 processChunkG1BatchAffine / batchAddG1Affine (multiexp_affine.go; the 350- and
// 400-point batches in the reference's default.pgo): affine + affine costs 5 multiplications, 1 squaring
// and a share of one inversion (Montgomery's trick) instead of the 8 + 2 of an extended-Jacobian mixed addition.
//
// MI355X formulation.  Input is the bucket-sorted entry list of an instance (k_msm_partition / k_msm_binsort / k_msm_sort).  One
// thread owns one piece of it (whole buckets, ~2048 entries, exactly as k_msm_accumulate) and reduces every bucket of the piece by
// halving rounds: round r adds the items of a bucket two by two, so a bucket of n entries is a single affine point after
// ceil(log2 n) rounds.  All additions of one round are independent, which is what the shared inversion needs:
//   forward  : walk the piece's pairs, multiply the running product c by the pair's denominator (x1 - x0), park the product before
//              it and the pair's descriptor in a per-wavefront HBM scratch ([slot][lane] layout: every access is one 2 KiB / 1 KiB row);
//   invert   : ONE inversion of c per thread and chunk of <= CH pairs (Field::inv_gcd, ~50 multiplications' worth);
//   backward : slots in reverse: 1 / den = inv * c_before, inv *= den, lambda, x3, y3, store the sum.
// The layout of every round is a pure function of the sorted layout (bucket_start) -- nothing is scanned or compacted.  Relative
// to its piece (first bucket k0), bucket k = k0 + k' holds its count_r items of round r at
//   start_r(k') = (start_{r-1}(k') + k') >> 1,  count_r = ceil(count_{r-1} / 2),  start_0(k') = bucket_start[k] - bucket_start[k0],
// which keeps start_r(k'+1) - start_r(k') >= count_r(k').  Odd rounds live in buffer A, even rounds in buffer B, each piece inside
// an interval of its own (tree_piece_base): threads run their rounds at their own pace, so a thread's round-3 items must never
// land on another thread's unread round-1 items.  A bucket's final point goes to final[k].  No synchronisation at all.
// Exceptional pairs (an operand at infinity, P + P, P - P, an odd item without partner) take denominator 1 and are resolved by
// selects in the backward pass, so parity with the oracle holds for adversarial inputs too (equal scalars, infinity bases).
#pragma once
#include "curve.cuh"

namespace zk {

static constexpr uint32_t TREE_ABSENT = 0xffffffffu;
static constexpr uint32_t TREE_FINAL = 0x80000000u;        // dst flag: the sum is the bucket's final point -> final[k]
static constexpr int TREE_CH = 256;                        // pairs per thread between two inversions
template <class F>
constexpr size_t tree_slot_bytes() { return sizeof(F) + 16; }       // prefix product + descriptor, per pair and lane

ZK_HD uint32_t tree_start(uint32_t s0, uint32_t k, int r) {
    for (int i = 0; i < r; i++) s0 = (s0 + k) >> 1;
    return s0;
}
ZK_HD uint32_t tree_count(uint32_t n0, int r) { return r >= 32 ? (n0 ? 1u : 0u) : (uint32_t)(((uint64_t)n0 + ((1ull << r) - 1)) >> r); }

template <class F>
ZK_HD F field_inv_shared(const F &x);
template <>
ZK_HD Fp field_inv_shared<Fp>(const Fp &x) { return x.inv_gcd(); }
template <>
ZK_HD Fp2 field_inv_shared<Fp2>(const Fp2 &x) {
    const Fp n = (x.a0.sqr() + x.a1.sqr()).inv_gcd();
    return Fp2{x.a0 * n, (x.a1 * n).neg()};
}

// classification of one pair; the SAME function decides the denominator in both passes
//   generic : distinct x, both finite -> den = x1 - x0
//   dbl     : P + P (y != 0)          -> den = 2 y
//   else    : den = 1 (an operand at infinity or absent, P - P)
template <class F>
struct PairKind { bool inf0, inf1, xeq, dbl, generic; };
template <class F>
ZK_HD PairKind<F> pair_kind(const Affine<F> &p0, const Affine<F> &p1) {
    PairKind<F> k;
    k.inf0 = p0.is_inf(); k.inf1 = p1.is_inf();
    k.xeq = p0.x == p1.x;
    k.generic = !k.inf0 && !k.inf1 && !k.xeq;
    k.dbl = !k.inf0 && !k.inf1 && k.xeq && p0.y == p1.y && !p0.y.is_zero();
    return k;
}

// Interval of piece p = buckets [k0, k1) of an instance inside buffer A (shift 1) or B (shift 2): it starts at
// (bucket_start[k0] >> shift) + k0 + 2 p, which leaves every piece at least max(entries >> shift, buckets) + buckets / ... slots --
// enough for start_r(k1 - k0) of every round r of that parity (start_r never exceeds max(start_{r-1}, buckets)).
ZK_HD size_t tree_piece_base(uint32_t bs_k0, uint32_t k0, uint32_t piece, int shift) { return (size_t)(bs_k0 >> shift) + k0 + 2 * (size_t)piece; }
// capacity of buffer A / B for an instance of at most max_entries entries, nb buckets, nlanes pieces
inline size_t tree_capacity(size_t max_entries, size_t nb, size_t nlanes, int shift) { return (max_entries >> shift) + 2 * nb + 2 * nlanes + 4; }

// sorted-layout walker of one piece for round r: emits (src0, src1, dst) of every pair of every bucket with >= 2 items.
// src / dst are absolute indices: into the entry list (round 1 sources) or into the instance's buffer A / B.
struct TreeWalker {
    const uint32_t *bs;
    uint32_t k, k_first, k_end, e_first, e0, e1;      // current bucket, its original [e0, e1)
    uint32_t t, m, n_prev, s_in, s_out;
    uint32_t base_in, base_out;                       // this piece's interval in the buffers of layout r-1 / r
    int r;
    bool more;                       // some bucket of the piece still has >= 2 items after this round
    ZK_HD void init(const uint32_t *bs_, uint32_t k0, uint32_t k1, uint32_t piece, int round) {
        bs = bs_; k = k0; k_first = k0; k_end = k1; r = round; t = 0; m = 0; more = false;
        if (k < k_end) {
            e0 = bs[k]; e1 = bs[k + 1]; e_first = e0;
            const uint32_t bA = (uint32_t)tree_piece_base(e_first, k0, piece, 1), bB = (uint32_t)tree_piece_base(e_first, k0, piece, 2);
            base_in = r == 1 ? e_first : (((r - 1) & 1) ? bA : bB);      // layout 0 = the entry list itself
            base_out = (r & 1) ? bA : bB;
            enter();
        }
    }
    ZK_HD void enter() {         // position on the first bucket >= k that has pairs
        while (k < k_end) {
            const uint32_t n0 = e1 - e0;
            n_prev = tree_count(n0, r - 1);
            if (n_prev >= 2) {
                m = (n_prev + 1) >> 1;
                s_in = base_in + tree_start(e0 - e_first, k - k_first, r - 1);
                s_out = base_out + tree_start(e0 - e_first, k - k_first, r);
                t = 0;
                more = more || m >= 2;
                return;
            }
            k++;
            if (k < k_end) { e0 = e1; e1 = bs[k + 1]; }
        }
        m = 0;
    }
    // next pair of this lane; false when the piece is exhausted
    ZK_HD bool next(uint32_t &src0, uint32_t &src1, uint32_t &dst) {
        if (k >= k_end) return false;
        src0 = s_in + 2 * t;
        src1 = (2 * t + 1 < n_prev) ? src0 + 1 : TREE_ABSENT;
        dst = (m == 1) ? (TREE_FINAL | k) : (s_out + t);
        t++;
        if (t >= m) {
            k++;
            if (k < k_end) { e0 = e1; e1 = bs[k + 1]; enter(); }
        }
        return true;
    }
};

template <class F>
ZK_HD Affine<F> tree_load_entry(const Affine<F> *__restrict__ table, uint32_t v) {
    Affine<F> p = table[v & 0x7fffffffu];
    if (v >> 31) p.y = p.y.neg();
    return p;
}

// piece `lane` of `nlanes` of an instance: buckets [k0, k1), cut at equal entry counts rounded to bucket boundaries
ZK_HD void tree_piece_bounds(const uint32_t *bs, uint32_t nb, uint32_t lane, uint32_t nlanes, uint32_t &k0, uint32_t &k1) {
    const uint32_t total = bs[nb];
    auto first_bucket = [&](uint32_t target) -> uint32_t {      // first bucket k with bs[k] >= target
        uint32_t lo = 0, hi = nb;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (bs[mid] >= target) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    k0 = first_bucket((uint32_t)(((uint64_t)total * lane + nlanes - 1) / nlanes));
    k1 = lane + 1 == nlanes ? nb : first_bucket((uint32_t)(((uint64_t)total * (lane + 1) + nlanes - 1) / nlanes));
    if (k0 > k1) k0 = k1;
}

// wave vote of the device build; the host build (tests/host emulation of one lane at a time) has nothing to vote with
ZK_HD bool tree_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(p) != 0;
#else
    return p;
#endif
}

// All halving rounds of the piece [k0, k1) of one instance.  pref / desc: this lane's columns of the wavefront's scratch rows
// (element i of a column sits `stride` elements after element i - 1: 64 on the device, 1 in the host emulation).
//
// Memory latency: a gathered base point takes 1-3 us to arrive while a multiplication takes ~0.5 us, so neither pass may wait for one
// slot's data at a time.  The forward pass handles TREE_G slots per step -- all their index words first, then all their x coordinates,
// then the arithmetic; the backward pass keeps the next slot's descriptor, prefix product and both points in flight while it
// computes the current one.
static constexpr int TREE_G = 4;

template <class F, bool FIRST>
struct TreeFwd {
    // forward pass over at most TREE_CH slots: returns the wave-uniform number of rows used; cnt = this lane's slots
    static ZK_HD int run(const Affine<F> *__restrict__ table, const uint32_t *ent, const Affine<F> *in, TreeWalker &w, F *pref, uint4 *desc, size_t stride,
                         F &cprod, int &cnt, uint32_t dbg) {
        int i = 0;
        cnt = 0;
        cprod = F::one();
        for (; i < TREE_CH; i += TREE_G) {
            uint32_t s0[TREE_G], s1[TREE_G], dst[TREE_G], v0[TREE_G], v1[TREE_G];
            bool have[TREE_G];
#pragma unroll
            for (int g = 0; g < TREE_G; g++) { s0[g] = 0; s1[g] = TREE_ABSENT; dst[g] = 0; have[g] = w.next(s0[g], s1[g], dst[g]); }
            if (!tree_any(have[0])) break;                        // a lane's slots are contiguous: nobody has one
            // index words (round 1: the sorted entries; later rounds: positions in the previous layout)
#pragma unroll
            for (int g = 0; g < TREE_G; g++) {
                v0[g] = s0[g]; v1[g] = s1[g];
                if (FIRST && have[g]) { v0[g] = ent[s0[g]]; if (s1[g] != TREE_ABSENT) v1[g] = ent[s1[g]]; }
            }
            const Affine<F> *q0[TREE_G], *q1[TREE_G];
            F x0[TREE_G], x1[TREE_G];
#pragma unroll
            for (int g = 0; g < TREE_G; g++) {
                q0[g] = FIRST ? table + (v0[g] & 0x7fffffffu) : in + v0[g];
                q1[g] = FIRST ? table + (v1[g] & 0x7fffffffu) : in + v1[g];
                x0[g] = F::zero(); x1[g] = F::zero();
                if (have[g]) { x0[g] = q0[g]->x; if (v1[g] != TREE_ABSENT) x1[g] = q1[g]->x; }
            }
            int rows = 0;
#pragma unroll
            for (int g = 0; g < TREE_G; g++) {
                rows += tree_any(have[g]) ? 1 : 0;
                if (have[g]) {
                    const bool has1 = v1[g] != TREE_ABSENT;
                    F den;
                    if (!(dbg & 16) && has1 && x0[g] != x1[g] && !x0[g].is_zero() && !x1[g].is_zero()) {
                        den = x1[g] - x0[g];                      // generic: the sign of y never matters here
                    } else {                                      // rare: look at the whole points
                        Affine<F> p0 = *q0[g], p1 = has1 ? *q1[g] : Affine<F>::inf();
                        if (FIRST) { if (v0[g] >> 31) p0.y = p0.y.neg(); if (has1 && (v1[g] >> 31)) p1.y = p1.y.neg(); }
                        const PairKind<F> kd = pair_kind(p0, p1);
                        den = kd.generic ? p1.x - p0.x : (kd.dbl ? p0.y.dbl() : F::one());
                    }
                    pref[(size_t)(i + g) * stride] = cprod;
                    desc[(size_t)(i + g) * stride] = uint4{v0[g], v1[g], dst[g], 0u};
                    cprod = cprod * den;
                    cnt++;
                }
            }
            if (rows < TREE_G) { i += rows; break; }
        }
        return i;
    }
};

template <class F, bool FIRST>
struct TreeBwd {
    struct Slot { uint4 d; F cb; Affine<F> p0, p1; };
    static ZK_HD uint4 load_desc(const uint4 *desc, size_t stride, int j, int cnt) { return (j >= 0 && j < cnt) ? desc[(size_t)j * stride] : uint4{0u, TREE_ABSENT, 0u, 0u}; }
    static ZK_HD void load_rest(Slot &s, const Affine<F> *__restrict__ table, const Affine<F> *in, const F *pref, size_t stride, int j, int cnt) {
        s.p1 = Affine<F>::inf();
        if (j >= 0 && j < cnt) {
            s.cb = pref[(size_t)j * stride];
            if (FIRST) {
                s.p0 = table[s.d.x & 0x7fffffffu];
                if (s.d.y != TREE_ABSENT) s.p1 = table[s.d.y & 0x7fffffffu];
            } else {
                s.p0 = in[s.d.x];
                if (s.d.y != TREE_ABSENT) s.p1 = in[s.d.y];
            }
        }
    }
    // backward pass over rows used-1 .. 0
    static ZK_HD void run(const Affine<F> *__restrict__ table, const Affine<F> *in, Affine<F> *out, Affine<F> *out_final, const F *pref, const uint4 *desc,
                          size_t stride, F inv, int used, int cnt) {
        Slot cur, nxt;
        cur.d = load_desc(desc, stride, used - 1, cnt);
        load_rest(cur, table, in, pref, stride, used - 1, cnt);
        nxt.d = load_desc(desc, stride, used - 2, cnt);
        for (int j = used - 1; j >= 0; j--) {
            // next slot: its points (its descriptor arrived during the previous step); the descriptor after that
            load_rest(nxt, table, in, pref, stride, j - 1, cnt);
            const uint4 d2 = load_desc(desc, stride, j - 2, cnt);
            if (j < cnt) {
                Affine<F> p0 = cur.p0, p1 = cur.p1;
                if (FIRST) { if (cur.d.x >> 31) p0.y = p0.y.neg(); if (cur.d.y != TREE_ABSENT && (cur.d.y >> 31)) p1.y = p1.y.neg(); }
                const PairKind<F> kd = pair_kind(p0, p1);
                Affine<F> res;
                if (kd.generic || kd.dbl) {
                    const F den = kd.generic ? p1.x - p0.x : p0.y.dbl();
                    const F dinv = inv * cur.cb;
                    inv = inv * den;
                    F num;
                    if (kd.generic) num = p1.y - p0.y;
                    else { const F xx = p0.x.sqr(); num = xx.dbl() + xx; }
                    const F lam = num * dinv;
                    res.x = lam.sqr() - p0.x - p1.x;
                    res.y = lam * (p0.x - res.x) - p0.y;
                } else {                                     // den = 1: inv unchanged
                    res = kd.inf0 ? p1 : (kd.inf1 ? p0 : Affine<F>::inf());
                }
                if (cur.d.z & TREE_FINAL) out_final[cur.d.z & ~TREE_FINAL] = res;
                else out[cur.d.z] = res;
            }
            cur = nxt;
            nxt.d = d2;
        }
    }
};

template <class F>
ZK_HD void tree_piece(const Affine<F> *__restrict__ table, const uint32_t *bs, const uint32_t *ent, uint32_t k0, uint32_t k1, uint32_t piece, Affine<F> *A, Affine<F> *Bf,
                      Affine<F> *out_final, F *pref, uint4 *desc, size_t stride, uint32_t dbg = 0) {
    for (int r = 1; r < 40; r++) {
        const Affine<F> *in = (r & 1) ? Bf : A;          // layout r-1 (r >= 2)
        Affine<F> *out = (r & 1) ? A : Bf;               // layout r
        TreeWalker w;
        w.init(bs, k0, k1, piece, r);
        while (true) {
            F cprod;
            int cnt = 0;
            const int used = r == 1 ? TreeFwd<F, true>::run(table, ent, in, w, pref, desc, stride, cprod, cnt, dbg)
                                    : TreeFwd<F, false>::run(table, ent, in, w, pref, desc, stride, cprod, cnt, dbg);
            if (used == 0) break;                                // wave-uniform: the longest lane
            if (dbg & 32) { if (used < TREE_CH) break; continue; }      // timing experiment: forward pass only (results are garbage)
            // ---- one inversion per thread and chunk
            const F inv = field_inv_shared<F>(cprod);
            if (r == 1) TreeBwd<F, true>::run(table, in, out, out_final, pref, desc, stride, inv, used, cnt);
            else TreeBwd<F, false>::run(table, in, out, out_final, pref, desc, stride, inv, used, cnt);
            if (used < TREE_CH) break;                           // every lane ran out of pairs inside this chunk
        }
        if (!tree_any(w.more)) break;
    }
}

// One thread per piece (same cut as k_msm_accumulate).  scratch: gridDim.x * TREE_CH * 64 * tree_slot_bytes<F>() bytes.
template <class F, bool DENSE>
__device__ __forceinline__ void tree_body(const Affine<F> *__restrict__ table, const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ entries,
                                          size_t max_entries, int c, size_t B, uint32_t nlanes, Affine<F> *__restrict__ bufA, size_t capA,
                                          Affine<F> *__restrict__ bufB, size_t capB, Affine<F> *__restrict__ fin, uint8_t *__restrict__ scratch, uint32_t dbg) {
    const uint32_t nb = 1u << (c - 1);
    const size_t tg = (dbg & 8) ? (size_t)blockIdx.x : (size_t)blockIdx.x * 64 + threadIdx.x;      // debug: one live lane per wavefront
    const bool live = tg < (size_t)nlanes * B && (!(dbg & 8) || threadIdx.x == 0);
    const size_t b = live ? tg / nlanes : 0;
    const uint32_t lane = live ? (uint32_t)(tg % nlanes) : 0;
    const uint32_t *bs = bucket_start + b * (nb + 1);
    const uint32_t *ent = entries + b * max_entries;
    uint32_t k0 = 0, k1 = 0;
    if (live) tree_piece_bounds(bs, nb, lane, nlanes, k0, k1);
    // per-wavefront scratch rows: [slot][lane]
    uint8_t *wave_base = scratch + (size_t)blockIdx.x * TREE_CH * 64 * tree_slot_bytes<F>();
    F *pref = (F *)wave_base + threadIdx.x;                                             // [TREE_CH][64]
    uint4 *desc = (uint4 *)(wave_base + (size_t)TREE_CH * 64 * sizeof(F)) + threadIdx.x;  // [TREE_CH][64]
    tree_piece<F>(table, bs, ent, k0, k1, lane, bufA + b * capA, bufB + b * capB, fin + b * nb, pref, desc, 64, dbg);
}

template <class F, bool DENSE>
__global__ __launch_bounds__(64) void k_msm_tree(const Affine<F> *__restrict__ table, const uint32_t *__restrict__ bucket_start, const uint32_t *__restrict__ entries,
                                                 size_t max_entries, int c, size_t B, uint32_t nlanes, Affine<F> *__restrict__ bufA, size_t capA,
                                                 Affine<F> *__restrict__ bufB, size_t capB, Affine<F> *__restrict__ fin, uint8_t *__restrict__ scratch, uint32_t dbg) {
    tree_body<F, DENSE>(table, bucket_start, entries, max_entries, c, B, nlanes, bufA, capA, bufB, capB, fin, scratch, dbg);
}

// thread (b, seg): running-sum reduction of buckets [seg*L, seg*L + L) of instance b, bucket sums as produced by k_msm_tree:
// n0 = 0 empty, n0 = 1 the entry itself (table), n0 >= 2 final[k]
template <class F>
__global__ __launch_bounds__(64) void k_msm_bucket_reduce_tree(const Affine<F> *__restrict__ fin, const Affine<F> *__restrict__ table,
                                                               const uint32_t *__restrict__ entries, size_t max_entries,
                                                               const uint32_t *__restrict__ bucket_start, int c, size_t B, XYZZ<F> *__restrict__ part,
                                                               size_t part_stride, size_t part_off, uint32_t L) {
    const uint32_t nb = 1u << (c - 1);
    const uint32_t nseg = nb / L;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nseg * B) return;
    const size_t b = t / nseg;
    const uint32_t seg = (uint32_t)(t % nseg);
    const uint32_t *bs = bucket_start + b * (nb + 1);
    const Affine<F> *in = fin + b * nb;
    const uint32_t *ent = entries + b * max_entries;
    XYZZ<F> run = XYZZ<F>::inf(), tot = XYZZ<F>::inf();
    const uint32_t s0 = seg * L;
    uint32_t hi = bs[s0 + L];
    for (int k = (int)L - 1; k >= 0; k--) {
        const uint32_t lo = bs[s0 + k];
        if (hi > lo) run.madd(hi - lo == 1 ? tree_load_entry(table, ent[lo]) : in[s0 + k]);
        hi = lo;
        tot.add(run);
    }
    // contribution = tot + s0 * run   (bucket index i carries weight i+1)
    if (s0 != 0 && !run.is_inf()) {
        XYZZ<F> m = XYZZ<F>::inf();
        for (int bit = 31 - __clz(s0); bit >= 0; bit--) {
            m = m.dbl();
            if ((s0 >> bit) & 1) m.add(run);
        }
        tot.add(m);
    }
    part[b * part_stride + part_off + seg] = tot;
}

}  // namespace zk
