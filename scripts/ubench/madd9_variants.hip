// Variants of the 9-limb mixed addition (curve9.cuh) timed against the production version:
//   V0 production XYZZ9::madd
//   V1 + squarings (45 instead of 81 products for P^2 and R^2)
//   V2 independent products run in lockstep (two or three accumulator chains interleaved: no dependent back-to-back mads)
//   V3 V2 + squarings
// Every variant must reproduce V0's limbs exactly (strict limbs are unique per integer).
#include "../../zk-symmetric-crypto_amd/csrc/curve9.cuh"
#include <stdio.h>
#include <string.h>
#include <vector>
using namespace zk;
using F9 = Fp9;

struct Term {
    const F9 *a, *b;    // b == nullptr: square of a
};
// K reductions in lockstep, reduction j = sum of np[j] <= 2 terms
template <int K>
__device__ __forceinline__ void mont_lockstep(const Term (&tm)[K][2], const int (&np)[K], F9 (&r)[K]) {
    int64_t acc[K];
    int32_t m[K][9];
    F9 a2[K][2];
#pragma unroll
    for (int j = 0; j < K; j++) {
        acc[j] = 0;
#pragma unroll
        for (int t = 0; t < 2; t++)
            if (t < np[j] && tm[j][t].b == nullptr) a2[j][t] = F9::add(*tm[j][t].a, *tm[j][t].a);
    }
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int lo = k > 8 ? k - 8 : 0, hi = k < 8 ? k : 8;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            const int jj = k - i;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < K; j++) {
                    if (t >= np[j]) continue;
                    const Term &T = tm[j][t];
                    if (T.b != nullptr) F9::mac(acc[j], T.a->l[i], T.b->l[jj]);
                    else if (i < jj) F9::mac(acc[j], a2[j][t].l[i], T.a->l[jj]);
                    else if (i == jj) F9::mac(acc[j], T.a->l[i], T.a->l[i]);
                }
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++)
#pragma unroll
                for (int j = 0; j < K; j++) F9::mac_const(acc[j], m[j][i], Fp9Params::M[k - i]);
#pragma unroll
            for (int j = 0; j < K; j++) m[j][k] = (int32_t)(((uint32_t)acc[j] * Fp9Params::PINV) & F9::MASK);
#pragma unroll
            for (int j = 0; j < K; j++) { F9::mac_const(acc[j], m[j][k], Fp9Params::M[0]); acc[j] >>= 29; }
        } else {
#pragma unroll
            for (int i = k - 8; i < 9; i++)
#pragma unroll
                for (int j = 0; j < K; j++) F9::mac_const(acc[j], m[j][i], Fp9Params::M[k - i]);
#pragma unroll
            for (int j = 0; j < K; j++) { r[j].l[k - 9] = (int32_t)((uint32_t)acc[j] & F9::MASK); acc[j] >>= 29; }
        }
    }
#pragma unroll
    for (int j = 0; j < K; j++) r[j].l[8] = (int32_t)acc[j];
}

template <int V>
__device__ __forceinline__ void madd_v(G1X9 &A, const F9 &x2, const F9 &y2) {
    if (V == 0) { A.madd(x2, y2); return; }
    if (A.is_inf()) { A.X = x2; A.Y = F9::norm(y2); A.ZZ = F9::one(); A.ZZZ = F9::one(); return; }
    constexpr bool LOCK = V >= 2, SQ = (V == 1 || V == 3);
    F9 U2, S2, PP, RR, PPP, Q;
    if (LOCK) {
        F9 r[2];
        const Term t[2][2] = {{{&x2, &A.ZZ}, {nullptr, nullptr}}, {{&y2, &A.ZZZ}, {nullptr, nullptr}}};
        const int np[2] = {1, 1};
        mont_lockstep<2>(t, np, r);
        U2 = r[0]; S2 = r[1];
    } else { U2 = F9::mul(x2, A.ZZ); S2 = F9::mul(y2, A.ZZZ); }
    const F9 Pv = F9::sub(U2, A.X), R = F9::sub(S2, A.Y);
    {
        F9 r[2];
        const Term t[2][2] = {{{&Pv, SQ ? nullptr : &Pv}, {nullptr, nullptr}}, {{&R, SQ ? nullptr : &R}, {nullptr, nullptr}}};
        const int np[2] = {1, 1};
        if (LOCK) { mont_lockstep<2>(t, np, r); PP = r[0]; RR = r[1]; }
        else {
            F9 r1[1];
            const Term t0[1][2] = {{{&Pv, SQ ? nullptr : &Pv}, {nullptr, nullptr}}};
            const Term t1[1][2] = {{{&R, SQ ? nullptr : &R}, {nullptr, nullptr}}};
            const int n1[1] = {1};
            mont_lockstep<1>(t0, n1, r1); PP = r1[0];
            mont_lockstep<1>(t1, n1, r1); RR = r1[0];
        }
    }
    if (PP.is_zero_mod()) { A = G1X9::madd_special(x2, F9::norm(y2), R); return; }
    F9 ZZ3;
    if (LOCK) {
        F9 r[3];
        const Term t[3][2] = {{{&Pv, &PP}, {nullptr, nullptr}}, {{&A.X, &PP}, {nullptr, nullptr}}, {{&A.ZZ, &PP}, {nullptr, nullptr}}};
        const int np[3] = {1, 1, 1};
        mont_lockstep<3>(t, np, r);
        PPP = r[0]; Q = r[1]; ZZ3 = r[2];
    } else { PPP = F9::mul(Pv, PP); Q = F9::mul(A.X, PP); ZZ3 = F9::mul(A.ZZ, PP); }
    F9 X3;
#pragma unroll
    for (int i = 0; i < 9; i++) X3.l[i] = RR.l[i] - PPP.l[i] - 2 * Q.l[i];
    X3 = F9::norm(X3);
    const F9 Tq = F9::sub(Q, X3), nY = F9::neg(A.Y);
    if (LOCK) {
        F9 r[2];
        const Term t[2][2] = {{{&R, &Tq}, {&nY, &PPP}}, {{&A.ZZZ, &PPP}, {nullptr, nullptr}}};
        const int np[2] = {2, 1};
        mont_lockstep<2>(t, np, r);
        A.Y = r[0]; A.ZZZ = r[1];
    } else { A.Y = F9::mulsub(R, Tq, A.Y, PPP); A.ZZZ = F9::mul(A.ZZZ, PPP); }
    A.X = X3;
    A.ZZ = ZZ3;
}

template <int V, int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_madd(G1X9 *out, const G1Affine *pts9, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    G1X9 a = G1X9::inf();
    for (int k = 0; k < iters; k++) {
        const G1Affine p = pts9[(i * 7 + k) & 1023];
        madd_v<V>(a, F9::from_words(p.x), F9::cneg(F9::from_words(p.y), (k & 4) ? -1 : 0));
    }
    out[i] = a;
}
__global__ void k_make_points(G1Affine *pts9, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine g{Fp::one(), Fp::one() + Fp::one()};
    uint32_t k[8] = {(uint32_t)(i * 2654435761u + 12345u), (uint32_t)i + 77u, 0, 0, 0, 0, 0, 0};
    G1Affine p = scalar_mul(g, k, 64).to_affine();
    Fp c32;
    for (int j = 0; j < 8; j++) c32.l[j] = Fp9Params::SCALE32[j];
    pts9[i] = G1Affine{p.x * c32, p.y * c32};
}

template <int V, int W>
static void run(const G1Affine *pts9, G1X9 *o, std::vector<G1X9> &ref, size_t TM, const char *name) {
    hipLaunchKernelGGL((k_madd<V, W>), dim3(TM / 64), dim3(64), 0, 0, o, pts9, 37);
    std::vector<G1X9> r(TM);
    hipMemcpy(r.data(), o, TM * sizeof(G1X9), hipMemcpyDeviceToHost);
    size_t bad = 0;
    if (ref.empty()) ref = r;
    else for (size_t i = 0; i < TM; i++) bad += memcmp(&r[i], &ref[i], sizeof(G1X9)) != 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms, best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL((k_madd<V, W>), dim3(TM / 64), dim3(64), 0, 0, o, pts9, 400); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-34s waves<=%d  %8.3f ms  %6.2f G madd/s  mismatches vs V0: %zu\n", name, W, best, (double)TM * 400 / best / 1e6, bad);
}
int main() {
    G1Affine *pts9; hipMalloc(&pts9, 1024 * 64);
    hipLaunchKernelGGL(k_make_points, dim3(4), dim3(256), 0, 0, pts9, 1024);
    size_t TM = 256 * 256 * 8;
    G1X9 *o; hipMalloc(&o, TM * sizeof(G1X9));
    std::vector<G1X9> ref;
    run<0, 4>(pts9, o, ref, TM, "V0 production");
    run<1, 4>(pts9, o, ref, TM, "V1 squarings");
    run<2, 4>(pts9, o, ref, TM, "V2 lockstep");
    run<3, 4>(pts9, o, ref, TM, "V3 lockstep + squarings");
    run<0, 3>(pts9, o, ref, TM, "V0 production");
    run<2, 3>(pts9, o, ref, TM, "V2 lockstep");
    run<3, 3>(pts9, o, ref, TM, "V3 lockstep + squarings");
    run<0, 5>(pts9, o, ref, TM, "V0 production");
    run<1, 5>(pts9, o, ref, TM, "V1 squarings");
    run<3, 2>(pts9, o, ref, TM, "V3 lockstep + squarings");
    return 0;
}
