// Proving-key folding at InitAlgorithm: the quotient bases pk.G1.Z in EVALUATION form.
//
// gnark's computeH (backend/groth16/bn254/prove.go, reached from provers.go:179,269) turns the per-constraint
// values a, b, c into the coefficients h_k of H = (A*B - C)/Z with 3 inverse + 3 coset-forward + 1 coset-inverse
// transforms, then Krs += sum_k h_k * Z_k.  Every map after the pointwise product, and every map applied to c, is a
// FIXED linear map, so it can be applied once to the bases instead of per proof to the scalars:
//
//   with  f_i = A(g w^i) * B(g w^i)   (coset evaluations, natural order)  and  D = g^N - 1:
//   sum_k h_k Z_k  =  sum_i f_i * Ze_i  +  sum_j c_j * Zc_j
//   Ze_i = 1/(N D) * sum_{k<N-1} w^{-ik} g^{-k} Z_k          (transpose of the inverse coset transform)
//   Zc_j = -1/(N D) * sum_{k<N-1} w^{-jk} Z_k                 (transpose of "interpolate c, drop x^(N-1)")
//
// (the x^(N-1) coefficient is dropped consistently on both sides because deg H <= N-2.)  Per proof this leaves
// 4 transforms instead of 7 and no pointwise pass over c; c enters through a second MSM whose scalars are almost
// all 0/+-1 (conditional-sum path).  The group elements of the proof are unchanged, so proofs stay byte-identical.
//
// c itself is a fixed linear map of the wire vector, c_j = sum_k O_jk w_k (the output matrix of the R1CS), so its bases fold
// one step further, into the wire-indexed bases of pk.G1.K:
//   sum_j c_j Zc_j = sum_k w_k * (sum_j O_jk Zc_j)   =>   K'_k = K_k + sum_j O_jk Zc_j       (fold_output_matrix)
// K' covers the private wires of K plus every wire that occurs in an output expression (public ones included; their
// values are in the wire vector too).  After this the prover never touches c.
//
// Both base sets are inverse DFTs of length N over G1 ("in the exponent"): radix-2 DIF butterflies with a
// variable-base scalar multiplication by the twiddle, ~N/2 log N of them, once per InitAlgorithm.
#include "kernels.h"
#include <algorithm>

namespace zk {

__device__ __forceinline__ Fr fr_pow_small(Fr base, uint64_t e) {
    Fr acc = Fr::one();
    while (e) {
        if (e & 1) acc = acc * base;
        base = base.sqr();
        e >>= 1;
    }
    return acc;
}

// out[k] = s_k * Z_k for k < N-1, infinity for k = N-1.   mode 0: s_k = -1/(N D)   mode 1: s_k = g^{-k}/(N D)
// consts: NttDomain::consts ([2]=g^-1 [3]=1/N [4]=1/D)
__global__ __launch_bounds__(64) void k_fold_scale(const G1Affine *__restrict__ z, const Fr *__restrict__ consts, size_t N, int mode, G1XYZZ *__restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    if (k == N - 1) { out[k] = G1XYZZ::inf(); return; }
    Fr s = consts[3] * consts[4];
    s = mode ? s * fr_pow_small(consts[2], k) : s.neg();
    const Fr sc = s.from_mont();
    out[k] = scalar_mul(z[k], sc.l);
}

// one DIF stage of the unscaled inverse DFT: block length len, twiddle w^{-j N/len}
__global__ __launch_bounds__(64) void k_point_butterfly(G1XYZZ *__restrict__ data, size_t N, size_t len, const Fr *__restrict__ tw_inv) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N / 2) return;
    const size_t half = len / 2, blk = t / half, j = t % half;
    const size_t i0 = blk * len + j, i1 = i0 + half;
    const G1XYZZ u = data[i0], v = data[i1];
    G1XYZZ sum = u;
    sum.add(v);
    G1XYZZ dif = u;
    dif.add(v.neg());
    data[i0] = sum;
    if (j != 0) {
        const Fr w = tw_inv[j * (N / len)].from_mont();
        dif = scalar_mul_x(dif, w.l);
    }
    data[i1] = dif;
}

// bit-reversed XYZZ -> natural-order affine (Montgomery); only the first n_out outputs are produced
__global__ __launch_bounds__(64) void k_point_unpermute(const G1XYZZ *__restrict__ in, int logn, size_t n_out, G1Affine *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const uint32_t r = logn ? (__brev((uint32_t)i) >> (32 - logn)) : 0;
    out[i] = in[r].to_affine();
}

void g1_point_idft(G1XYZZ *data, const NttDomain &dom, size_t n_out, G1Affine *out, hipStream_t s) {
    const size_t N = dom.n;
    for (size_t len = N; len >= 2; len >>= 1)
        hipLaunchKernelGGL(k_point_butterfly, dim3(ceil_div(N / 2, 64)), dim3(64), 0, s, data, N, len, dom.tw_inv.p);
    hipLaunchKernelGGL(k_point_unpermute, dim3(ceil_div(n_out, 64)), dim3(64), 0, s, data, dom.logn, n_out, out);
    ZK_HIP(hipGetLastError());
}

void fold_quotient_bases(const G1Affine *d_z, const NttDomain &dom, size_t n_constraints, G1Affine *d_zeval, G1Affine *d_zc, hipStream_t s) {
    const size_t N = dom.n;
    DevBuf<G1XYZZ> work(N);
    hipLaunchKernelGGL(k_fold_scale, dim3(ceil_div(N, 64)), dim3(64), 0, s, d_z, dom.consts.p, N, 1, work.p);
    g1_point_idft(work.p, dom, N, d_zeval, s);
    hipLaunchKernelGGL(k_fold_scale, dim3(ceil_div(N, 64)), dim3(64), 0, s, d_z, dom.consts.p, N, 0, work.p);
    g1_point_idft(work.p, dom, n_constraints, d_zc, s);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------------------ output matrix into K
// term e: coefficient coeffs[cid_e] (Montgomery) times Zc[cons_e]
__global__ __launch_bounds__(64) void k_fold_terms(const G1Affine *__restrict__ zc, const uint32_t *__restrict__ cons, const uint32_t *__restrict__ cid,
                                                   const Fr *__restrict__ coeffs, size_t n, G1XYZZ *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    Fr v = coeffs[cid[e]].from_mont();
    const Fr nv = coeffs[cid[e]].neg().from_mont();
    bool neg = false;
    for (int k = 7; k >= 0; k--) {                      // use the shorter of v and r - v
        if (v.l[k] != nv.l[k]) { neg = v.l[k] > nv.l[k]; break; }
    }
    if (neg) v = nv;
    int nbits = 0;
    for (int k = 7; k >= 0; k--) if (v.l[k]) { nbits = 32 * k + 32 - __clz(v.l[k]); break; }
    G1XYZZ t = scalar_mul(zc[cons[e]], v.l, nbits);
    out[e] = neg ? t.neg() : t;
}
// partial[c] = sum of terms [chunk_start[c], chunk_start[c+1])
__global__ __launch_bounds__(64) void k_fold_chunks(const G1XYZZ *__restrict__ terms, const uint32_t *__restrict__ chunk_start, size_t nchunks, G1XYZZ *__restrict__ partial) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    G1XYZZ acc = G1XYZZ::inf();
    for (uint32_t e = chunk_start[c]; e < chunk_start[c + 1]; e++) acc.add(terms[e]);
    partial[c] = acc;
}
// out[i] = k_in[i] (or infinity when k_index[i] < 0) + sum of the partials of output row i
__global__ __launch_bounds__(64) void k_fold_rows(const G1XYZZ *__restrict__ partial, const uint32_t *__restrict__ row_chunk, const int32_t *__restrict__ k_index,
                                                  const G1Affine *__restrict__ k_in, size_t nrows, G1Affine *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    G1XYZZ acc = k_index[i] >= 0 ? G1XYZZ::from_affine(k_in[k_index[i]]) : G1XYZZ::inf();
    for (uint32_t c = row_chunk[i]; c < row_chunk[i + 1]; c++) acc.add(partial[c]);
    out[i] = acc.to_affine();
}

void fold_output_matrix(const G1Affine *d_zc, const G1Affine *d_k, const std::vector<uint32_t> &rows_k, std::vector<OutputTerm> terms,
                        const Fr *d_coeffs, DevBuf<G1Affine> &out_points, std::vector<uint32_t> &out_rows, hipStream_t s) {
    std::stable_sort(terms.begin(), terms.end(), [](const OutputTerm &a, const OutputTerm &b) { return a.wire < b.wire; });
    // output rows = union of K's wires and the wires of the terms, ascending
    out_rows.clear();
    {
        size_t a = 0, b = 0;
        while (a < rows_k.size() || b < terms.size()) {
            const uint32_t wa = a < rows_k.size() ? rows_k[a] : 0xFFFFFFFFu, wb = b < terms.size() ? terms[b].wire : 0xFFFFFFFFu;
            const uint32_t w = wa < wb ? wa : wb;
            out_rows.push_back(w);
            while (a < rows_k.size() && rows_k[a] == w) a++;
            while (b < terms.size() && terms[b].wire == w) b++;
        }
    }
    const size_t nrows = out_rows.size(), nt = terms.size();
    static constexpr uint32_t CHUNK = 64;
    std::vector<uint32_t> cons(nt + 1), cid(nt + 1), chunk_start, row_chunk(nrows + 1);
    std::vector<int32_t> k_index(nrows + 1);
    for (size_t e = 0; e < nt; e++) { cons[e] = terms[e].cons; cid[e] = terms[e].coeff; }
    size_t a = 0, b = 0;
    for (size_t i = 0; i < nrows; i++) {
        const uint32_t w = out_rows[i];
        k_index[i] = -1;
        if (a < rows_k.size() && rows_k[a] == w) k_index[i] = (int32_t)a++;
        row_chunk[i] = (uint32_t)chunk_start.size();
        size_t e0 = b;
        while (b < nt && terms[b].wire == w) b++;
        for (size_t e = e0; e < b; e += CHUNK) chunk_start.push_back((uint32_t)e);
    }
    row_chunk[nrows] = (uint32_t)chunk_start.size();
    const size_t nchunks = chunk_start.size();
    chunk_start.push_back((uint32_t)nt);
    // a chunk must not run into the next row: chunk c ends at min(next chunk start, end of its row) -- rows are contiguous
    // in e and every row starts a new chunk, so chunk_start[c + 1] is exactly that bound.
    DevBuf<uint32_t> d_cons(nt + 1), d_cid(nt + 1), d_chunk(nchunks + 1), d_row_chunk(nrows + 1);
    DevBuf<int32_t> d_kidx(nrows + 1);
    DevBuf<G1XYZZ> d_terms(nt + 1), d_partial(nchunks + 1);
    d_cons.upload(cons.data(), nt + 1); d_cid.upload(cid.data(), nt + 1); d_chunk.upload(chunk_start.data(), nchunks + 1);
    d_row_chunk.upload(row_chunk.data(), nrows + 1); d_kidx.upload(k_index.data(), nrows + 1);
    out_points.alloc(nrows ? nrows : 1);
    ZK_HIP(hipDeviceSynchronize());
    if (nt) hipLaunchKernelGGL(k_fold_terms, dim3(ceil_div(nt, 64)), dim3(64), 0, s, d_zc, d_cons.p, d_cid.p, d_coeffs, nt, d_terms.p);
    if (nchunks) hipLaunchKernelGGL(k_fold_chunks, dim3(ceil_div(nchunks, 64)), dim3(64), 0, s, d_terms.p, d_chunk.p, nchunks, d_partial.p);
    if (nrows) hipLaunchKernelGGL(k_fold_rows, dim3(ceil_div(nrows, 64)), dim3(64), 0, s, d_partial.p, d_row_chunk.p, d_kidx.p, d_k, nrows, out_points.p);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(s));
}

}  // namespace zk
