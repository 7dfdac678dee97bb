/* ORACLE (test infrastructure only).  Bucket-method MSM + fixed-base batch
 * multiplication template (instantiated for G1 and G2 by oracle.c).
 *
 * Restates the algorithm of gnark-crypto v0.19.2 ecc/bn254 multiexp.go
 * (`MultiExp`: signed-digit windows, per-window bucket accumulation in extended
 * Jacobian coordinates `processChunkG1Jacobian`, per-window running-sum reduction
 * `msmReduceChunk`, Horner fold over windows) that groth16.Prove calls from
 * gnark/libraries/prover/impl/provers.go:179,269.  The batch-affine bucket variant
 * of gnark-crypto is a speed optimisation with identical results (group arithmetic
 * is exact), so it is not restated.
 *
 * Required: PFX, CT, C_MUL, C_INV, C_ISZERO, C_SQR
 */

/* scalars: plain 256-bit integers (canonical, < r). result: ext point */
static void PFX(msm)(PFX(ext) *result, const PFX(aff) *pts, const fe_t *scalars, long n, int nthreads) {
    PFX(ext_set_inf)(result);
    if (n <= 0) return;
    int c;
    if (n < 32) c = 3; else if (n < 256) c = 5; else if (n < 2048) c = 8; else if (n < 16384) c = 10;
    else if (n < 131072) c = 12; else c = 14;
    const int nwin = (254 + c) / c + 1;        /* +1 for the signed-digit carry */
    const long nb = 1L << (c - 1);
    /* signed digit decomposition */
    int32_t *digits = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * nwin);
    for (long i = 0; i < n; i++) {
        int carry = 0;
        for (int w = 0; w < nwin; w++) {
            int bit = w * c;
            uint64_t v = 0;
            if (bit < 256) {
                int li = bit >> 6, sh = bit & 63;
                v = scalars[i].l[li] >> sh;
                if (sh + c > 64 && li < 3) v |= scalars[i].l[li + 1] << (64 - sh);
                v &= ((1ULL << c) - 1);
            }
            int d = (int)v + carry;
            if (d > (int)nb) { d -= (1 << c); carry = 1; } else carry = 0;
            digits[i * nwin + w] = d;
        }
    }
    PFX(ext) *winsum = (PFX(ext) *)malloc(sizeof(PFX(ext)) * nwin);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int w = 0; w < nwin; w++) {
        PFX(ext) *buckets = (PFX(ext) *)calloc((size_t)nb, sizeof(PFX(ext)));
        for (long i = 0; i < n; i++) {
            int d = digits[i * nwin + w];
            if (d > 0) PFX(ext_madd)(&buckets[d - 1], &pts[i]);
            else if (d < 0) PFX(ext_msub)(&buckets[-d - 1], &pts[i]);
        }
        PFX(ext) run, tot; PFX(ext_set_inf)(&run); PFX(ext_set_inf)(&tot);
        for (long b = nb - 1; b >= 0; b--) {
            PFX(ext_add)(&run, &buckets[b]);
            PFX(ext_add)(&tot, &run);
        }
        winsum[w] = tot;
        free(buckets);
    }
    PFX(ext) acc; PFX(ext_set_inf)(&acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) PFX(ext_dbl)(&acc, &acc);
        PFX(ext_add)(&acc, &winsum[w]);
    }
    *result = acc;
    free(winsum); free(digits);
}

/* batch ext -> affine (Montgomery trick on ZZZ; 1/ZZ = ZZ^2 * (1/ZZZ)^2) */
static void PFX(batch_to_aff)(PFX(aff) *out, const PFX(ext) *in, long n) {
    CT *pref = (CT *)malloc(sizeof(CT) * (size_t)(n + 1));
    CT acc; C_ONE(&acc);
    for (long i = 0; i < n; i++) {
        pref[i] = acc;
        if (!C_ISZERO(&in[i].ZZ)) C_MUL(&acc, &acc, &in[i].ZZZ);
    }
    CT inv; C_INV(&inv, &acc);
    for (long i = n - 1; i >= 0; i--) {
        if (C_ISZERO(&in[i].ZZ)) { memset(&out[i], 0, sizeof(out[i])); continue; }
        CT zi3, zi2, t;
        C_MUL(&zi3, &inv, &pref[i]);           /* 1/ZZZ_i */
        C_MUL(&inv, &inv, &in[i].ZZZ);
        C_SQR(&t, &zi3); C_SQR(&zi2, &in[i].ZZ); C_MUL(&zi2, &zi2, &t);   /* 1/ZZ_i */
        C_MUL(&out[i].x, &in[i].X, &zi2);
        C_MUL(&out[i].y, &in[i].Y, &zi3);
    }
    free(pref);
}

/* out[i] = scalars[i] * base, scalars plain canonical integers. 8-bit fixed windows. */
static void PFX(fixed_base_batch)(PFX(aff) *out, const PFX(aff) *base, const fe_t *scalars, long n, int nthreads) {
    enum { W = 8, NW = 32, TS = 255 };
    PFX(ext) *tab_e = (PFX(ext) *)malloc(sizeof(PFX(ext)) * NW * TS);
    PFX(aff) *tab = (PFX(aff) *)malloc(sizeof(PFX(aff)) * NW * TS);
    PFX(ext) b; PFX(ext_from_aff)(&b, base);
    for (int w = 0; w < NW; w++) {
        PFX(ext) acc = b;
        for (int j = 0; j < TS; j++) {
            tab_e[w * TS + j] = acc;
            PFX(ext_add)(&acc, &b);
        }
        b = acc;   /* 256 * previous base */
    }
    PFX(batch_to_aff)(tab, tab_e, NW * TS);
    free(tab_e);
    PFX(ext) *res = (PFX(ext) *)malloc(sizeof(PFX(ext)) * (size_t)(n > 0 ? n : 1));
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (long i = 0; i < n; i++) {
        PFX(ext) acc; PFX(ext_set_inf)(&acc);
        for (int w = 0; w < NW; w++) {
            unsigned d = (unsigned)((scalars[i].l[w >> 3] >> ((w & 7) * 8)) & 0xFF);
            if (d) PFX(ext_madd)(&acc, &tab[w * TS + d - 1]);
        }
        res[i] = acc;
    }
    PFX(batch_to_aff)(out, res, n);
    free(res); free(tab);
}
