/* ORACLE (test infrastructure only -- never linked into the product library).
 *
 * Plain-C restatement of the BN254 base/scalar field arithmetic that the reference
 * prover gets from github.com/consensys/gnark-crypto v0.19.2 (gnark/go.mod:7;
 * packages ecc/bn254/fp and ecc/bn254/fr -- not vendored in /root/reference).
 * Same representation as gnark-crypto: 4 x 64-bit little-endian limbs in Montgomery
 * form with R = 2^256 (this is also the on-disk form of the R1CS coefficient table,
 * SURVEY.md App. A).  Call sites in the reference: impl/provers.go:175-188,265-279.
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe_t;

typedef struct {
    fe_t m;        /* modulus */
    uint64_t inv;  /* -m^-1 mod 2^64 */
    fe_t one;      /* R mod m */
    fe_t r2;       /* R^2 mod m */
} field_ctx;

extern field_ctx FP, FR;
void oracle_fields_init(void);

static inline int fe_is_zero(const fe_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe_t *a, const fe_t *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int fe_geq(const fe_t *a, const fe_t *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static inline uint64_t fe_add_raw(fe_t *z, const fe_t *a, const fe_t *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; z->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t fe_sub_raw(fe_t *z, const fe_t *a, const fe_t *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - br;
        z->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline void f_add(const field_ctx *F, fe_t *z, const fe_t *a, const fe_t *b) {
    fe_t t; uint64_t c = fe_add_raw(&t, a, b);
    if (c || fe_geq(&t, &F->m)) fe_sub_raw(&t, &t, &F->m);
    *z = t;
}
static inline void f_sub(const field_ctx *F, fe_t *z, const fe_t *a, const fe_t *b) {
    fe_t t; if (fe_sub_raw(&t, a, b)) fe_add_raw(&t, &t, &F->m);
    *z = t;
}
static inline void f_neg(const field_ctx *F, fe_t *z, const fe_t *a) {
    if (fe_is_zero(a)) { *z = *a; return; }
    fe_t t; fe_sub_raw(&t, &F->m, a); *z = t;
}
static inline void f_dbl(const field_ctx *F, fe_t *z, const fe_t *a) { f_add(F, z, a, a); }

/* CIOS Montgomery multiplication */
static inline void f_mul(const field_ctx *F, fe_t *z, const fe_t *x, const fe_t *y) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    const uint64_t *m = F->m.l;
    for (int i = 0; i < 4; i++) {
        u128 c;
        uint64_t xi = x->l[i];
        c = (u128)xi * y->l[0] + t0; t0 = (uint64_t)c; c >>= 64;
        c += (u128)xi * y->l[1] + t1; t1 = (uint64_t)c; c >>= 64;
        c += (u128)xi * y->l[2] + t2; t2 = (uint64_t)c; c >>= 64;
        c += (u128)xi * y->l[3] + t3; t3 = (uint64_t)c; c >>= 64;
        c += t4; t4 = (uint64_t)c; uint64_t t5 = (uint64_t)(c >> 64);
        uint64_t q = t0 * F->inv;
        c = (u128)q * m[0] + t0; c >>= 64;
        c += (u128)q * m[1] + t1; t0 = (uint64_t)c; c >>= 64;
        c += (u128)q * m[2] + t2; t1 = (uint64_t)c; c >>= 64;
        c += (u128)q * m[3] + t3; t2 = (uint64_t)c; c >>= 64;
        c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);
    }
    fe_t t = {{t0, t1, t2, t3}};
    if (t4 || fe_geq(&t, &F->m)) fe_sub_raw(&t, &t, &F->m);
    *z = t;
}
static inline void f_sqr(const field_ctx *F, fe_t *z, const fe_t *x) { f_mul(F, z, x, x); }
static inline void f_to_mont(const field_ctx *F, fe_t *z, const fe_t *a) { f_mul(F, z, a, &F->r2); }
static inline void f_from_mont(const field_ctx *F, fe_t *z, const fe_t *a) {
    fe_t one = {{1, 0, 0, 0}}; f_mul(F, z, a, &one);
}
void f_pow(const field_ctx *F, fe_t *z, const fe_t *a, const fe_t *e /* plain integer */);
void f_inv(const field_ctx *F, fe_t *z, const fe_t *a);   /* 0 -> 0 */
void f_batch_inv(const field_ctx *F, fe_t *v, long n, fe_t *scratch); /* zeros stay zero */
void f_set_u64(const field_ctx *F, fe_t *z, uint64_t v);

/* ---- Fp2 = Fp[u]/(u^2+1) ---- */
typedef struct { fe_t a0, a1; } fp2_t;
static inline void fp2_add(fp2_t *z, const fp2_t *a, const fp2_t *b) { f_add(&FP, &z->a0, &a->a0, &b->a0); f_add(&FP, &z->a1, &a->a1, &b->a1); }
static inline void fp2_sub(fp2_t *z, const fp2_t *a, const fp2_t *b) { f_sub(&FP, &z->a0, &a->a0, &b->a0); f_sub(&FP, &z->a1, &a->a1, &b->a1); }
static inline void fp2_neg(fp2_t *z, const fp2_t *a) { f_neg(&FP, &z->a0, &a->a0); f_neg(&FP, &z->a1, &a->a1); }
static inline void fp2_dbl(fp2_t *z, const fp2_t *a) { fp2_add(z, a, a); }
static inline void fp2_mul(fp2_t *z, const fp2_t *a, const fp2_t *b) {
    fe_t t0, t1, s0, s1, m;
    f_mul(&FP, &t0, &a->a0, &b->a0);
    f_mul(&FP, &t1, &a->a1, &b->a1);
    f_add(&FP, &s0, &a->a0, &a->a1);
    f_add(&FP, &s1, &b->a0, &b->a1);
    f_mul(&FP, &m, &s0, &s1);
    f_sub(&FP, &m, &m, &t0);
    f_sub(&FP, &z->a1, &m, &t1);
    f_sub(&FP, &z->a0, &t0, &t1);
}
static inline void fp2_sqr(fp2_t *z, const fp2_t *a) {
    fe_t s, d, m;
    f_add(&FP, &s, &a->a0, &a->a1);
    f_sub(&FP, &d, &a->a0, &a->a1);
    f_mul(&FP, &m, &a->a0, &a->a1);
    f_mul(&FP, &z->a0, &s, &d);
    f_dbl(&FP, &z->a1, &m);
}
static inline int fp2_is_zero(const fp2_t *a) { return fe_is_zero(&a->a0) && fe_is_zero(&a->a1); }
static inline int fp2_eq(const fp2_t *a, const fp2_t *b) { return fe_eq(&a->a0, &b->a0) && fe_eq(&a->a1, &b->a1); }
void fp2_inv(fp2_t *z, const fp2_t *a);
#endif
