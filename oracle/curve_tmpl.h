/* ORACLE (test infrastructure only).  Curve-arithmetic template, instantiated twice
 * by curve.h: G1 over Fp (y^2 = x^3 + 3) and G2 over Fp2 (y^2 = x^3 + 3/(9+u)).
 * Restates gnark-crypto v0.19.2 ecc/bn254 g1.go / g2.go "extended Jacobian" (XYZZ)
 * bucket arithmetic (g1JacExtended: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) -- formulas
 * are the EFD madd-2008-s / add-2008-s / dbl-2008-s-1 used there.
 *
 * Required macros: PFX(name), CT (coord type), C_ADD,C_SUB,C_MUL,C_SQR,C_DBL,C_NEG,
 * C_ISZERO,C_EQ,C_INV,C_ONE(ptr)
 */
typedef struct { CT x, y; } PFX(aff);             /* infinity = (0,0) as in gnark-crypto */
typedef struct { CT X, Y, ZZ, ZZZ; } PFX(ext);    /* infinity: ZZ == 0 */

static inline int PFX(aff_is_inf)(const PFX(aff) *p) { return C_ISZERO(&p->x) && C_ISZERO(&p->y); }
static inline int PFX(ext_is_inf)(const PFX(ext) *p) { return C_ISZERO(&p->ZZ); }
static inline void PFX(ext_set_inf)(PFX(ext) *p) { memset(p, 0, sizeof(*p)); }
static inline void PFX(ext_from_aff)(PFX(ext) *r, const PFX(aff) *p) {
    if (PFX(aff_is_inf)(p)) { PFX(ext_set_inf)(r); return; }
    r->X = p->x; r->Y = p->y; C_ONE(&r->ZZ); C_ONE(&r->ZZZ);
}
static inline void PFX(ext_neg)(PFX(ext) *r, const PFX(ext) *p) { *r = *p; C_NEG(&r->Y, &p->Y); }
static inline void PFX(aff_neg)(PFX(aff) *r, const PFX(aff) *p) { *r = *p; C_NEG(&r->y, &p->y); }

static inline void PFX(ext_dbl_aff)(PFX(ext) *r, const PFX(aff) *p) {
    CT U, V, W, S, M, t;
    C_DBL(&U, &p->y); C_SQR(&V, &U); C_MUL(&W, &U, &V); C_MUL(&S, &p->x, &V);
    C_SQR(&t, &p->x); C_DBL(&M, &t); C_ADD(&M, &M, &t);
    C_SQR(&r->X, &M); C_SUB(&r->X, &r->X, &S); C_SUB(&r->X, &r->X, &S);
    C_SUB(&t, &S, &r->X); C_MUL(&t, &M, &t); C_MUL(&U, &W, &p->y); C_SUB(&r->Y, &t, &U);
    r->ZZ = V; r->ZZZ = W;
}
static inline void PFX(ext_dbl)(PFX(ext) *r, const PFX(ext) *p) {
    if (PFX(ext_is_inf)(p)) { *r = *p; return; }
    CT U, V, W, S, M, t, X3;
    C_DBL(&U, &p->Y); C_SQR(&V, &U); C_MUL(&W, &U, &V); C_MUL(&S, &p->X, &V);
    C_SQR(&t, &p->X); C_DBL(&M, &t); C_ADD(&M, &M, &t);
    C_SQR(&X3, &M); C_SUB(&X3, &X3, &S); C_SUB(&X3, &X3, &S);
    C_SUB(&t, &S, &X3); C_MUL(&t, &M, &t); C_MUL(&U, &W, &p->Y);
    C_SUB(&r->Y, &t, &U); r->X = X3;
    C_MUL(&r->ZZ, &V, &p->ZZ); C_MUL(&r->ZZZ, &W, &p->ZZZ);
}
/* r += q (affine) */
static inline void PFX(ext_madd)(PFX(ext) *r, const PFX(aff) *q) {
    if (PFX(aff_is_inf)(q)) return;
    if (PFX(ext_is_inf)(r)) { r->X = q->x; r->Y = q->y; C_ONE(&r->ZZ); C_ONE(&r->ZZZ); return; }
    CT U2, S2, P, R, PP, PPP, Q, t, X3;
    C_MUL(&U2, &q->x, &r->ZZ); C_MUL(&S2, &q->y, &r->ZZZ);
    C_SUB(&P, &U2, &r->X); C_SUB(&R, &S2, &r->Y);
    if (C_ISZERO(&P)) {
        if (C_ISZERO(&R)) { PFX(ext_dbl_aff)(r, q); return; }
        PFX(ext_set_inf)(r); return;
    }
    C_SQR(&PP, &P); C_MUL(&PPP, &P, &PP); C_MUL(&Q, &r->X, &PP);
    C_SQR(&X3, &R); C_SUB(&X3, &X3, &PPP); C_SUB(&X3, &X3, &Q); C_SUB(&X3, &X3, &Q);
    C_SUB(&t, &Q, &X3); C_MUL(&t, &R, &t); C_MUL(&U2, &r->Y, &PPP); C_SUB(&r->Y, &t, &U2);
    r->X = X3;
    C_MUL(&r->ZZ, &r->ZZ, &PP); C_MUL(&r->ZZZ, &r->ZZZ, &PPP);
}
static inline void PFX(ext_msub)(PFX(ext) *r, const PFX(aff) *q) {
    PFX(aff) n; PFX(aff_neg)(&n, q); PFX(ext_madd)(r, &n);
}
/* r += q (ext) */
static inline void PFX(ext_add)(PFX(ext) *r, const PFX(ext) *q) {
    if (PFX(ext_is_inf)(q)) return;
    if (PFX(ext_is_inf)(r)) { *r = *q; return; }
    CT U1, U2, S1, S2, P, R, PP, PPP, Q, t, X3;
    C_MUL(&U1, &r->X, &q->ZZ); C_MUL(&U2, &q->X, &r->ZZ);
    C_MUL(&S1, &r->Y, &q->ZZZ); C_MUL(&S2, &q->Y, &r->ZZZ);
    C_SUB(&P, &U2, &U1); C_SUB(&R, &S2, &S1);
    if (C_ISZERO(&P)) {
        if (C_ISZERO(&R)) { PFX(ext_dbl)(r, r); return; }
        PFX(ext_set_inf)(r); return;
    }
    C_SQR(&PP, &P); C_MUL(&PPP, &P, &PP); C_MUL(&Q, &U1, &PP);
    C_SQR(&X3, &R); C_SUB(&X3, &X3, &PPP); C_SUB(&X3, &X3, &Q); C_SUB(&X3, &X3, &Q);
    C_SUB(&t, &Q, &X3); C_MUL(&t, &R, &t); C_MUL(&S1, &S1, &PPP); C_SUB(&r->Y, &t, &S1);
    r->X = X3;
    C_MUL(&t, &r->ZZ, &q->ZZ); C_MUL(&r->ZZ, &t, &PP);
    C_MUL(&t, &r->ZZZ, &q->ZZZ); C_MUL(&r->ZZZ, &t, &PPP);
}
static inline void PFX(ext_to_aff)(PFX(aff) *r, const PFX(ext) *p) {
    if (PFX(ext_is_inf)(p)) { memset(r, 0, sizeof(*r)); return; }
    CT zi, zi2, zi3;           /* ZZZ/ZZ = Z ; 1/Z^2 = 1/ZZ ; 1/Z^3 = 1/ZZZ */
    C_INV(&zi2, &p->ZZ); C_INV(&zi3, &p->ZZZ); (void)zi;
    C_MUL(&r->x, &p->X, &zi2); C_MUL(&r->y, &p->Y, &zi3);
}
/* r = k * p ; k = 256-bit plain integer (4 LE limbs) -- double-and-add, MSB first */
static inline void PFX(ext_scalar_mul)(PFX(ext) *r, const PFX(aff) *p, const fe_t *k) {
    PFX(ext) acc; PFX(ext_set_inf)(&acc);
    for (int i = 255; i >= 0; i--) {
        PFX(ext_dbl)(&acc, &acc);
        if ((k->l[i >> 6] >> (i & 63)) & 1) PFX(ext_madd)(&acc, p);
    }
    *r = acc;
}
