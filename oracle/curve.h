/* ORACLE (test infrastructure only).  BN254 G1 / G2 instantiations of curve_tmpl.h. */
#ifndef ORACLE_CURVE_H
#define ORACLE_CURVE_H
#include "field.h"

/* ---- G1 over Fp ---- */
#define PFX(n) g1_##n
#define CT fe_t
#define C_ADD(z, a, b) f_add(&FP, z, a, b)
#define C_SUB(z, a, b) f_sub(&FP, z, a, b)
#define C_MUL(z, a, b) f_mul(&FP, z, a, b)
#define C_SQR(z, a) f_sqr(&FP, z, a)
#define C_DBL(z, a) f_dbl(&FP, z, a)
#define C_NEG(z, a) f_neg(&FP, z, a)
#define C_ISZERO(a) fe_is_zero(a)
#define C_EQ(a, b) fe_eq(a, b)
#define C_INV(z, a) f_inv(&FP, z, a)
#define C_ONE(z) (*(z) = FP.one)
#include "curve_tmpl.h"
#undef PFX
#undef CT
#undef C_ADD
#undef C_SUB
#undef C_MUL
#undef C_SQR
#undef C_DBL
#undef C_NEG
#undef C_ISZERO
#undef C_EQ
#undef C_INV
#undef C_ONE

/* ---- G2 over Fp2 ---- */
static inline void fp2_set_one(fp2_t *z) { z->a0 = FP.one; memset(&z->a1, 0, sizeof(fe_t)); }
#define PFX(n) g2_##n
#define CT fp2_t
#define C_ADD(z, a, b) fp2_add(z, a, b)
#define C_SUB(z, a, b) fp2_sub(z, a, b)
#define C_MUL(z, a, b) fp2_mul(z, a, b)
#define C_SQR(z, a) fp2_sqr(z, a)
#define C_DBL(z, a) fp2_dbl(z, a)
#define C_NEG(z, a) fp2_neg(z, a)
#define C_ISZERO(a) fp2_is_zero(a)
#define C_EQ(a, b) fp2_eq(a, b)
#define C_INV(z, a) fp2_inv(z, a)
#define C_ONE(z) fp2_set_one(z)
#include "curve_tmpl.h"
#undef PFX
#undef CT
#undef C_ADD
#undef C_SUB
#undef C_MUL
#undef C_SQR
#undef C_DBL
#undef C_NEG
#undef C_ISZERO
#undef C_EQ
#undef C_INV
#undef C_ONE
#endif
