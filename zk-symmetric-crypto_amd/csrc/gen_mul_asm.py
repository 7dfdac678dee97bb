#!/usr/bin/env python3
"""Generates field_mul_asm.inc: the gfx950 Montgomery multiplication as ONE inline-asm block.

Finely integrated product scanning (FIPS): 16 columns, each column accumulated in a 96-bit window
{lo, mid | hi} with  v_mad_u64_u32 (carry-out -> vcc) + v_addc_co_u32 (third word).  One block so that
hipcc cannot interleave its own s_nop / v_mov between the carry pairs (measured +44 % over the compiled
CIOS loop on MI355X, scripts/ubench/mul_variants.hip).

Register plan (explicit, declared as clobbers): accumulator pairs A = v[16:17], B = v[18:19]
(column k uses pair P = A if k even else B as {lo, mid}; the carry word is the HIGH half of the other pair,
so the only data movement per column is one v_mov of `mid` into the other pair's low half);
s[16:23] = modulus limbs, s24 = -M^-1 mod 2^32.
Operands: %0-%7 result / m_k temporaries (early clobber), %8-%15 a, %16-%23 b.
"""
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def gen(mod):
    inv = (-pow(mod, -1, 1 << 32)) % (1 << 32)
    L = []
    for i in range(8):
        L.append("s_mov_b32 s%d, 0x%08x" % (16 + i, (mod >> (32 * i)) & 0xFFFFFFFF))
    L.append("s_mov_b32 s24, 0x%08x" % inv)
    pair = [("v[16:17]", "v16", "v17"), ("v[18:19]", "v18", "v19")]
    a = lambda i: "%%%d" % (8 + i)
    b = lambda i: "%%%d" % (16 + i)
    t = lambda i: "%%%d" % i
    for k in range(16):
        Pp, P0, P1 = pair[k & 1]
        Qp, Q0, Q1 = pair[(k + 1) & 1]
        first = [True]

        def mac(x, y):
            if k == 0 and first[0]:
                L.append("v_mad_u64_u32 %s, vcc, %s, %s, 0" % (Pp, x, y))      # cannot carry
                first[0] = False
                first.append("pending")       # carry word still uninitialised
                return
            L.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (Pp, x, y, Pp))
            if first[0] or (len(first) > 1 and first[-1] == "pending"):
                L.append("v_addc_co_u32_e64 %s, vcc, 0, 0, vcc" % Q1)            # initialise the carry word
                first[0] = False
                if len(first) > 1:
                    first[-1] = "done"
            else:
                L.append("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (Q1, Q1))

        if k < 8:
            for i in range(k + 1):
                mac(a(i), b(k - i))
            for i in range(k):
                mac(t(i), "s%d" % (16 + k - i))
            L.append("v_mul_lo_u32 %s, %s, s24" % (t(k), P0))
            mac(t(k), "s16")
        else:
            for i in range(k - 7, 8):
                mac(a(i), b(k - i))
            for i in range(k - 7, 8):
                mac(t(i), "s%d" % (16 + k - i))
            if k == 15:
                L.append("v_mov_b32 %s, %s" % (t(7), P0))
                break
            L.append("v_mov_b32 %s, %s" % (t(k - 8), P0))
        L.append("v_mov_b32 %s, %s" % (Q0, P1))
    return L


def limbs(v):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def gen_reduce(mod, r, d):
    """r (8 register names, value < 2 mod) -> r mod `mod`; d = 8 scratch registers.
    d = r + (2^256 - mod): the carry out of the top limb says r >= mod."""
    neg = limbs((1 << 256) - mod)
    # a literal (or an SGPR) next to the carry in VCC would be two constant-bus reads: the constant goes through the scratch VGPRs
    L = ["v_mov_b32_e32 %s, 0x%08x" % (d[i], neg[i]) for i in range(8)]
    L.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (d[0], d[0], r[0]))
    for i in range(1, 8):
        L.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (d[i], d[i], r[i]))
    for i in range(8):
        L.append("v_cndmask_b32_e32 %s, %s, %s, vcc" % (r[i], r[i], d[i]))
    return L


def gen_add(mod):
    """carry chain only: %0-%7 r = a + b (no reduction), %8-%15 a, %16-%23 b; followed by ZK_RED_ASM"""
    r = ["%%%d" % i for i in range(8)]
    a = ["%%%d" % (8 + i) for i in range(8)]
    b = ["%%%d" % (16 + i) for i in range(8)]
    L = ["v_add_co_u32_e32 %s, vcc, %s, %s" % (r[0], a[0], b[0])]
    for i in range(1, 8):
        L.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (r[i], a[i], b[i]))
    return L


def gen_sub(mod):
    """%0-%7 r = a - b mod 2^256, %8 = all ones iff it borrowed, %9-%16 a, %17-%24 b; followed by ZK_SUBFIX_ASM"""
    r = ["%%%d" % i for i in range(8)]
    m = "%8"
    a = ["%%%d" % (9 + i) for i in range(8)]
    b = ["%%%d" % (17 + i) for i in range(8)]
    L = ["v_sub_co_u32_e32 %s, vcc, %s, %s" % (r[0], a[0], b[0])]
    for i in range(1, 8):
        L.append("v_subb_co_u32_e32 %s, vcc, %s, %s, vcc" % (r[i], a[i], b[i]))
    L.append("v_subb_co_u32_e64 %s, vcc, 0, 0, vcc" % m)
    return L


def gen_subfix(mod):
    """%0-%7 r (in/out) += mod & mask, %8-%15 scratch (out), %16 mask (in; operand numbering: the tied inputs of r come after it)"""
    r = ["%%%d" % i for i in range(8)]
    d = ["%%%d" % (8 + i) for i in range(8)]
    m = "%16"
    ml = limbs(mod)
    L = []
    for i in range(8):
        L.append("v_and_b32_e32 %s, 0x%08x, %s" % (d[i], ml[i], m))
    L.append("v_add_co_u32_e32 %s, vcc, %s, %s" % (r[0], r[0], d[0]))
    for i in range(1, 8):
        L.append("v_addc_co_u32_e32 %s, vcc, %s, %s, vcc" % (r[i], r[i], d[i]))
    return L


def limbs9(v):
    return [(v >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]


def gen_field9():
    """field9_consts.inc: constants of the 9 x 29-bit signed-limb representation (field9.cuh), Montgomery radix 2^261"""
    out = ["// GENERATED by gen_mul_asm.py -- do not edit.  Constants of field9.cuh (9 x 29-bit limbs, Montgomery radix 2^261)."]
    fmt = lambda v: "{ " + ", ".join("0x%08x" % x for x in limbs9(v)) + " }"
    for name, mod in (("FP", P), ("FR", R)):
        out.append("#define ZK_F9_%s_M %s" % (name, fmt(mod)))
        out.append("#define ZK_F9_%s_PINV 0x%08xu                  /* -M^-1 mod 2^29 */" % (name, (-pow(mod, -1, 1 << 29)) % (1 << 29)))
        out.append("#define ZK_F9_%s_CIN %s    /* 2^266 mod M: 2^256-Montgomery words -> 2^261-Montgomery limbs */" % (name, fmt(pow(2, 266, mod))))
        out.append("#define ZK_F9_%s_COUT %s   /* 2^256 mod M: and back */" % (name, fmt(pow(2, 256, mod))))
        out.append("#define ZK_F9_%s_ONE %s    /* 2^261 mod M */" % (name, fmt(pow(2, 261, mod))))
        for k in (2, 4, 5):
            out.append("#define ZK_F9_%s_%dM %s    /* %d M, strict limbs */" % (name, k, "{ " + ", ".join("0x%08x" % ((k * mod >> (29 * i)) & ((1 << 29) - 1) if i < 8 else (k * mod) >> 232) for i in range(9)) + " }", k))
        out.append("#define ZK_F9_%s_SCALE32 { %s }   /* 32 in 2^256-Montgomery words: x -> 2^5 x moves a stored value to the 2^261 form */"
                   % (name, ", ".join("0x%08xu" % x for x in limbs((32 << 256) % mod))))
    open("field9_consts.inc", "w").write("\n".join(out) + "\n")


def main():
    gen_field9()
    out = ["// GENERATED by gen_mul_asm.py -- do not edit.",
           "// ZK_MUL_ASM_*: clobbers v16-v27, s16-s24, vcc; the product leaves the block fully reduced",
           "// ZK_ADD_ASM_* + ZK_RED_ASM_* = modular addition, ZK_SUB_ASM_* + ZK_SUBFIX_ASM_* = modular subtraction (two statements each:",
           "// an asm statement takes at most 30 operands); they clobber vcc only, scratch registers are operands"]
    for name, mod in (("FP", P), ("FR", R)):
        lines = gen(mod) + gen_reduce(mod, ["%%%d" % i for i in range(8)], ["v%d" % (20 + i) for i in range(8)])
        out.append("#define ZK_MUL_ASM_%s \\" % name)
        out.append(" \\\n".join('    "%s\\n\\t"' % l for l in lines))
        out.append("")
        # weak form: operands and result in [0, 2 mod) -- no conditional subtraction (4 mod < 2^256)
        out.append("#define ZK_MULW_ASM_%s \\" % name)
        out.append(" \\\n".join('    "%s\\n\\t"' % l for l in gen(mod)))
        out.append("")
        for tag, fn in (("ADD", gen_add), ("SUB", gen_sub), ("SUBFIX", gen_subfix)):
            out.append("#define ZK_%s_ASM_%s \\" % (tag, name))
            out.append(" \\\n".join('    "%s\\n\\t"' % l for l in fn(mod)))
            out.append("")
        # weak form ([0, 2 mod) operands): conditional subtraction of 2 mod after an addition, + 2 mod after a borrowing subtraction
        out.append("#define ZK_REDW_ASM_%s \\" % name)
        out.append(" \\\n".join('    "%s\\n\\t"' % l for l in gen_reduce(2 * mod, ["%%%d" % i for i in range(8)], ["%%%d" % (8 + i) for i in range(8)])))
        out.append("")
        out.append("#define ZK_SUBFIXW_ASM_%s \\" % name)
        out.append(" \\\n".join('    "%s\\n\\t"' % l for l in gen_subfix(2 * mod)))
        out.append("")
        out.append("#define ZK_RED_ASM_%s \\" % name)
        out.append(" \\\n".join('    "%s\\n\\t"' % l for l in gen_reduce(mod, ["%%%d" % i for i in range(8)], ["%%%d" % (8 + i) for i in range(8)])))
        out.append("")
    open("field_mul_asm.inc", "w").write("\n".join(out) + "\n")
    print("MACs per mul:", sum(1 for l in gen(P) if "v_mad_u64_u32" in l), "instructions:", len(gen(P)))


if __name__ == "__main__":
    main()
