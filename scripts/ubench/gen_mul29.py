#!/usr/bin/env python3
"""Prototype: Montgomery multiplication on 9 x 29-bit limbs (R = 2^261), one inline-asm block, NO carry instructions:
column sums of up to 18 products < 2^58 fit a 64-bit v_mad_u64_u32 accumulator."""
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
def gen(mod):
    L=[]
    mask=(1<<29)-1
    for i in range(9): L.append("s_mov_b32 s%d, 0x%08x"%(16+i,(mod>>(29*i))&mask))
    L.append("s_mov_b32 s25, 0x%08x"%((-pow(mod,-1,1<<29))%(1<<29)))
    L.append("s_mov_b32 s26, 0x%08x"%mask)
    a=lambda i:"%%%d"%(9+i); b=lambda i:"%%%d"%(18+i); t=lambda i:"%%%d"%i
    first=True
    def mac(x,y):
        nonlocal first
        if first: L.append("v_mad_u64_u32 v[16:17], vcc, %s, %s, 0"%(x,y)); first=False
        else: L.append("v_mad_u64_u32 v[16:17], vcc, %s, %s, v[16:17]"%(x,y))
    for k in range(9):
        for i in range(k+1): mac(a(i),b(k-i))
        for i in range(k): mac(t(i),"s%d"%(16+k-i))
        L.append("v_mul_lo_u32 %s, v16, s25"%t(k))
        L.append("v_and_b32 %s, s26, %s"%(t(k),t(k)))
        mac(t(k),"s16")
        L.append("v_lshrrev_b64 v[16:17], 29, v[16:17]")
    for k in range(9,17):
        for i in range(k-8,9): mac(a(i),b(k-i))
        for i in range(k-8,9): mac(t(i),"s%d"%(16+k-i))
        L.append("v_and_b32 %s, s26, v16"%t(k-9))
        L.append("v_lshrrev_b64 v[16:17], 29, v[16:17]")
    L.append("v_mov_b32 %s, v16"%t(8))
    return L
lines=gen(P)
open("mul29_asm.inc","w").write("#define ZK_MUL29_ASM_FP \\\n"+" \\\n".join('    "%s\\n\\t"'%l for l in lines)+"\n")
print(len(lines), sum('v_mad' in l for l in lines))
