# generates the stage-statement form of N interleaved products for the JIT prelude
def gen(N):
    L=[]
    A=lambda s: L.append(s)
    A(f"template <> __device__ inline __attribute__((always_inline)) void lz_mulN<{N}>(u64 (&r)[{N}], const u64 (&a)[{N}], const u64 (&b)[{N}]) {{")
    A(f"  u64 p00[{N}], m[{N}], hi[{N}], t[{N}], lo[{N}], acc[{N}], cm[{N}], k1[{N}], k2[{N}], k3[{N}], c1[{N}], bb[{N}], bw[{N}], c3[{N}], mk[{N}], d[{2*N}];")
    A(f"  u32 w1[{N}], accl[{N}], acch[{N}], rl[{N}], rh[{N}];")
    A("  const u32 zero = 0;")
    gap = '"s_nop 0\\n\\t" ' if N==2 else ''
    def stmt(instrs, outs, ins, lead=''):
        # instrs: list of template strings using {o0}.. placeholders resolved by order
        text = '\\n\\t'.join(instrs)
        A(f'  asm({lead}"{text}"\n      : {", ".join(outs)}\n      : {", ".join(ins)});')
    # helper to build numbered operands
    def build(per_product, two_pass=False):
        """per_product(i) -> list of (template, outs[list of (constraint, expr)], ins[list of (constraint, expr)]) ; numbering assigned globally"""
        items=[]
        for i in range(N): items += per_product(i)
        outs=[]; ins=[]
        for tpl,o,inn in items: outs += o
        no=len(outs)
        texts=[]
        oi=0; ii=no
        for tpl,o,inn in items:
            names={}
            for k,(c,e) in enumerate(o): names[f'o{k}']=f'%{oi}'; oi+=1
            for k,(c,e) in enumerate(inn):
                if c.isdigit():  # tied to this instr's output index c
                    names[f'i{k}']=names[f'o{c}']
                else:
                    names[f'i{k}']=f'%{ii}'
                ii+=1
            texts.append(tpl.format(**names))
        outs_s=[f'"{c}"({e})' for c,e in outs]
        ins_s=[]
        # resolve tied constraints to absolute output numbers
        oi=0
        for tpl,o,inn in items:
            base=oi
            for c,e in inn:
                if c.isdigit(): ins_s.append(f'"{base+int(c)}"({e})')
                else: ins_s.append(f'"{c}"({e})')
            oi+=len(o)
        return texts,outs_s,ins_s
    # S1+S2: p00 = a0 b0 ; m = a0 b1
    def s12(i):
        return [("v_mad_u64_u32 {o0}, {o1}, {i0}, {i1}, 0",[("=&v",f"p00[{i}]"),("=&s",f"d[{2*i}]")],[("v",f"jlo(a[{i}])"),("v",f"jlo(b[{i}])")]),
                ("v_mad_u64_u32 {o0}, {o1}, {i0}, {i1}, 0",[("=&v",f"m[{i}]"),("=&s",f"d[{2*i+1}]")],[("v",f"jlo(a[{i}])"),("v",f"jhi(b[{i}])")])]
    t,o,n=build(s12); stmt(t,o,n)
    # S3: m += a1 b0 (carry cm)
    def s3(i): return [("v_mad_u64_u32 {o0}, {o1}, {i0}, {i1}, {i2}",[("=&v",f"m[{i}]"),("=&s",f"cm[{i}]")],[("v",f"jhi(a[{i}])"),("v",f"jlo(b[{i}])"),("0",f"m[{i}]")])]
    t,o,n=build(s3); stmt(t,o,n)
    # S4: w1 = p00.hi + m.lo (k1)
    def s4(i): return [("v_add_co_u32_e64 {o0}, {o1}, {i0}, {i1}",[("=&v",f"w1[{i}]"),("=&s",f"k1[{i}]")],[("v",f"jhi(p00[{i}])"),("v",f"jlo(m[{i}])")])]
    t,o,n=build(s4); stmt(t,o,n)
    # S5: accl = m.hi + k1 (k2)
    def s5(i): return [("v_addc_co_u32_e64 {o0}, {o1}, {i0}, 0, {i1}",[("=&v",f"accl[{i}]"),("=&s",f"k2[{i}]")],[("v",f"jhi(m[{i}])"),("s",f"k1[{i}]")])]
    t,o,n=build(s5); stmt(t,o,n,gap)
    A(f"  for (int i = 0; i < {N}; i++) k3[i] = cm[i] | k2[i];  // scalar unit; cm and k2 exclude each other (a carried m leaves m.hi <= 2^32 - 5)")
    # S6: acch = 0 + k3
    def s6(i): return [("v_addc_co_u32_e64 {o0}, {o1}, {i0}, 0, {i1}",[("=&v",f"acch[{i}]"),("=&s",f"d[{i}]")],[("v","zero"),("s",f"k3[{i}]")])]
    t,o,n=build(s6); stmt(t,o,n)
    A(f"  for (int i = 0; i < {N}; i++) {{ acc[i] = ((u64)acch[i] << 32) | accl[i]; lo[i] = ((u64)w1[i] << 32) | jlo(p00[i]); }}")
    # S7: hi = a1 b1 + acc
    def s7(i): return [("v_mad_u64_u32 {o0}, {o1}, {i0}, {i1}, {i2}",[("=&v",f"hi[{i}]"),("=&s",f"d[{i}]")],[("v",f"jhi(a[{i}])"),("v",f"jhi(b[{i}])"),("v",f"acc[{i}]")])]
    t,o,n=build(s7); stmt(t,o,n)
    # S8: t = hi.lo * (2^32 - 1) + lo (c1)
    def s8(i): return [("v_mad_u64_u32 {o0}, {o1}, {i0}, -1, {i1}",[("=&v",f"t[{i}]"),("=&s",f"c1[{i}]")],[("v",f"jlo(hi[{i}])"),("v",f"lo[{i}]")])]
    t,o,n=build(s8); stmt(t,o,n)
    # S9 (all products) then S10 (all products) in ONE statement: rl = t.lo - hi.hi - c1 (bb); rh = t.hi + c1
    def s9(i): return [("v_subb_co_u32_e64 {o0}, {o1}, {i0}, {i1}, {i2}",[("=&v",f"rl[{i}]"),("=&s",f"bb[{i}]")],[("v",f"jlo(t[{i}])"),("v",f"jhi(hi[{i}])"),("s",f"c1[{i}]")])]
    def s10(i): return [("v_addc_co_u32_e64 {o0}, {o1}, {i0}, 0, {i1}",[("=&v",f"rh[{i}]"),("=&s",f"d[{i}]")],[("v",f"jhi(t[{i}])"),("s",f"c1[{i}]")])]
    # build S9 for all i, then S10 for all i, one numbering
    items9=[x for i in range(N) for x in s9(i)]; items10=[x for i in range(N) for x in s10(i)]
    seq=items9+items10
    cnt=[0]
    def per(i):
        return []
    # manual build over seq
    outs=[]; 
    for tpl,o,inn in seq: outs+=o
    no=len(outs); oi=0; ii=no; texts=[]; ins_s=[]
    for tpl,o,inn in seq:
        names={}
        for k,(c,e) in enumerate(o): names[f'o{k}']=f'%{oi}'; oi+=1
        for k,(c,e) in enumerate(inn): names[f'i{k}']=f'%{ii}'; ii+=1; ins_s.append(f'"{c}"({e})')
        texts.append(tpl.format(**names))
    stmt(texts,[f'"{c}"({e})' for c,e in outs],ins_s,gap)
    # S11: rh -= bb (bw)
    def s11(i): return [("v_subb_co_u32_e64 {o0}, {o1}, {i0}, 0, {i1}",[("=&v",f"rh[{i}]"),("=&s",f"bw[{i}]")],[("0",f"rh[{i}]"),("s",f"bb[{i}]")])]
    t,o,n=build(s11); stmt(t,o,n)
    # S12: rl += bw (c3)
    def s12b(i): return [("v_addc_co_u32_e64 {o0}, {o1}, {i0}, 0, {i1}",[("=&v",f"rl[{i}]"),("=&s",f"c3[{i}]")],[("0",f"rl[{i}]"),("s",f"bw[{i}]")])]
    t,o,n=build(s12b); stmt(t,o,n,gap)
    A(f"  for (int i = 0; i < {N}; i++) mk[i] = bw[i] & ~c3[i];  // scalar unit")
    # S13: rh -= mk
    def s13(i): return [("v_subb_co_u32_e64 {o0}, {o1}, {i0}, 0, {i1}",[("=&v",f"rh[{i}]"),("=&s",f"d[{i}]")],[("0",f"rh[{i}]"),("s",f"mk[{i}]")])]
    t,o,n=build(s13); stmt(t,o,n)
    A(f"  for (int i = 0; i < {N}; i++) r[i] = ((u64)rh[i] << 32) | rl[i];")
    A("}")
    return "\n".join(L)
if __name__=="__main__":
    print("\n".join(gen(N) for N in (2,3,4)))
