# Diagnostic: instruction classes of the pair loop of stft_chroma32_kernel<1,3,0> from the gfx950 assembly of fingerprint32.hip
#   hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Ineedle_amd/csrc --offload-arch=gfx950 -S --cuda-device-only -o /tmp/fp32.s needle_amd/csrc/fingerprint32.hip
#   python tools/isa_classify.py /tmp/fp32.s
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
# find kernel body lines between label and s_endpgm
start = next(i for i,l in enumerate(src) if l.startswith('_ZN6needle4stft20stft_chroma32_kernelILi1ELi3ELi0E'))
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
body = src[start:end+1]
# find loop: labels .LBB0_x and backward branches
labels = {}
for i,l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i,l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
print('backward branches:', [(a,b,c,b-a) for a,b,c in loops])
def classify(ins, ops):
    if ins.startswith('v_cvt'): return 'v_cvt'
    if ins.endswith('_dpp') or 'row_' in ops or 'quad_perm' in ops: return 'valu_dpp'
    if ins.endswith('_sdwa'): return 'valu_sdwa'
    if ins.startswith('v_pk_'): return 'v_pk'
    if ins.startswith('v_fma') or ins.startswith('v_fmac') or ins.startswith('v_mad'):
        return 'v_fma'
    if ins.startswith('v_mul_f32'): return 'v_mul_f32'
    if ins.startswith('v_add_f32') or ins.startswith('v_sub_f32') or ins.startswith('v_subrev_f32'): return 'v_add/sub_f32'
    if ins.startswith('v_mov'): return 'v_mov'
    if ins.startswith('v_'): return 'valu_other'
    if ins.startswith('ds_'): return ins
    if ins.startswith('global_') or ins.startswith('buffer_') or ins.startswith('flat_'): return ins
    if ins.startswith('s_waitcnt'): return 's_waitcnt'
    if ins.startswith('s_barrier'): return 's_barrier'
    if ins.startswith('s_'): return 'salu/other'
    return 'other'
for a,b,name in loops:
    if b-a < 300: continue
    cnt = collections.Counter(); enc = collections.Counter(); lit=0; three=0; bankc=0
    for l in body[a:b+1]:
        l=l.split(';')[0].strip()
        if not l or l.endswith(':') or l.startswith('.'): continue
        parts=l.split(None,1); ins=parts[0]; ops=parts[1] if len(parts)>1 else ''
        c=classify(ins,ops); cnt[c]+=1
        if ins.startswith('v_'):
            enc['e64' if ins.endswith('_e64') else 'e32' if ins.endswith('_e32') else 'other']+=1
            regs=re.findall(r'\bv(\d+)\b', ops)
            srcs=regs[1:] if regs else []
            if re.search(r'0x[0-9a-f]+', ops): lit+=1
            if len(srcs)>=3:
                three+=1
                banks=[int(x)%4 for x in srcs[:3]]
                if len(set(banks))<3: bankc+=1
    print('loop', name, 'lines', b-a)
    tot_valu=sum(v for k,v in cnt.items() if k.startswith('v_') or k.startswith('valu'))
    for k,v in sorted(cnt.items(), key=lambda x:-x[1]): print(f'  {k:28s} {v}')
    print('  total VALU', tot_valu, 'encodings', dict(enc), 'with literal', lit, 'three-VGPR-source', three, 'of which two sources share a bank (mod 4)', bankc)
