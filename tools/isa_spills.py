"""tools/isa_spills.py <file.s> <kernel-name-substring>: static instruction mix of the kernels of an AMDGPU assembly listing
(tools/kernel_regs.sh writes /tmp/isa/msk_gpu-hip-amdgcn-amd-amdhsa-gfx950.s): VALU / SALU counts, v_writelane / v_readlane (SGPR
spills into VGPR lanes: VALU instructions both), fp64, scalar loads, LDS and global / buffer memory instructions.  Round 5 used
it for the shading kernel's SGPR diet; profiles/r05_shade_spills.txt is its output for the three variants."""
import sys,re,collections
s=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2]
out={}
cur=None
for l in s:
    m=re.match(r'^(_ZN3msk\S*):',l)
    if m: cur=m.group(1); out[cur]=collections.Counter(); continue
    if l.startswith('\t.end_amdhsa_kernel') or l.startswith('.Lfunc_end'): cur=None
    if cur and l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        op=l.split()[0]
        out[cur][op]+=1
for k,c in out.items():
    if pat in k:
        tot=sum(c.values())
        valu=sum(v for o,v in c.items() if o.startswith('v_'))
        salu=sum(v for o,v in c.items() if o.startswith('s_'))
        print(k[:70],'total',tot,'valu',valu,'salu',salu,'writelane',c['v_writelane_b32'],'readlane',c['v_readlane_b32'],'f64',sum(v for o,v in c.items() if 'f64' in o),'s_load',sum(v for o,v in c.items() if o.startswith('s_load')), 'ds', sum(v for o,v in c.items() if o.startswith('ds_')), 'glob', sum(v for o,v in c.items() if o.startswith('global_') or o.startswith('buffer_')))
