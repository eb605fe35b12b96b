"""Which kernels of two `hipcc -S --cuda-device-only` listings have the same instruction text (labels renumbered, comments dropped)?
python scripts/isa_same_kernels.py old.s new.s  -- used to show that a new variant of a textually included kernel (lga_apply_pp.inc,
lga_filter_grad_pp.inc) left every existing instantiation as it was."""
import re,sys
def kernels(path):
    t=open(path).read().split('\n')
    out={}
    i=0
    while i<len(t):
        m=re.match(r'^(_ZN2ga\S+):', t[i])
        if m:
            j=i
            while 's_endpgm' not in t[j]: j+=1
            body=[l for l in t[i+1:j+1] if l.strip() and not l.strip().startswith(';')]
            body=[l.split(';')[0].rstrip() for l in body]
            body=[re.sub(r'\.(LBB|Ltmp|LJTI|Lfunc_\w+)\d+(_\d+)?', r'.\1', l) for l in body]
            out[m.group(1)]=body
            i=j
        i+=1
    return out
a=kernels(sys.argv[1]); b=kernels(sys.argv[2])
same=diff=0
for k in a:
    if k not in b: print('missing', k[:60]); continue
    if a[k]==b[k]: same+=1
    else:
        diff+=1
        d=[(x,y) for x,y in zip(a[k],b[k]) if x!=y][:2]
        print('DIFF', k[:70], len(a[k]), len(b[k]), d)
print('same',same,'diff',diff,'new',len(set(b)-set(a)))
