import re, collections, sys
s=open('/tmp/rtx_g.s').read()
name=sys.argv[1] if len(sys.argv)>1 else '_Z13render_kernelILb0ELb1ELb0EEv12RenderParams'
i=s.index(name+':'); j=s.index('.Lfunc_end', i)
files={}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[int(m.group(1))]=(m.group(3) or m.group(2)).split('/')[-1]
src={f:open('/root/repo/ray-tracer_amd/csrc/'+f).read().split('\n') for f in ('rtx_device.hpp','rtx.hip')}
cur=None
cnt=collections.Counter(); tot=0
for l in s[i:j].split('\n'):
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur=(files.get(int(m.group(1)),'?'), int(m.group(2))); continue
    t=l.strip()
    if re.match(r'v_[a-z]', t):
        cnt[cur]+=1; tot+=1
print('total static VALU', tot)
def region(f,l):
    if f=='rtx_device.hpp':
        for k in range(l-1,0,-1):
            m=re.match(r'^__device__.*?\b(\w+)\(', src[f][k])
            if m: return 'dev:'+m.group(1)
        return 'dev:?'
    if f=='rtx.hip':
        return 'hip:%d %s'%(l, src[f][l-1].strip()[:60])
    return f
reg=collections.Counter()
for (k,v) in cnt.items():
    if k is None: reg['none']+=v; continue
    reg[region(*k)]+=v
for k,v in sorted(reg.items(), key=lambda x:-x[1])[:45]: print(f'{v:5d}  {k}')
