import re,sys,glob
# for each kernel in each gfx950 .s: count "store-wait-store" patterns: s_waitcnt vmcnt(N) that follows a store with no load in between
for f in sorted(glob.glob('/tmp/isa/*-hip-amdgcn-amd-amdhsa-gfx950.s'))+['/tmp/deconv3d-hip-amdgcn-amd-amdhsa-gfx950.s','/tmp/deconv3d_pl-hip-amdgcn-amd-amdhsa-gfx950.s']:
    name=None; pend_store=0; cnt={}
    stores={}
    for line in open(f):
        m=re.match(r'^(_Z\w+):',line)
        if m: name=m.group(1); pend_store=0; cnt[name]=0; stores[name]=0; continue
        if name is None: continue
        t=line.strip()
        if t.startswith('s_endpgm'): name=None; continue
        op=t.split()[0] if t else ''
        if op.startswith('global_store') or op.startswith('buffer_store'):
            pend_store+=1; stores[name]+=1
        elif op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('global_atomic'):
            pend_store=0   # a wait after a load is legit (conservative)
        elif op=='s_waitcnt' and 'vmcnt' in t:
            n=int(re.search(r'vmcnt\((\d+)\)',t).group(1))
            if pend_store>n:  # waits for at least one store
                cnt[name]+=1
            pend_store=min(pend_store,n)
    for k,v in cnt.items():
        if v>0: print(f.split('/')[-1].split('-hip')[0], k[:70], 'store-waits', v, 'of stores', stores[k])
