import re,collections,sys
s=open(sys.argv[1]).read()
kn=sys.argv[2]
k=s[s.index(kn+':'):]
k=k[:k.index('.Lfunc_end')]
blocks=[];cur=[];name='entry'
for l in k.splitlines():
    t=l.strip()
    if re.match(r'^\.LBB\d+_\d+:',t):
        blocks.append((name,cur)); cur=[]; name=t.split(':')[0]
    elif t and not t.startswith(('.',';','//')):
        cur.append(t)
blocks.append((name,cur))
for n,b in blocks:
    c=collections.Counter(x.split()[0] for x in b)
    m=sum(v for kk,v in c.items() if 'mfma' in kk)
    if m<8: continue
    gaps=[];g=0
    for x in b:
        if 'v_mfma' in x: gaps.append(g); g=0
        else: g+=1
    print(n,len(b),'mfma',m,'accrd',c.get('v_accvgpr_read_b32',0),'accwr',c.get('v_accvgpr_write_b32',0),'pkmul',c.get('v_pk_mul_f32',0),'exp',c.get('v_exp_f32_e32',0),'ds',sum(v for kk,v in c.items() if kk.startswith('ds_')),'waitcnt',c.get('s_waitcnt',0),'nop',c.get('s_nop',0))
    print('   gaps',gaps)
