#!/bin/bash
# per-phase VALU / SALU of a kernel choice: usage bash tools/r3_phases.sh <tag> <KVQ_KERNEL> <KVQ_LG>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
KVQ_LG=$3 bash tools/r3_valu.sh $1 $2 "0 1 2 130 34" > /dev/null 2>&1
python3 - $1 <<'PY'
import sys,re
t=open('gpurun_out/%s/valu.txt'%sys.argv[1]).read()
d={}
for blk in t.split('== ')[1:]:
    k=blk.split()[1].split('=')[1]
    d[k]={m.group(1):float(m.group(2)) for m in re.finditer(r'(SQ_\w+)\s+(\d+)',blk)}
n=5e6
def per(k,c): return d[k][c]/n
names=[('front','34',None),('trim','130','34'),('scaffold','2','130'),('filter','1','2'),('verify','0','1')]
for nm,a,b in names:
    v=per(a,'SQ_INSTS_VALU')-(per(b,'SQ_INSTS_VALU') if b else 0); s=per(a,'SQ_INSTS_SALU')-(per(b,'SQ_INSTS_SALU') if b else 0)
    l=per(a,'SQ_INSTS_LDS')-(per(b,'SQ_INSTS_LDS') if b else 0); c=(d[a]['SQ_BUSY_CYCLES']-(d[b]['SQ_BUSY_CYCLES'] if b else 0))/1e6
    print('%-9s VALU %5.1f  SALU %5.1f  LDS %4.1f  busy %5.1fM' % (nm,v,s,l,c))
print('total     VALU %5.1f  SALU %5.1f  LDS %4.1f  busy %5.1fM (diagnostic build)' % (per('0','SQ_INSTS_VALU'),per('0','SQ_INSTS_SALU'),per('0','SQ_INSTS_LDS'),d['0']['SQ_BUSY_CYCLES']/1e6))
PY
