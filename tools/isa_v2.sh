#!/bin/bash
# usage: tools/isa_v2.sh <layout 0|1|2> <TM> <TN>   - prints the biggest MFMA basic block of the v2 kernel
cd /root/repo/vilbert-multi-task_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DVB_V2_LAYOUT=$1 -S --cuda-device-only gemm_v2.hip -o /tmp/v2_$1.s 2>&1 | grep -v warning | head
python3 - $1 $2 $3 <<'PY'
import re,sys
s=open('/tmp/v2_%s.s'%sys.argv[1]).read()
name=[n for n in re.findall(r'^(_Z\w*gemm_v2_kernelILi%sELi%sELi0E\w*):'%(sys.argv[2],sys.argv[3]),s,flags=re.M)][0]
i=s.index(name+':'); body=s[i:s.index('.Lfunc_end',i)]
blocks=re.split(r'\n(?=\.LBB)',body)
best=max(blocks,key=lambda b:b.count('v_mfma'))
print(best.count('v_mfma'), 'mfma in block; lines', best.count('\n'))
out=[]
for l in best.split('\n'):
    l=l.split(';')[0].rstrip()
    if l.strip(): out.append(l.strip())
print('\n'.join(out))
PY
