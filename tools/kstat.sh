#!/bin/bash
# Compiles the kernels (dev-single mode unless FULL=1) and prints registers / scratch / loop instruction mix
# of one instantiation.  usage: tools/kstat.sh [mangled-substring]
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
sub=${1:-Li10ELi64ELi16ELi4ELi16ELi1ELi4EtLb0ELb1E}
mkdir -p /tmp/kstat && cd /tmp/kstat
flags="-DFDOCT_DEV_SINGLE"; [ "$FULL" = 1 ] && flags=""
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags $EXTRA -save-temps -c "$root/fdoct_amd/csrc/fdoct_kernels.hip" -o k.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A7 "$sub" | grep -E "VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: *//'
python3 - "$sub" <<'PY'
import re,sys
from collections import Counter
s=open('/tmp/kstat/fdoct_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for f in re.split(r'\n(?=_ZN5fdoct\w+:)', s):
    name=f.split(':')[0]
    if sys.argv[1] in name:
        lines=f.split('\n'); labels={}
        for i,ln in enumerate(lines):
            m=re.match(r'^(\.LBB\d+_\d+):',ln)
            if m: labels[m.group(1)]=i
        best=None
        for i,ln in enumerate(lines):
            m=re.search(r'\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)',ln)
            if m and m.group(2) in labels and labels[m.group(2)]<i:
                n=i-labels[m.group(2)]
                if best is None or n>best[0]: best=(n,labels[m.group(2)],i)
        body=lines[best[1]:best[2]]
        ins=[l.strip().split()[0] for l in body if l.strip() and not l.strip().startswith((';','.'))]
        c=Counter(ins)
        print("loop lines %d-%d instrs %d VALU %d DS %d SALU %d scratch %d" % (best[1],best[2],len(ins),sum(v for k,v in c.items() if k.startswith('v_')),sum(v for k,v in c.items() if k.startswith('ds_')),sum(v for k,v in c.items() if k.startswith('s_')),sum(v for k,v in c.items() if k.startswith('scratch'))))
        print('  '.join(f"{k}:{v}" for k,v in c.most_common(40)))
        break
PY
