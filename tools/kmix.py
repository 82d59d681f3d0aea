"""Instruction mix of one fused-kernel instantiation from a `hipcc -save-temps` assembly file: whole kernel and the
row loop (largest backward-branch loop).  usage: kmix.py file.s [mangled-name-substring]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else "fused_kernel"
for f in re.split(r'\n(?=_ZN5fdoct\w+:)', s):
    name = f.split(':')[0]
    if sub not in name or 'fused_kernel' not in name:
        continue
    f = f.split('.Lfunc_end')[0]
    lines = f.split('\n')
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, ln in enumerate(lines):
        m = re.search(r'\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', ln)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            n = i - labels[m.group(2)]
            if best is None or n > best[0]:
                best = (n, labels[m.group(2)], i)
    body = lines[best[1]:best[2]]
    ins = [l.strip().split()[0] for l in body if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter(ins)
    cyc = 0
    for k, v in c.items():
        if k.startswith('v_pk_') or k.endswith('_f64') or 'f64' in k:
            cyc += 4 * v
        elif k.startswith(('v_sqrt', 'v_log', 'v_rcp', 'v_rsq', 'v_exp')):
            cyc += 8 * v
        elif k.startswith('v_'):
            cyc += 2 * v
    print(name[:120])
    print("loop lines %d-%d: instrs %d VALU %d DS %d SALU %d VMEM %d  est. VALU pipe cycles %d" % (
        best[1], best[2], len(ins), sum(v for k, v in c.items() if k.startswith('v_')),
        sum(v for k, v in c.items() if k.startswith('ds_')), sum(v for k, v in c.items() if k.startswith('s_')),
        sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_'))), cyc))
    print('  '.join("%s:%d" % kv for kv in c.most_common(60)))
