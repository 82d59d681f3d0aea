"""Prints a compact schedule trace of a kernel's hot loop from hipcc -save-temps output:
v<n> = run of n VALU instructions, s<n> SALU, R/Wr/B = LDS reads / writes / bpermutes,
GL/GS = global loads / stores, |...| = s_waitcnt.  Usage: sched_trace.py file.s <mangled-name-substring> [from to]"""
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
for f in re.split(r'\n(?=_ZN5fdoct\w+:)', s):
    if sub not in f.split(':')[0]:
        continue
    L = f.split('\n')
    out, run, cnt = [], None, 0
    for l in L[lo:hi]:
        t = l.strip()
        if not t or t.startswith((';', '.')):
            continue
        op = t.split()[0]
        if op.startswith('ds_'):
            k = 'R' if 'read' in op else ('B' if 'permute' in op else 'Wr')
        elif op == 's_waitcnt':
            if run:
                out.append("%s%d" % (run, cnt))
            run, cnt = None, 0
            out.append('|' + t.split(None, 1)[1].replace('lgkmcnt', 'lgkm').replace('vmcnt', 'vm') + '|')
            continue
        elif op.startswith('global_load'):
            k = 'GL'
        elif op.startswith('global_store'):
            k = 'GS'
        elif op.startswith('scratch'):
            k = 'SCR'
        elif op.startswith('v_'):
            k = 'v'
        elif op.startswith('s_'):
            k = 's'
        else:
            k = '?'
        if k != run:
            if run:
                out.append("%s%d" % (run, cnt))
            run, cnt = k, 0
        cnt += 1
    if run:
        out.append("%s%d" % (run, cnt))
    print(len(L), 'lines')
    print(' '.join(out))
    break
