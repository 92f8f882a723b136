"""Tabulates a tools/lab_run.sh output file: one row per GEMM, one column per configuration (TFLOP/s)."""
import re
import sys

txt = open(sys.argv[1]).read()
secs = re.split(r'=== ', txt)[1:]
tab, names = {}, []
for sec in secs:
    name = sec.split('\n')[0]
    names.append(name)
    for l in sec.split('\n'):
        m = re.match(r'(\w+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) nseg=(\d)\s+([\d.]+) us\s+([\d.]+) TF\s+err (\S+)', l)
        if m:
            key = (m.group(1), m.group(3), m.group(4), m.group(5))
            tab.setdefault(key, {})[name] = (float(m.group(7)), float(m.group(8)))
    m = re.search(r'aggregate: ([\d.]+)', sec)
    tab.setdefault(('agg', '', '', ''), {})[name] = (float(m.group(1)), 0) if m else (0, 0)
print('%-22s' % 'shape' + ''.join('%15s' % n[-14:] for n in names))
for k, v in tab.items():
    print('%-22s' % ' '.join(k) + ''.join('%10.1f%s' % (v.get(n, (0, 0))[0], ' BAD ' if v.get(n, (0, 0))[1] > 1e-4 else '     ')
                                        for n in names))
