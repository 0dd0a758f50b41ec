"""Innermost MFMA loop of a kernel in a hipcc -S listing: python scratch/asm_loop.py file.s <name substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(lines) if name in l and not l.startswith('\t') and ':' in l.split(';')[0])
end = start
while 's_endpgm' not in lines[end]:
    end += 1
body = lines[start:end + 1]
mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
best = None
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        lo, hi = labels[m.group(1)], i
        nm = sum(1 for k in mf if lo <= k <= hi)
        if nm and (best is None or hi - lo < best[1] - best[0]):
            best = (lo, hi, nm)
lo, hi, nm = best
ops = [x.split()[0] for x in body[lo:hi + 1] if x.startswith('\t') and not x.startswith('\t.') and not x.startswith('\t;')]
c = collections.Counter(ops)
print(name, "innermost MFMA loop: MFMA", nm, "VALU", sum(n for o, n in c.items() if o.startswith('v_') and 'mfma' not in o),
      "SALU", sum(n for o, n in c.items() if o.startswith('s_')), "LDS", sum(n for o, n in c.items() if o.startswith('ds_')),
      "global", sum(n for o, n in c.items() if o.startswith('global_')))
print("  ", c.most_common(24))
