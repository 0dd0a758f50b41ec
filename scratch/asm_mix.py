"""Instruction mix of one kernel in a hipcc -S listing:  python scratch/asm_mix.py file.s <name substring> [...]
(hipcc -O3 --offload-arch=gfx950 -I include -S --cuda-device-only graphsage_amd/csrc/linear.hip -o /tmp/linear.s)"""
import collections
import sys

lines = open(sys.argv[1]).read().split('\n')
for name in sys.argv[2:]:
    start = next((i for i, l in enumerate(lines) if name in l and not l.startswith('\t') and ':' in l.split(';')[0]), None)
    if start is None:
        print("not found:", name)
        continue
    end = start
    while 's_endpgm' not in lines[end]:
        end += 1
    ops = [l.split()[0] for l in lines[start:end + 1] if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    c = collections.Counter(ops)
    tot = lambda pred: sum(n for o, n in c.items() if pred(o))
    print(name, "VALU", tot(lambda o: o.startswith('v_') and 'mfma' not in o), "MFMA", tot(lambda o: 'mfma' in o),
          "SALU", tot(lambda o: o.startswith('s_')), "LDS", tot(lambda o: o.startswith('ds_')),
          "global", tot(lambda o: o.startswith('global_')), "scratch", tot(lambda o: o.startswith('scratch_')))
    print("  ", c.most_common(26))
