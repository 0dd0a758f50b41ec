"""Where the side-stream gather sits in the overlapped step, over a whole kernel trace:
usage: tl_stats.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fwd = [r for r in rows if 'gemm_x3_kernel<64>' in r['Kernel_Name']]
gat = [r for r in rows if 'gather_reduce_fwd_kernel' in r['Kernel_Name']]
adam = [r for r in rows if 'adam' in r['Kernel_Name']]
S = lambda r: int(r['Start_Timestamp']); E = lambda r: int(r['End_Timestamp'])
import bisect
fs = [S(r) for r in fwd]
offs, durs, steps, waits = [], [], [], []
for g in gat:
    i = bisect.bisect_right(fs, S(g)) - 1
    if i < 0 or i + 1 >= len(fs):
        continue
    offs.append((S(g) - fs[i]) / 1e3)
    durs.append((E(g) - S(g)) / 1e3)
    steps.append((fs[i + 1] - fs[i]) / 1e3)
    waits.append((fs[i + 1] - E(g)) / 1e3)
def q(v, p):
    v = sorted(v); return v[min(len(v) - 1, int(p * len(v)))]
n = len(offs)
print("steps", n)
for name, v in (("gather start after its step's forward start (us)", offs), ("gather duration", durs),
                ("step length (forward start to forward start)", steps), ("next forward start after gather end", waits)):
    v2 = v[n // 4:]
    print("%-55s 10/50/90 %%: %.1f %.1f %.1f" % (name, q(v2, 0.1), q(v2, 0.5), q(v2, 0.9)))
