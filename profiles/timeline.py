#!/usr/bin/env python3
"""Print one training step's kernel timeline from a rocprofv3 --kernel-trace CSV
(usage: timeline.py <kernel_trace.csv> [step_from_end])."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name'] and 'sage::' in r['Kernel_Name']]      # adam_kernel, adam4_kernel, adam_parts*_kernel
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = idx[-k], idx[-k + 1]
t0 = int(rows[a + 1]['Start_Timestamp'])
prev = t0
for r in rows[a + 1:b + 2]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('sage::(anonymous namespace)::', '').replace('void ', '')[:70]
    print(f"{(s - t0) / 1e3:9.1f} gap={(s - prev) / 1e3:7.1f} dur={(e - s) / 1e3:8.1f} "
          f"grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} vgpr={r['VGPR_Count']:>3s}+{r['Accum_VGPR_Count']:>3s} "
          f"lds={r['LDS_Block_Size']:>6s} {name}")
    prev = e
