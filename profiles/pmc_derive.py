#!/usr/bin/env python3
"""Derived SQ metrics per (kernel, grid) from the output of profiles/pmc_summary.py:
  cu_busy  = SQ_BUSY_CU_CYCLES / (256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)      share of the launch's CU-cycles with a wave resident
  mfma     = SQ_VALU_MFMA_BUSY_CYCLES / 4 SIMDs / SQ_BUSY_CU_CYCLES        MFMA pipe busy, of the CU-busy cycles
  lds      = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES                         LDS array active, of the CU-busy cycles
  wait     = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                             wave-cycles spent waiting on any counter
usage: pmc_derive.py <pmc_summary output>"""
import collections
import re
import sys

acc = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+grid=\s*(\d+)\s+(\S+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", line)
    if m:
        acc[(m.group(1).strip(), int(m.group(2)))][m.group(3)] = (float(m.group(5)), int(m.group(4)))
print(f"{'kernel':60s} {'grid':>8s} {'n':>4s} {'cu_busy':>8s} {'mfma':>6s} {'lds':>6s} {'wait':>6s} {'us@2.4GHz':>10s}")
for (name, grid), c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', (0, 0))[0]):
    need = ('SQ_BUSY_CU_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES')
    if not all(k in c for k in need):
        continue
    busy, grbm = c['SQ_BUSY_CU_CYCLES'][0], c['GRBM_GUI_ACTIVE'][0]
    cu_cycles = 256.0 * grbm / 8.0
    mfma = c['SQ_VALU_MFMA_BUSY_CYCLES'][0] / 4.0 / busy if busy else 0.0
    lds = c.get('SQ_LDS_IDX_ACTIVE', (0.0, 0))[0] / busy if busy else 0.0
    wait = c.get('SQ_WAIT_INST_ANY', (0.0, 0))[0] / c['SQ_WAVE_CYCLES'][0]
    print(f"{name[:60]:60s} {grid:8d} {c['SQ_WAVE_CYCLES'][1]:4d} {busy / cu_cycles:8.2f} {mfma:6.2f} {lds:6.2f} {wait:6.2f} "
          f"{grbm / 8.0 / 2400.0:10.1f}")
