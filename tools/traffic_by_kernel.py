#!/usr/bin/env python
"""Per-kernel HBM-side traffic table from a tools/rocprof_summary.py `pmc` text (FETCH_SIZE x 2 + WRITE_SIZE, KiB per dispatch; gfx950 correction as in
tools/traffic_from_pmc.py): MB per step, launches per step, MB per launch, mean duration, TB/s.

    python tools/traffic_by_kernel.py profiles/r04_train_b64_pmc_summary.txt 15 > profiles/r04_train_b64_traffic_by_kernel.txt"""
import collections
import re
import sys

rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r'(\S+)\s+(\S+)\s+dispatches=\s*(\d+)\s+mean_per_dispatch=([0-9.eE+-]+)\s+mean_us=([0-9.]+)', line)
    if m:
        rows[m.group(1)][m.group(2)] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else max([d['FETCH_SIZE'][0] for k, d in rows.items() if 'loss_fwd_kernel' in k and 'FETCH_SIZE' in d] + [1])
tab = []
for k, d in rows.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        n, f, us = d['FETCH_SIZE']
        w = d['WRITE_SIZE'][1]
        mb = (2 * f + w) * 1024 / 1e6
        busy = d.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0, 0))[1]
        gui = d.get('GRBM_GUI_ACTIVE', (0, 0, 0))[1]
        tab.append((n * mb / steps, k, n / steps, mb, us, busy / (gui * 128) if gui else 0.0))
tab.sort(reverse=True)
print('%d steps profiled; total %.1f GB per step (FETCH_SIZE x 2 + WRITE_SIZE, L2 memory side, Infinity-Cache hits included)' % (steps, sum(t[0] for t in tab) / 1e3))
print('%10s %9s %11s %9s %7s %9s  %s' % ('MB/step', 'launches', 'MB/launch', 'mean us', 'TB/s', 'MFMA busy', 'kernel'))
for t in tab:
    if t[0] < 50:
        continue
    print('%10.0f %9.1f %11.1f %9.1f %7.2f %9.2f  %s' % (t[0], t[2], t[3], t[4], t[3] / t[4] if t[4] else 0, t[5], t[1][:110]))
