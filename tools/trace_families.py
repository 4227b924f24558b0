#!/usr/bin/env python3
"""Per kernel TEMPLATE time of the steady part of a rocprofv3 kernel trace (the `trace` member of profiles/*_traffic.json, which bench.py's
`roofline.frac` is computed from): {steps, families: {template name: {calls, total_us}}}.

    python tools/trace_families.py <trace.db> <step kernel> [start fraction]

step kernel: a kernel launched exactly once per step (loss_fwd_kernel for training, conv0_kernel for detect) - its dispatch count in the window
is the number of steps.  start fraction: dispatches before this fraction of the trace's time span are left out (warm-up passes, capture)."""
import collections
import json
import re
import sqlite3
import sys


def demangled_template(name):
    m = re.match(r'_ZN\d+_GLOBAL__N_1(\d+)', name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    m = re.match(r'_Z(\d+)', name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    return name.replace('.kd', '')


def main():
    path, step_kernel = sys.argv[1], sys.argv[2]
    start = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    cur = sqlite3.connect(path).cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if x.startswith('rocpd_kernel_dispatch')][0]
    ks = [x for x in t if x.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    # the window opens at the first step kernel after `start` of the time span and closes at the last one: whole steps only
    t0, t1 = rows[0][1], rows[-1][2]
    marks = [st for n, st, en in rows if step_kernel in n and st >= t0 + start * (t1 - t0)]
    fam = collections.OrderedDict()
    steps = max(len(marks) - 1, 0)
    if steps:
        for n, st, en in rows:
            if marks[0] <= st < marks[-1]:
                f = fam.setdefault(demangled_template(n), {'calls': 0, 'total_us': 0.0})
                f['calls'] += 1
                f['total_us'] += (en - st) / 1e3
    print(json.dumps({'steps': steps, 'window': 'from the first to the last %s dispatch after %.0f %% of the trace (whole steps)' % (step_kernel, 100 * start),
                      'families': {k: {'calls': v['calls'], 'total_us': round(v['total_us'], 1)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['total_us'])}}))


if __name__ == '__main__':
    main()
