#!/usr/bin/env python3
"""Summarise rocprofv3 sqlite outputs (ROCm 7.2 writes rocpd .db files): per-kernel stats and per-kernel PMC means.

    python tools/rocprof_summary.py stats <trace.db>          -> per-kernel count / total / avg / min / max (us)
    python tools/rocprof_summary.py pmc <pmc.db> [<pmc.db>..] -> per (kernel, counter): mean per dispatch summed over instances
"""
import collections
import sqlite3
import sys


def tables(cur):
    return [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]


def stats(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    t = tables(cur)
    kd = [x for x in t if x.startswith('rocpd_kernel_dispatch')][0]
    ks = [x for x in t if x.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
                            f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print('%-100s %6s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for r in rows:
        print('%-100s %6d %12.1f %10.1f %10.1f %10.1f %6.2f' % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))


def pmc(paths):
    for path in paths:
        con = sqlite3.connect(path)
        cur = con.cursor()
        t = tables(cur)
        ev = [x for x in t if x.startswith('rocpd_pmc_event')][0]
        info = [x for x in t if x.startswith('rocpd_info_pmc')][0]
        kd = [x for x in t if x.startswith('rocpd_kernel_dispatch')][0]
        ks = [x for x in t if x.startswith('rocpd_info_kernel_symbol')][0]
        q = (f"select s.kernel_name, p.name, d.dispatch_id, sum(e.value), (d.end-d.start)/1e3 from {ev} e join {info} p on e.pmc_id=p.id "
             f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, p.name, d.dispatch_id")
        agg = collections.defaultdict(list)
        for k, c, _, v, us in cur.execute(q):
            agg[(k, c)].append((v, us))
        for (k, c), vals in sorted(agg.items()):
            n = len(vals)
            print('%-90s %-28s dispatches=%4d mean_per_dispatch=%.6g mean_us=%.1f' % (k[:90], c, n, sum(v for v, _ in vals) / n, sum(u for _, u in vals) / n))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
