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


def listing(path, start_frac=0.0, count=400):
    con = sqlite3.connect(path)
    cur = con.cursor()
    t = tables(cur)
    kd = [x for x in t if x.startswith('rocpd_kernel_dispatch')][0]
    ks = [x for x in t if x.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    i0 = int(len(rows) * start_frac)
    prev_end = None
    for name, st, en, gx, gy, wx in rows[i0:i0 + count]:
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        prev_end = en
        short = name.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')[:60]
        print('%-62s wgs=%7d dur=%9.1fus gap=%7.1fus' % (short, gx // max(wx, 1) * max(gy, 1), (en - st) / 1e3, gap))


def stats_by_grid(path):
    """Per (kernel, workgroup count): launches of one template instance that do different jobs (a grouped Winograd GEMM and a 1x1
    convolution run the same conv_fwd_dma_kernel instantiation) are told apart by their grids."""
    con = sqlite3.connect(path)
    cur = con.cursor()
    t = tables(cur)
    kd = [x for x in t if x.startswith('rocpd_kernel_dispatch')][0]
    ks = [x for x in t if x.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute(f"select s.kernel_name, d.grid_size_x / max(d.workgroup_size_x, 1) * max(d.grid_size_y, 1), count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, "
                            f"min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, 2 order by 4 desc"))
    tot = sum(r[3] for r in rows)
    print('%-92s %8s %6s %12s %10s %10s %10s %6s' % ('kernel', 'wgs', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for r in rows:
        print('%-92s %8d %6d %12.1f %10.1f %10.1f %10.1f %6.2f' % (r[0].replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')[:92], r[1], r[2], r[3], r[4], r[5], r[6], 100 * r[3] / tot))


if __name__ == '__main__':
    if sys.argv[1] == 'by_grid':
        stats_by_grid(sys.argv[2])
    elif sys.argv[1] == 'list':
        listing(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.0, int(sys.argv[4]) if len(sys.argv) > 4 else 400)
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
