#!/usr/bin/env python
"""Single-image detect latency (BASELINE configs[0] on the GPU, detect.py:141-153): the bench.py latency leg alone, with a FRESH per-layer
measurement (Y2_TUNE_DEFAULTS=0 by default here) so that kernel-side experiments are seen by the algorithm selection.

    python tools/latency_b1.py [batch ...]      # one JSON line per batch size"""
import json
import os
import sys
import time

os.environ.setdefault('Y2_TUNE_DEFAULTS', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import bench_data  # noqa: E402
import detect  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    inf, anchors = bench_data.build_model(20, dev, 'darknet')
    dnn = inf.dnn
    kw = dict(fix=True, threshold_cls=0.005, overlap=0.45, limit=200)
    for B in [int(v) for v in sys.argv[1:]] or [1]:
        x = bench_data.images(B, 416, seed=40 + B).to(dev)

        def eager(i):
            with torch.no_grad():
                return detect.detect_batch(dnn.forward_nhwc(x), anchors, **kw)
        for i in range(3):
            eager(i)
        torch.cuda.synchronize()
        table = bench.kernel_table(eager, 4)
        plan = dnn._plan_cache[1]
        algos = [(int(p.algo), int(p.tile)) for p in plan['arr'][:plan['n']]]
        g = detect.GraphedDetector(dnn, anchors, x, static_input=True, **kw)
        for _ in range(20):
            g.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            g.run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 300 * 1e3
        rows = bench.top_kernels(table, 0.02)[0]
        print(json.dumps({'batch': B, 'ms_per_step': round(ms, 4), 'launches': round(sum(e['launches'] for e in table.values()), 1),
                          'kernel_ms_sum_eager': round(sum(e['ms'] for e in table.values()), 4), 'algos': algos,
                          'top': [(r['kernel'], r['launches_per_step'], r['ms_per_step']) for r in rows]}))
        del g


if __name__ == '__main__':
    main()
