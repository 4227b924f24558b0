#!/usr/bin/env python
"""profiles/*_traffic.json from a tools/rocprof_summary.py `pmc` text: HBM-side bytes per bench step of the conv chain.

    python tools/traffic_from_pmc.py gpurun_out/prof/pmc_summary.txt > profiles/rNN_detect_b32_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch (TCC-EA requests, summed over the 16 channels/XCDs); on gfx950
FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) and is doubled.  The conv
chain = conv_fwd_dma_kernel family + split-K fix-up + the Winograd transform kernels; steps = conv0 dispatches."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo2-pytorch_amd'))
try:
    import _hip
    KERNELS = _hip.kernel_hash()
except Exception:
    KERNELS = None

FAMILY = re.compile(r'conv_fwd_dma_kernel|conv_splitk_fixup_kernel|wino_(input|output|fused\d*|tile_table)_kernel|conv0_kernel|gemm_split_kernel|wino_input_split_kernel')
TRAIN = 'train' in sys.argv[2:]       # every kernel of the training step; steps = loss_fwd_kernel dispatches
if TRAIN:
    FAMILY = re.compile(r'.')
rows = []
for line in open(sys.argv[1]):
    m = re.match(r'(\S+)\s+(\S+)\s+dispatches=\s*(\d+)\s+mean_per_dispatch=([0-9.eE+-]+)', line)
    if m:
        rows.append((m.group(1), m.group(2), int(m.group(3)), float(m.group(4))))
steps = max([n for k, c, n, v in rows if ('loss_fwd_kernel' if TRAIN else 'conv0_kernel') in k and c == 'FETCH_SIZE'] + [1])
tot = {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0}
launches = 0
for k, c, n, v in rows:
    if FAMILY.search(k) and c in tot:
        tot[c] += n * v * 1024.0
        if c == 'FETCH_SIZE':
            launches += n
fetch, write = tot['FETCH_SIZE'] / steps, tot['WRITE_SIZE'] / steps
dominant = None
by_grid = next((a for a in sys.argv[2:] if a.endswith('by_grid.txt')), None)
if by_grid and os.path.exists(by_grid):
    # the kernel with the largest total time in the kernel trace of the same command (tools/rocprof_summary.py by_grid): its average launch
    # duration under the schedule the command times (detect: graph replays pipelined over two streams)
    for line in open(by_grid):
        m = re.match(r'(\S+)\s+(\d+)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$', line)
        if m:
            dominant = {'kernel': m.group(1), 'wgs': int(m.group(2)), 'calls': int(m.group(3)), 'avg_us': float(m.group(5)), 'share_percent': float(m.group(8)),
                        'source': os.path.basename(by_grid)}
            break
trace = None
trace_file = next((a for a in sys.argv[2:] if a.endswith('trace.json')), None)
if trace_file and os.path.exists(trace_file):
    # per kernel-template time of the steady part of the kernel trace of the same command (tools/trace_families.py): what bench.py's roofline.frac divides by
    trace = json.load(open(trace_file))
print(json.dumps({
    'source': ('rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_profile.sh) on `tools/train_steady.py` (autotune cache pre-populated): every kernel of the training step' if TRAIN else
               'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_profile.sh) on `bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-train --no-direct-leg --no-conv3 --no-split-leg --no-multiscale --no-latency --no-resnet` (autotune cache pre-populated); conv0 + conv_fwd_dma_kernel family + split-K fix-up + all Winograd kernels'),
    'kernel_launches_profiled': launches, 'steps_profiled': steps,
    'fetch_size_bytes_raw_per_step': fetch, 'write_size_bytes_raw_per_step': write,
    'correction': 'gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md HBM section) -> doubled; WRITE_SIZE uncorrected; counters are L2 memory-side requests, Infinity-Cache hits included',
    'traffic_bytes_per_step': 2 * fetch + write,
    'dominant_trace': dominant,
    'trace': trace,
    'kernels': KERNELS,          # hash of the kernel sources the profile was taken on: bench.py reports the figure only for the same sources
}, indent=1))
