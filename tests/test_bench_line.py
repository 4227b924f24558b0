"""The driver parses the LAST stdout line of bench.py and keeps a bounded tail: the line must stay small (round 4's 30 KB line came back
as `parsed: null`), carry every contract key once, and its static traffic figure must be the sum of the per-kernel table of the same PMC passes."""
import glob
import json
import os
import re
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def _worst_case():
    """A result with MORE scalars than any real run produces (every leg present, long kernel names, long strings)."""
    head = {'metric': 'images/sec (416x416) train+detect, Darknet-19 YOLOv2: value = TRAIN step, batch 64 per GPU (configs[2]) at every N; detect (configs[1]) = roofline.detect_images_per_sec', 'value': 2051.87,
            'unit': 'images/sec', 'n_gpus': 1, 'ranks_seen': 1, 'steps': 20, 'warmup': 5, 'ms_per_step': 31.3526, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'headline': 'train',
            'config': {'workload': 'Darknet-19 YOLOv2 20-class train 416x416 batch-64/GPU: fwd + region loss + bwd + SGD (BASELINE configs[2])', 'classes': 20, 'global_batch': 64,
                       'parallelism': 'single GPU', 'weights': 'random-init seed 0 (bench_data.randomize)'}}
    roof = {'bound': 'mfma', 'peak': 157.3, 'unit': 'TFLOP/s', 'what': 'x' * 400, 'kernel': 'conv_fwd_dma_kernel', 'achieved': 111.6, 'frac': 0.7096, 'frac_uncontended': 0.8075,
            'avg_launch_us': 373.3, 'avg_launch_us_uncontended': 327.9, 'kernel_share_of_step': 0.41, 'launches_per_step': 29.0, 'traffic': 8.49e9, 'traffic_source': 'static: ' + 'p' * 300,
            'frac_source': 'f' * 300, 'step_frac_executed': 0.5179, 'step_frac_direct_equiv': 1.14, 'kernel_ms_per_step': 4.9, 'top_kernels': [{'kernel': 'k%d' % i, 'frac': 0.5} for i in range(40)],
            'conv_chain': {'a': 1}, 'families': {'k': {'a': 1}}, 'definition': 'd' * 500}
    extra = {}
    for i in range(12):
        extra['frac_some_quite_long_kernel_name_%d_implicit' % i] = 0.61234
    for k in ('detect_images_per_sec', 'detect_ms_per_step', 'detect_streams', 'detect_serial_images_per_sec', 'detect_serial_ms_per_step', 'conv_chain_ms_per_step', 'conv_chain_frac',
              'detect_direct_only_frac', 'detect_to_host_images_per_sec', 'detect_dominant_frac', 'detect_dominant_frac_uncontended', 'detect_step_frac_executed',
              'detect_step_frac_direct_equiv', 'detect_traffic', 'train_images_per_sec', 'train_ms_per_step', 'train_host_issue_ms_per_step', 'train_dp_exposed_comm_ms_per_step', 'train_ms_per_step_contended',
              'train_single_gpu_images_per_sec', 'train_traffic_bytes_per_step', 'train_mfma_frac', 'train_mfma_ms_per_step', 'train_kernel_ms_sum_single_stream', 'train_dominant_frac',
              'train_dominant_avg_launch_us', 'conv3x3_b64_mfma_util', 'conv3x3_b64_direct_only_util', 'latency_b1_ms', 'latency_b1_launches', 'latency_b1_executed_mfma_frac',
              'latency_b1_direct_equiv_frac', 'latency_b8_ms', 'latency_b8_launches', 'latency_b8_executed_mfma_frac', 'latency_b8_direct_equiv_frac', 'latency_b1_weight_tbs',
              'resnet50_608_detect_images_per_sec', 'resnet50_608_train_images_per_sec', 'resnet50_608_train_ms_per_step', 'resnet50_608_train_direct_equiv_frac',
              'resnet50_608_train_mfma_frac', 'resnet50_608_train_mfma_ms_per_step', 'resnet50_608_train_kernel_ms_sum', 'multiscale_images_per_sec', 'multiscale_ms_per_step_mean',
              'multiscale_switch_cost_ms_max', 'multiscale_first_visit_ms_mean', 'multiscale_first_visit_ms_max', 'split_bf16x6_detect_images_per_sec', 'split_f16x3_detect_images_per_sec'):
        extra[k] = 98765.4321
    extra['train_dominant_kernel'] = 'conv_wgrad_kernel[grouped]'
    cb = {'value': 16.3, 'unit': 'images/sec', 'cores': 32, 'kind': 'port', 'cpu_model': 'AMD EPYC 9575F 64-Core Processor', 'cpu_count': 256, 'sample': 's' * 400,
          'b1_ms_per_image': 77.1, 'b1_images_per_sec': 13.0, 'b1_sample': 'b' * 200, 'train_b8_images_per_sec': 8.1, 'train_b8_s_per_step': 0.98, 'train_b8_sample': 't' * 200,
          'nms_n200_ms': 1.41, 'nms_sample': 'n' * 200}
    return head, roof, extra, cb


def test_final_line_is_small_flat_and_complete():
    import bench
    head, roof, extra, cb = _worst_case()
    line = bench.compact_line(head, roof, extra, cb, 'gpurun_out/bench_full.json')
    assert '\n' not in line and len(line) <= bench.LINE_LIMIT <= 6000, len(line)
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_uncontended', 'step_frac_executed', 'step_frac_direct_equiv', 'frac_source'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert all(not isinstance(v, (dict, list)) for v in r.values())           # scalars only: tables live in the file `tables` names
    for k in ('train_images_per_sec', 'train_ms_per_step', 'train_traffic_bytes_per_step', 'conv3x3_b64_mfma_util', 'latency_b1_ms', 'multiscale_first_visit_ms_mean',
 'resnet50_608_train_direct_equiv_frac', 'detect_serial_ms_per_step', 'detect_images_per_sec', 'detect_step_frac_executed', 'detect_to_host_images_per_sec'):
        assert k in r, k                                                     # the figures a reviewer looks for survive the trimming
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert 'summary' not in d                                                # nothing is printed twice
    # a result far beyond any real one still fits: the optional scalars go first, the contract keys never
    for i in range(200):
        extra['zz_filler_%03d' % i] = 1.2345678
    line = bench.compact_line(head, roof, extra, cb, 'x.json')
    assert len(line) <= bench.LINE_LIMIT
    r = json.loads(line)['roofline']
    assert all(k in r for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_uncontended', 'step_frac_executed', 'step_frac_direct_equiv'))


def test_roofline_frac_follows_the_committed_trace_only_when_it_describes_the_same_launches(tmp_path, monkeypatch):
    """`roofline.frac` is the figure profiles/ reproduces: this run's executed FLOPs per step of the dominant kernel TEMPLATE over the committed trace's time per
    step of that template - taken only for the same kernel sources and the same launches per step; otherwise the event-hook figure, labelled as such."""
    import bench
    import _hip
    table = {'conv_fwd_dma_kernel[grouped]': dict(launches=14.0, ms=6.0, flops=14 * 55e9), 'conv_fwd_dma_kernel[persistent]': dict(launches=15.0, ms=1.5, flops=15 * 10e9),
             'conv_wgrad_kernel[grouped]': dict(launches=13.0, ms=5.3, flops=13 * 50e9), 'bn_act_bwd_kernel': dict(launches=45.0, ms=3.0, flops=0.0)}
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    rec = {'kernels': _hip.kernel_hash(), 'trace': {'steps': 10, 'families': {'conv_fwd_dma_kernel': {'calls': 290, 'total_us': 82000.0}}}}
    (prof / 'r99_train_b64_traffic.json').write_text(json.dumps(rec))
    r = bench.roofline_from(table, 'x', trace_tag='train_b64')
    flops = 14 * 55e9 + 15 * 10e9
    assert r['kernel'] == 'conv_fwd_dma_kernel' and r['launches_per_step'] == 29.0
    assert abs(r['frac'] - flops / 8.2e-3 / 1e12 / 157.3) < 1e-3 and 'r99_train_b64_traffic.json' in r['frac_source']
    assert abs(r['frac_uncontended'] - flops / 7.5e-3 / 1e12 / 157.3) < 1e-3 and r['frac'] < r['frac_uncontended']
    # another launch count per step (a changed algorithm table): the trace no longer describes this run
    rec['trace']['families']['conv_fwd_dma_kernel']['calls'] = 250
    (prof / 'r99_train_b64_traffic.json').write_text(json.dumps(rec))
    r = bench.roofline_from(table, 'x', trace_tag='train_b64')
    assert r['frac'] == r['frac_uncontended'] and 'event hooks' in r['frac_source'] and '25.00 x per step in the trace' in r['frac_source']
    # other kernel sources
    rec['kernels'] = 'deadbeef'
    rec['trace']['families']['conv_fwd_dma_kernel']['calls'] = 290
    (prof / 'r99_train_b64_traffic.json').write_text(json.dumps(rec))
    r = bench.roofline_from(table, 'x', trace_tag='train_b64')
    assert r['frac'] == r['frac_uncontended'] and 'other kernel sources' in r['frac_source']


def test_tables_go_to_a_file(tmp_path):
    import bench
    p = str(tmp_path / 'sub' / 'full.json')
    at = bench.write_tables({'a': [1, 2, 3]}, p)
    assert at == p and json.load(open(p)) == {'a': [1, 2, 3]}


def test_static_traffic_equals_the_per_kernel_table():
    """tools/traffic_from_pmc.py (the figure bench.py prints) against tools/traffic_by_kernel.py (the table a reader sums) on the committed PMC summaries:
    round 4 printed 5.99 GB for 8.49 GB because the kernel-name filter missed wino_fused3_kernel."""
    pmcs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0[4-9]_*_pmc_summary.txt')))
    assert pmcs
    for pmc in pmcs:
        tag = os.path.basename(pmc).replace('_pmc_summary.txt', '')
        train = 'train' in tag
        js = os.path.join(ROOT, 'profiles', tag + '_traffic.json')
        assert os.path.exists(js), js
        want = json.load(open(js))
        got = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', 'traffic_from_pmc.py'), pmc] + (['train'] if train else [])))
        assert abs(got['traffic_bytes_per_step'] - want['traffic_bytes_per_step']) <= 1e-6 * want['traffic_bytes_per_step'], tag
        tab = subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', 'traffic_by_kernel.py'), pmc, str(want['steps_profiled'])]).decode()
        total_gb = float(re.search(r'total ([0-9.]+) GB per step', tab).group(1))
        # the table covers EVERY kernel of the run (decode / NMS included: a few MB), the figure the conv chain - they agree to the table's rounding
        assert abs(total_gb - want['traffic_bytes_per_step'] / 1e9) <= 0.06 + 0.005 * total_gb, (tag, total_gb, want['traffic_bytes_per_step'])


def test_trace_families_names_kernel_templates_like_the_hook_table(tmp_path):
    """tools/trace_families.py groups the dispatches of a rocprofv3 trace by kernel TEMPLATE (the name bench.roofline_from derives from the library's hook names):
    a synthetic rocpd database with two steps of two kernels, one warm-up dispatch in front of the window."""
    import sqlite3
    db = str(tmp_path / 't.db')
    con = sqlite3.connect(db)
    con.execute('create table rocpd_info_kernel_symbol_x (id integer, kernel_name text)')
    con.execute('create table rocpd_kernel_dispatch_x (kernel_id integer, start integer, end integer)')
    names = {1: '_ZN12_GLOBAL__N_119conv_fwd_dma_kernelILi64ELi128ELi2ELb0ELb0ELi2ELb0ELi256ELb0EEEvNS_8ConvArgsE.kd', 2: '_ZN12_GLOBAL__N_115loss_fwd_kernelENS_8LossArgsEPdS1_.kd',
             3: '_Z16zero_fill_kernelP15HIP_vector_typeIfLj4EEm.kd'}
    for i, n in names.items():
        con.execute('insert into rocpd_info_kernel_symbol_x values (?, ?)', (i, n))
    t = 0
    rows = [(3, 5000)]                                                   # a warm-up dispatch
    for step in range(3):                                               # three step markers = two whole steps in the window
        rows += [(2, 1000), (1, 300000), (1, 200000)]
    for kid, dur in rows:
        con.execute('insert into rocpd_kernel_dispatch_x values (?, ?, ?)', (kid, t, t + dur))
        t += dur + 100
    con.commit()
    con.close()
    out = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', 'trace_families.py'), db, 'loss_fwd_kernel', '0.0']))
    assert out['steps'] == 2
    f = out['families']
    assert f['conv_fwd_dma_kernel'] == {'calls': 4, 'total_us': 1000.0} and f['loss_fwd_kernel']['calls'] == 2 and 'zero_fill_kernel' not in f
    import bench
    assert bench.family('conv_fwd_dma_kernel[grouped]') == 'conv_fwd_dma_kernel'
