"""Host logic of the binding that needs no GPU: the per-shape plan LRU (multi-scale training, utils/data.py:135-141), the measured-choice
table's export / import (what data-parallel ranks exchange, train.py:427-433) and the static algorithm preferences (Y2_AUTOTUNE=0)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo2-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def test_plan_cache_is_an_lru_bounded_by_entries_and_bytes():
    import _hip
    c = _hip.PlanCache(entries=3, gbytes=1.0)
    for i in range(3):
        c.put(('shape', i), {'id': i}, 100 << 20)
    assert c.get(('shape', 0))['id'] == 0                      # touch 0: 1 is now the oldest
    c.put(('shape', 3), {'id': 3}, 100 << 20)
    assert c.get(('shape', 1)) is None and c.get(('shape', 0)) is not None and c.get(('shape', 3)) is not None
    c.put(('shape', 4), {'id': 4}, 900 << 20)                  # the byte bound evicts until the newest fits with what is left
    assert c.get(('shape', 4)) is not None and sum(p['nbytes'] for p in c.d.values()) <= (1 << 30)
    c.put(('huge',), {'id': 9}, 5 << 30)                       # a plan larger than the bound still stays (the cache never drops its only entry)
    assert c.latest()['id'] == 9 and len(c.d) == 1
    assert c.hits == 4 and c.misses == 1


def test_tune_table_round_trip_keeps_choices_and_device_placeholders(monkeypatch):
    import _hip
    monkeypatch.setattr(_hip, '_TUNE', {})
    monkeypatch.setattr(_hip, 'TUNE_CACHE', None)
    key_conv = (64, 13, 13, 512, 512, 1024, 3, True, False, False, 0, 0, 0, False, 0, 0, 0, 'cuda:0', True, True)
    key_wgrad = ('wgrad', 64, 13, 13, 512, 512, 1024, 1024, True, 'cuda:0')
    _hip._TUNE[key_conv] = [2, 3]
    _hip._TUNE[key_wgrad] = 2
    blob = _hip.export_tune()
    monkeypatch.setattr(_hip, '_TUNE', {})
    epoch = _hip.tune_epoch()
    _hip.import_tune(blob, 'cuda:3')                           # another rank's device name
    assert len(_hip._TUNE) == 2 and _hip.tune_epoch() != epoch     # plans built on the old choices are invalidated
    got = {k: v for k, v in _hip._TUNE.items()}
    assert any(k[0] == 'wgrad' and k[-1] == 'cuda:3' and v == 2 for k, v in got.items())
    assert any(k[0] == 64 and 'cuda:3' in k and list(v) == [2, 3] for k, v in got.items())


def test_tune_table_merge_keeps_local_entries_and_moves_the_epoch_only_on_change(monkeypatch):
    import _hip
    monkeypatch.setattr(_hip, '_TUNE', {('a', 'cuda:1'): [1, 5], ('mine', 'cuda:1'): 2})
    epoch = _hip.tune_epoch()
    _hip.import_tune([(('a', '@dev'), (1, 5))], 'cuda:1', merge=True)          # nothing new: plans stay valid
    assert _hip.tune_epoch() == epoch and len(_hip._TUNE) == 2
    _hip.import_tune([(('a', '@dev'), [2, 0]), (('b', '@dev'), 1)], 'cuda:1', merge=True)
    assert _hip.tune_epoch() == epoch + 1
    assert _hip._TUNE == {('a', 'cuda:1'): [2, 0], ('mine', 'cuda:1'): 2, ('b', 'cuda:1'): 1}


@pytest.mark.parametrize('cin,hw,want', [(32, 208, 0), (64, 104, 0), (128, 52, 1), (256, 26, 1), (512, 13, 2), (1024, 19, 2), (1280, 13, 2)])
def test_static_weight_gradient_preferences(monkeypatch, cin, hw, want):
    """Y2_AUTOTUNE=0: the choices the measurements converge to - direct kernel below 128 input channels, the 2x2-tile Winograd reduction above,
    its 4x4-tile form on the 13x13 / 19x19 layers; the deterministic mode never takes the 4x4 form (the library refuses it there)."""
    import _hip
    monkeypatch.setattr(_hip, 'AUTOTUNE', False)
    monkeypatch.setattr(_hip, 'DETERMINISTIC', False)
    monkeypatch.setattr(_hip, 'WINOGRAD', True)
    monkeypatch.setattr(_hip, 'WGRAD_F34', True)
    assert _hip.wgrad_choice(64, hw, hw, cin, cin, 2 * cin, 2 * cin, 3, True, 'cuda:0') == want
    assert _hip.wgrad_choice(64, hw, hw, cin, cin, 2 * cin, 2 * cin, 1, True, 'cuda:0') == 0          # 1x1 layers: the direct kernel
    monkeypatch.setattr(_hip, 'DETERMINISTIC', True)
    assert _hip.wgrad_choice(64, hw, hw, cin, cin, 2 * cin, 2 * cin, 3, True, 'cuda:0') == (1 if cin >= 128 else 0)


def test_default_tune_table_is_adopted_only_for_the_kernels_it_was_measured_on(monkeypatch, tmp_path):
    """The committed table (yolo2-pytorch_amd/tune/default_gfx950.json) carries the hash of the kernel sources: entries are adopted for the
    device they are asked for, never over a choice this process already holds, and not at all when the sources have changed."""
    import json

    import _hip
    monkeypatch.setattr(_hip, '_TUNE', {('mine', 'cuda:2'): [1, 5]})
    monkeypatch.setattr(_hip, '_DEFAULTS_SEEN', {})
    monkeypatch.setattr(_hip, 'TUNE_DEFAULTS', True)
    h = _hip.kernel_hash()
    assert h is not None and len(h) == 16 and h == _hip.kernel_hash()
    good = tmp_path / 'good.json'
    json.dump({'kernels': h, 'entries': [[['mine', '@dev'], [0, 0]], [['wgrad', 64, 13, 13, 512, 512, 1024, 1024, True, '@dev'], 2], [[32, 13, 13, True, '@dev', False], [1, 5]]]}, open(good, 'w'))
    assert _hip.load_tune_defaults('cuda:2', str(good)) == 2
    assert _hip._TUNE[('mine', 'cuda:2')] == [1, 5]                       # what the process held stays
    assert _hip._TUNE[('wgrad', 64, 13, 13, 512, 512, 1024, 1024, True, 'cuda:2')] == 2 and _hip._TUNE[(32, 13, 13, True, 'cuda:2', False)] == [1, 5]
    stale = tmp_path / 'stale.json'
    json.dump({'kernels': '0' * 16, 'entries': [[['other', '@dev'], [2, 0]]]}, open(stale, 'w'))
    assert _hip.load_tune_defaults('cuda:2', str(stale)) == 0 and ('other', 'cuda:2') not in _hip._TUNE
    # the committed file, when there is one, parses and names a hash
    if os.path.exists(_hip.DEFAULTS_PATH):
        d = json.load(open(_hip.DEFAULTS_PATH))
        assert len(d['kernels']) == 16 and all(len(e) == 2 for e in d['entries'])


def test_step_runner_keeps_one_plan_per_shape_and_degrades_per_shape(monkeypatch):
    """train.StepRunner's bookkeeping without a GPU (the plans are stand-ins): a batch with fewer boxes runs in the plan captured for more, a batch with
    more supersedes it, warm-up is counted per input shape, a capture that runs out of memory drops the OTHER captured steps and leaves that shape on
    eager launches, any other capture failure stops capturing - and no failure ever stops the step from running."""
    import torch

    import train
    from model import train_graph

    made = []

    class FakePlan(object):
        WARM = 3

        def __init__(self, inference, anchors, hparam, threshold, dp=None, pool=None, shared=None, arena=None, scope=None):
            self.ops, self.capture_error, self.calls, self.static, self.params, self.last_grads = None, None, 0, None, [], {}
            self.used_last = None
            self.fail = None
            made.append(self)

        def _alloc(self, data, npad):
            self.static = {'npad': npad}

        def valid(self):
            return True

        def run(self, data, capture=True):
            self.calls += 1
            if self.ops is None and capture and self.calls > self.WARM and self.capture_error is None:
                if self.fail is not None:
                    self.capture_error = self.fail
                else:
                    self.ops = [('graph', None)]
            return {'ran': self.calls}
    monkeypatch.setattr(train_graph, 'StepPlan', FakePlan)
    monkeypatch.setattr(torch.cuda, 'graph_pool_handle', lambda: object())
    monkeypatch.setattr(torch.cuda, 'empty_cache', lambda: None)
    monkeypatch.setattr(train.StepRunner, 'eligible', lambda self, data: True)
    r = train.StepRunner(object(), None, object(), {'foreground': 5.0}, 0.6)

    def batch(S, n):
        return {'tensor': torch.zeros(2, 3, S, S), 'yx_min': torch.zeros(2, n, 2), 'yx_max': torch.zeros(2, n, 2), 'cls': torch.zeros(2, n, dtype=torch.int64)}
    for _ in range(5):
        assert r.step(batch(96, 6))['ran']
    assert len(r.plans) == 1 and r.captures == 1 and made[-1].static['npad'] == 16
    assert r.step(batch(96, 3)) and len(r.plans) == 1 and len(made) == 1                 # fewer boxes: the same plan
    assert r.step(batch(96, 40)) and len(r.plans) == 1 and len(made) == 2               # more boxes: a 64-row plan supersedes it ...
    assert made[-1].static['npad'] == 64 and r.captures == 2                             # ... captured at its first call (the shape is warm)
    assert r.step(batch(96, 6)) and len(made) == 2                                       # and serves the small batches from now on
    for _ in range(4):
        r.step(batch(128, 6))
    assert len(r.plans) == 2 and r.captures == 3
    # out of memory while capturing a third shape: the step runs, the other captured steps are dropped, the shape stays eager
    for i in range(3):
        r.step(batch(160, 6))
    made[-1].fail = torch.cuda.OutOfMemoryError('HIP out of memory')
    assert r.step(batch(160, 6))['ran'] == 4
    assert len(r.plans) == 1 and not r.broken and len(r.eager_only) == 1
    assert r.step(batch(160, 6))['ran'] == 5 and made[-1].ops is None                   # no second attempt for that shape
    for _ in range(4):
        r.step(batch(96, 6))                                                             # the others capture again, in a fresh pool
    assert r.captures == 4
    # any other failure: no more captures for this model, steps keep running
    for i in range(3):
        r.step(batch(192, 6))
    made[-1].fail = RuntimeError('boom')
    assert r.step(batch(192, 6))['ran'] == 4 and r.broken
    for _ in range(5):
        assert r.step(batch(224, 6))
    assert made[-1].ops is None and r.captures == 4
