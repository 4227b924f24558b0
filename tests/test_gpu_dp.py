"""RCCL path of the data-parallel wrapper on the single visible MI355X (world_size 1, backend nccl): the Darknet
backward announces gradients layer by layer (grad_ready_hook), buckets are all-reduced through RCCL and written back;
with one rank the result must equal the unwrapped gradients (up to the run-to-run summation-order noise of the split-K
weight-gradient atomics)."""
import configparser
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import darknet as odark
from oracle import loss as oloss
from oracle import synth
from oracle.make_golden import NARROW

pytestmark = pytest.mark.gpu


def build(sd):
    import model
    import model.yolo2
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1'}})
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    dnn = model.yolo2.Darknet(model.ConfigChannels(cfg, sd), anchors, 20)
    dnn.load_state_dict(sd, strict=False)
    return model.Inference(cfg, dnn, anchors).cuda().train(), anchors


def test_dp_wrapper_rccl_world1_equals_plain_backward():
    import model
    import train
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    try:
        widths = dict(NARROW)
        widths['layers1.5'] = 8
        sd = odark.init_state_dict(5, 20, seed=0, channels=widths, head_scale=1 / 8.0)
        x = synth.images(2, 96, seed=1).cuda()
        data = synth.norm_data(synth.labels(2, 96, 20, seed=2), 96, 96, 3, 3)
        grads = []
        for wrap in (False, True):
            inf, anchors = build(sd)
            m = train.DataParallelRCCL(inf, bucket_bytes=4096) if wrap else inf
            if wrap:
                assert inf.dnn.grad_ready_hook is not None
            pred = model._inference(m, x)
            loss, _ = model.loss(anchors, data, pred, 0.6)
            sum(loss[k] * oloss.HPARAM[k] for k in loss).backward()
            torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in inf.named_parameters()})
        for k in grads[0]:
            a, b = grads[0][k], grads[1][k]
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-9, k
    finally:
        dist.destroy_process_group()
