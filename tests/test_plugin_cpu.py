"""CPU-side checks of the plugin boundary: constructor signature, state_dict keys/shapes identical to
the reference plugin, ConfigChannels-driven widths, config-driven plugin resolution; construction is CPU-only."""
import configparser

import pytest
import torch

from oracle import darknet as odark
from oracle import synth
from oracle.make_golden import NARROW

import model
import model.yolo2
import utils


def config(bn=True):
    cfg = configparser.ConfigParser()
    cfg.read_dict({'batch_norm': {'enable': '1' if bn else '0'}, 'model': {'dnn': 'model.yolo2.Darknet'}})
    return cfg


def test_state_dict_keys_and_shapes_match_reference_layout():
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    dnn = model.yolo2.Darknet(model.ConfigChannels(config()), anchors, 20)
    sd = dnn.state_dict()
    ref = odark.init_state_dict(5, 20)  # keys/shapes of the reference plugin (pinned by tests/test_oracle.py)
    ours = {k: tuple(v.shape) for k, v in sd.items() if not k.endswith('num_batches_tracked')}
    assert list(ours.keys()) == list(ref.keys())
    for k, v in ref.items():
        assert ours[k] == tuple(v.shape), k
    assert sum(v.numel() for k, v in sd.items() if not k.endswith('num_batches_tracked') and 'running' not in k) == 50655389 - 0
    assert all(not p.is_cuda for p in dnn.parameters())


def test_plugin_resolution_by_dotted_path():
    cls = utils.parse_attr(config().get('model', 'dnn'))
    assert cls is model.yolo2.Darknet


def test_config_channels_follow_checkpoint():
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    sd = odark.init_state_dict(5, 20, channels=NARROW)
    dnn = model.yolo2.Darknet(model.ConfigChannels(config(), sd), anchors, 20)
    res = dnn.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    assert all(k.endswith('num_batches_tracked') for k in res.missing_keys)
    assert dnn.layers1[5].conv.weight.shape == (6, NARROW['layers1.4'], 1, 1)
    assert dnn.layers3[0].conv.weight.shape[1] == 4 * NARROW['passthrough'] + NARROW['layers2.7']


def test_no_batchnorm_variant_has_conv_bias():
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    dnn = model.yolo2.Darknet(model.ConfigChannels(config(bn=False)), anchors, 20)
    keys = list(dnn.state_dict().keys())
    assert 'layers1.0.conv.bias' in keys and not any('.bn.' in k for k in keys)


def test_output_channels_and_meshgrid():
    assert model.output_channels(5, 20) == 125 and model.output_channels(5, 80) == 425 and model.output_channels(5, 1) == 25
    g = model.meshgrid(3, 3)
    assert g.tolist()[:4] == [[0, 0], [0, 1], [0, 2], [1, 0]]


def test_forward_refuses_cpu_tensor():
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    sd = odark.init_state_dict(5, 20, channels=NARROW)
    dnn = model.yolo2.Darknet(model.ConfigChannels(config(), sd), anchors, 20).eval()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dnn(torch.zeros(1, 3, 32, 32))


@pytest.mark.parametrize('arch,width', [('resnet50', 64), ('resnet18', 64), ('resnet50', 8)])
def test_resnet_state_dict_matches_reference_layout(arch, width):
    import model.resnet
    from oracle import resnet as ores
    anchors = torch.from_numpy(synth.ANCHORS_VOC)
    ref = ores.init_state_dict(arch, 5, 80, width=width)   # keys/shapes pinned against the reference by oracle/make_golden_resnet.py
    net = getattr(model.resnet, arch)(model.ConfigChannels(config(), ref if width != 64 else None), anchors, 80)
    ours = {k: tuple(v.shape) for k, v in net.state_dict().items() if not k.endswith('num_batches_tracked')}
    assert list(ours.keys()) == list(ref.keys())
    for k, v in ref.items():
        assert ours[k] == tuple(v.shape), k
    assert utils.parse_attr('model.resnet.' + arch) is getattr(model.resnet, arch)


def test_utils_config_helpers(tmp_path):
    """ini overlay / modify / anchors / plugin resolution with the reference's file formats (config.ini, config/anchors/*.tsv)."""
    (tmp_path / 'anchors.tsv').write_text('width\theight\n1.08\t1.19\n3.42\t4.41\n')
    (tmp_path / 'cat').write_text('aeroplane\nbicycle\n')
    (tmp_path / 'a.ini').write_text('[config]\nroot = %s\n[model]\nname = model\nanchors = %s\ndnn = model.yolo2.Tiny\n[cache]\nname = cache\ncategory = %s\n[train]\nclip_ = 5\n'
                                    % (tmp_path, tmp_path / 'anchors.tsv', tmp_path / 'cat'))
    (tmp_path / 'b.ini').write_text('[model]\ndnn = model.yolo2.Darknet\n')
    cfg = configparser.ConfigParser()
    utils.load_config(cfg, [str(tmp_path / 'a.ini'), str(tmp_path / 'b.ini')])
    assert cfg.get('model', 'dnn') == 'model.yolo2.Darknet'           # later files overlay earlier ones
    a = utils.get_anchors(cfg)
    assert a.dtype.name == 'float32' and a.tolist() == [[pytest.approx(1.19), pytest.approx(1.08)], [pytest.approx(4.41), pytest.approx(3.42)]]   # (height, width)
    assert utils.get_category(cfg) == ['aeroplane', 'bicycle']
    assert utils.get_model_dir(cfg) == str(tmp_path / 'model' / 'model.yolo2.Darknet')
    utils.modify_config(cfg, 'model/dnn=model.resnet.resnet50')
    assert utils.parse_attr(cfg.get('model', 'dnn')) is __import__('model.resnet').resnet.resnet50
    utils.modify_config(cfg, 'train/clip_=')
    assert not cfg.has_option('train', 'clip_')
    utils.modify_config(cfg, 'nosuch/option=')                          # silently ignored like the reference


def test_fused_optimizer_is_torch_optim_compatible_and_has_no_cpu_fallback():
    """utils.optim.{SGD,Adam}: torch.optim.Optimizer subclasses for the ini lambda (config.ini:72); CPU tensors must raise."""
    import utils
    p = torch.nn.Parameter(torch.randn(5, 3))
    for cls, kw in ((utils.optim.SGD, dict(momentum=0.9)), (utils.optim.Adam, dict(betas=(0.9, 0.999), eps=1e-8))):
        opt = eval('lambda params, lr: utils.optim.%s(params, lr, **kw)' % cls.__name__, dict(utils=utils, kw=kw))([p], 1e-3)
        assert isinstance(opt, torch.optim.Optimizer) and opt.param_groups[0]['lr'] == 1e-3
        torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[60, 90], gamma=0.1)      # config.ini:75
        p.grad = torch.randn(5, 3)
        with pytest.raises(RuntimeError, match='GPU'):
            opt.step()
        assert set(opt.state_dict().keys()) == {'state', 'param_groups'}
    with pytest.raises(NotImplementedError):
        utils.optim.Adam([p], amsgrad=True)
