"""CPU: host-side logic of the plugin mirror -- module structure / state_dict keys against the
oracle (itself pinned to the reference), option defaults, scheduler rule, the flat Adam arena, and
the no-fallback guarantee (HIP ops refuse CPU tensors)."""
import pytest
import torch

from dfmir_amd import DfmirHipError
from dfmir_amd import networks as N
from dfmir_amd import ops
from dfmir_amd import voxelmorph as V
from dfmir_amd.optim import FlatAdam
from dfmir_amd.options import default_options
from oracle import dfmir_oracle as O


def test_generator_structure_matches_reference_indices():
    hg = N.ResnetGenerator(1, 1, 8, norm_layer=N.get_norm_layer('instance'), n_blocks=9)
    og = O.Generator(1, 1, 8, 9)
    assert list(hg.state_dict().keys()) == list(og.state_dict().keys())
    for k, v in og.state_dict().items():
        assert tuple(hg.state_dict()[k].shape) == tuple(v.shape), k
    assert len(hg.model) == 32
    assert isinstance(hg.model[0], N.ReflectionPad2d) and isinstance(hg.model[4], N.Conv2d)
    assert isinstance(hg.model[7], N.Downsample) and isinstance(hg.model[12], N.ResnetBlock)
    assert isinstance(hg.model[21], N.Upsample) and isinstance(hg.model[31], N.Tanh)
    n = sum(p.numel() for p in N.ResnetGenerator(1, 1, 64, norm_layer=N.InstanceNorm2d, n_blocks=9).parameters())
    assert n == 11365633


def test_blur_buffers_match_reference_filters():
    assert torch.allclose(N.Downsample(2).filt[0, 0], torch.tensor([[1., 2, 1], [2, 4, 2], [1, 2, 1]]) / 16)
    assert torch.allclose(N.Upsample(2).filt[0, 0].sum(), torch.tensor(4.0))


def test_vxm_structure():
    hv = V.VxmDense((64, 64), O.PLUGIN_UNET_FEATURES, int_steps=7, bidir=True)
    ov = O.VxmDense((64, 64), O.PLUGIN_UNET_FEATURES, 7, True)
    hk = [k for k in hv.state_dict().keys() if not k.endswith('.grid')]
    assert hk == list(ov.state_dict().keys())
    assert 'transformer.grid' in hv.state_dict() and 'integrate.transformer.grid' in hv.state_dict()
    assert sum(p.numel() for p in hv.parameters()) == 356258
    h3 = V.VxmDense((32, 32, 32), None, int_steps=7, bidir=True)
    assert sum(p.numel() for p in h3.parameters()) == 301411
    assert float(hv.flow.weight.abs().max()) < 1e-3 and float(hv.flow.bias.abs().max()) == 0.0
    assert hv.config['int_steps'] == 7 and hv.config['bidir'] is True
    assert hv.transformer.grid.shape == (1, 2, 64, 64) and float(hv.transformer.grid[0, 0, 5, 0]) == 5.0


def test_patch_sampler_keys():
    f = N.PatchSampleF(use_mlp=True, nc=32)
    f.mlp_init = False
    feats = [torch.zeros(1, 1, 4, 4), torch.zeros(1, 16, 4, 4)]
    f.create_mlp(feats)
    o = O.PatchSampler(32, True)
    o.create_mlp(feats)
    assert list(f.state_dict().keys()) == list(o.state_dict().keys())
    assert float(f.mlp_0[0].bias.abs().max()) == 0.0


def test_model_class_discovery_and_options():
    import dfmir_amd.registration_model as rm
    names = [n for n, c in vars(rm).items() if n.lower() == 'registrationmodel']
    assert names == ['REGISTRATIONModel']
    opt = default_options()
    assert opt.nce_idt is True and opt.lambda_NCE == 0.25 and opt.nce_layers == '0,4,8,12,16'
    assert opt.lr == 2e-4 and opt.beta1 == 0.5 and opt.num_patches == 256 and opt.nce_T == 0.07


def test_scheduler_linear_rule():
    opt = default_options(n_epochs=2, n_epochs_decay=2, epoch_count=1)
    p = [torch.nn.Parameter(torch.zeros(3))]
    o = torch.optim.SGD(p, lr=1.0)
    s = N.get_scheduler(o, opt)
    lrs = []
    for _ in range(4):
        lrs.append(o.param_groups[0]['lr'])
        o.step(); s.step()
    assert lrs == [1.0, 1.0, pytest.approx(2 / 3), pytest.approx(1 / 3)]


def test_ops_refuse_cpu_tensors():
    x = torch.zeros(1, 2, 4, 4)
    with pytest.raises(DfmirHipError):
        ops.instance_norm(x)
    with pytest.raises(DfmirHipError):
        ops.warp(x, torch.zeros(1, 2, 4, 4))
    with pytest.raises(DfmirHipError):
        ops.conv(x, torch.zeros(3, 2, 3, 3), None, None, 1, 1, 0, 0, 0.0)


def test_flat_adam_arena_aliases_parameters():
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
    ps = list(net.parameters())
    before = [p.detach().clone() for p in ps]
    o = FlatAdam(ps, lr=1e-3, betas=(0.5, 0.999))
    assert o.flat_p.numel() == sum(p.numel() for p in ps)
    off = 0
    for p, b in zip(ps, before):
        assert torch.equal(p.detach(), b)
        assert p.data_ptr() == o.flat_p.data_ptr() + 4 * off
        assert p.grad.data_ptr() == o.flat_g.data_ptr() + 4 * off
        off += p.numel()
    net(torch.ones(1, 3)).sum().backward()
    assert float(o.flat_g.abs().sum()) > 0      # autograd accumulated straight into the arena
    o.zero_grad()
    assert float(o.flat_g.abs().sum()) == 0
    with pytest.raises(DfmirHipError):            # the fused Adam kernel is HIP-only: no CPU fallback
        o.step()


def test_unet_channel_plan_matches_reference_layouts():
    """(cin, cout) of every U-Net conv for the two feature lists on the path (reference networks.py:60-86; SURVEY
    Appendix A): the plugin's 6-level 2-D list and the stock 4-level default."""
    down, up, extras = V.unet_channel_plan([16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16])
    assert down == [(2, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 64)]
    assert up == [(64, 64), (128, 64), (128, 64), (96, 32), (64, 32), (48, 32)]
    assert extras == [(34, 16)]
    enc, dec = V.default_unet_features()
    down, up, extras = V.unet_channel_plan(enc, dec)
    assert down == [(2, 16), (16, 32), (32, 32), (32, 32)]
    assert up == [(32, 32), (64, 32), (64, 32), (48, 32)]
    assert extras == [(34, 32), (32, 16), (16, 16)]


def test_base_model_surface_and_graph_defaults():
    """The plugin surface train.py / test.py drive and the build-defined options' defaults."""
    from dfmir_amd.base_model import BaseModel
    from dfmir_amd.options import default_options
    for name in ("setup", "parallelize", "data_dependent_initialize", "set_input", "forward", "optimize_parameters", "eval",
                 "test", "compute_visuals", "get_image_paths", "update_learning_rate", "get_current_visuals",
                 "get_current_losses", "save_networks", "load_networks", "print_networks", "sync_gradients"):
        assert callable(getattr(BaseModel, name)), name
    opt = default_options()
    assert opt.capture_step is False and opt.dvf_image == 'synthetic' and opt.reuse_key_features and opt.batch_query_passes


def test_distributed_helpers_single_process():
    from dfmir_amd import distributed as D
    assert not D.is_distributed() and D.world_size() == 1
    assert D.allreduce_arenas([torch.zeros(3)]) == [] and D.allreduce_arenas([torch.zeros(3)], async_op=True) == []
    assert D.allreduce_max(1.5, "cpu") == 1.5
    D.barrier()


def test_checkpoint_keys_equal_the_reference_written_files():
    """Row N2 pinned against the REFERENCE, not the oracle: tests/golden/checkpoint_keys.json lists, for the three .pth
    files the reference's own save_networks wrote (models/base_model.py:164-180; generated by
    tests/golden/make_checkpoint_keys.py in the build container), every key in order with shape and dtype.  The HIP-side
    modules -- which need no device to be constructed -- must produce the same list, so that reference-trained files load
    with strict=True and files written here load into the reference."""
    import json
    import os
    import torch
    from dfmir_amd import networks as N
    from dfmir_amd.voxelmorph import VxmDense
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_keys.json")))
    sig = lambda net: [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
    for ngf in (8, 64):
        want = fx["size64_ngf%d" % ngf]
        g = N.define_G(1, 1, ngf, 'resnet_9blocks', 'instance', False, 'xavier', 0.02, False, False, [], None)
        assert sig(g) == want["G"]
        r = VxmDense((64, 64), [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]], int_steps=7, bidir=True)
        assert sig(r) == want["R"]                         # incl. transformer.grid / integrate.transformer.grid buffers
        import argparse
        f = N.define_F(1, 'mlp_sample', 'instance', False, 'xavier', 0.02, False, [], argparse.Namespace(netF_nc=256))
        # the MLPs are created from the five tapped features' channel counts (networks.py:587-595): 1, 2*ngf, 4*ngf x 3
        f.create_mlp([torch.zeros(1, c, 1, 1) for c in (1, 2 * ngf, 4 * ngf, 4 * ngf, 4 * ngf)])
        assert sig(f) == want["F"]
