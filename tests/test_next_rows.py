"""The rows either side of the hot path: data pipeline and logging on CPU;
checkpoint interchange, inference path and the training driver on the GPU."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from dfmir_amd.options import default_options
from tests.golden import common as C


def _make_folders(root, n=5, size=80):
    rng = np.random.RandomState(0)
    for d in ("trainA", "trainB"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
        for i in range(n):
            a = (rng.rand(size, size + 8) * 255).astype(np.uint8)
            Image.fromarray(a, mode="L").save(os.path.join(root, d, "%s_%02d.png" % (d, i)))


def test_dataset_pipeline(tmp_path):
    from dfmir_amd.data import create_dataset, make_dataset, slice_transform
    root = str(tmp_path)
    _make_folders(root)
    opt = default_options(dataroot=root, phase="train", batch_size=2, load_size=72, crop_size=64, no_flip=True,
                          serial_batches=True, num_threads=0, max_dataset_size=float("inf"))
    assert len(make_dataset(os.path.join(root, "trainA"))) == 5
    ds = create_dataset(opt)
    assert len(ds) == 5
    batches = list(ds)
    assert len(batches) == 2                       # drop_last in training (data/__init__.py:79)
    b = batches[0]
    assert b["A"].shape == (2, 1, 64, 64) and b["B"].dtype == torch.float32
    assert float(b["A"].min()) >= -1.0 and float(b["A"].max()) <= 1.0
    assert b["A_paths"][0].endswith("trainA_00.png") and b["B_paths"][1].endswith("trainB_01.png")
    # transform = Grayscale -> bicubic resize -> crop -> (x-0.5)/0.5, checked against the same PIL calls
    img = Image.open(os.path.join(root, "trainA", "trainA_00.png"))
    o2 = default_options(load_size=64, crop_size=64, no_flip=True)
    t = slice_transform(img, o2)
    ref = np.asarray(img.convert("L").resize((64, 64), Image.BICUBIC), dtype=np.float32) / 255.0
    assert torch.allclose(t[0], torch.from_numpy((ref - 0.5) / 0.5))
    # sizes not divisible by 4 are rounded like __make_power_2(base=4)
    o3 = default_options(load_size=70, crop_size=70, no_flip=True)
    assert slice_transform(img, o3).shape == (1, 72, 72)


def test_dataset_pipeline_vs_reference_fixture(tmp_path, golden):
    """Row N3 against the REFERENCE's own UnalignedDataset + get_transform (tests/golden/make_golden.py ran them over
    the same seeded PNG folders): same files paired, same pixels after grayscale / bicubic resize / random crop /
    x4 rounding / flips / normalisation -- bit for bit (both sides are the same PIL calls, drawn from the same RNGs)."""
    import random
    from tests.golden import common as C
    from dfmir_amd.data import UnalignedDataset
    g = golden("dataset.npz")
    root = str(tmp_path)
    C.write_slice_folders(root)
    for tag, no_flip in (("flip", False), ("noflip", True)):
        opt = default_options(dataroot=root, phase="train", batch_size=1, load_size=30, crop_size=24, no_flip=no_flip,
                              serial_batches=False, num_threads=0, max_dataset_size=float("inf"),
                              preprocess="resize_and_crop")
        ds = UnalignedDataset(opt)
        assert len(ds) == int(g["len_" + tag])
        for idx in range(4):
            random.seed(500 + idx)
            torch.manual_seed(700 + idx)
            item = ds[idx]
            names = os.path.basename(item["A_paths"]) + "|" + os.path.basename(item["B_paths"])
            assert names == str(g["names_%s_%d" % (tag, idx)])
            assert np.array_equal(item["A"].numpy(), g["A_%s_%d" % (tag, idx)]), (tag, idx, "A")
            assert np.array_equal(item["B"].numpy(), g["B_%s_%d" % (tag, idx)]), (tag, idx, "B")


def test_visualizer_log_format(tmp_path):
    from collections import OrderedDict
    from dfmir_amd.visualizer import Visualizer
    opt = default_options(checkpoints_dir=str(tmp_path), name="exp")
    v = Visualizer(opt)
    msg = v.print_current_losses(3, 200, OrderedDict([("G", 1.23456), ("R", 0.5)]), 0.0123, 0.001)
    assert msg == "(epoch: 3, iters: 200, time: 0.012, data: 0.001) G: 1.235 R: 0.500 "
    assert open(os.path.join(str(tmp_path), "exp", "loss_log.txt")).read().strip().endswith(msg.strip())


def test_train_option_parser():
    from dfmir_amd.train import parse
    opt = parse(["--dataroot", "/x", "--batch_size", "16", "--ngf", "32", "--nce_idt", "false"])
    assert opt.batch_size == 16 and opt.ngf == 32 and opt.nce_idt is False and opt.lambda_NCE == 0.25
    assert opt.isTrain and opt.nce_layers == "0,4,8,12,16"


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_checkpoint_interchange(tmp_path):
    """<epoch>_net_{G,F,R}.pth written by the HIP model load into the oracle modules and back
    (base_model.py:164-224; key names = the reference's)."""
    from oracle import dfmir_oracle as O
    from tests.test_gpu_models import _hip_model_from_oracle, _load
    torch.manual_seed(5)
    st = O.RegistrationStep(64, 1, ngf=8)
    model, opt = _hip_model_from_oracle(st, 64, 1, 8)
    opt.checkpoints_dir, opt.name = str(tmp_path), "ck"
    model.save_dir = os.path.join(str(tmp_path), "ck")
    A0, B0 = C.image_pair(6, 1, 64, 64)
    model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""], "B_paths": [""]})
    model.save_networks("latest")
    for nm in ("G", "F", "R"):
        assert os.path.exists(os.path.join(model.save_dir, "latest_net_%s.pth" % nm))
    sdG = torch.load(os.path.join(model.save_dir, "latest_net_G.pth"))
    og = O.Generator(1, 1, 8, 9)
    og.load_state_dict(sdG)                                               # strict: identical key set
    sdR = torch.load(os.path.join(model.save_dir, "latest_net_R.pth"))
    assert "transformer.grid" in sdR and "integrate.transformer.grid" in sdR   # like the reference's R checkpoints
    orr = O.VxmDense((64, 64), O.PLUGIN_UNET_FEATURES, 7, True)
    missing, unexpected = orr.load_state_dict(sdR, strict=False)
    assert not missing and all(k.endswith(".grid") for k in unexpected)
    # perturb, reload, compare
    with torch.no_grad():
        for p in model.netG.parameters():
            p.add_(1.0)
    model.load_networks("latest")
    for k, v in model.netG.state_dict().items():
        assert torch.equal(v.cpu(), sdG[k]), k


@pytest.mark.gpu
def test_inference_path_vs_oracle():
    from oracle import dfmir_oracle as O
    from dfmir_amd.infer import register_pair
    from tests.test_gpu_models import _hip_model_from_oracle
    from tests.test_gpu_ops import close
    torch.manual_seed(9)
    st = O.RegistrationStep(64, 2, ngf=8)
    with torch.no_grad():
        st.netR.flow.weight.mul_(1e5)
    model, opt = _hip_model_from_oracle(st, 64, 2, 8)
    model.eval()
    A, B = C.image_pair(10, 2, 64, 64)
    label = (C.rand(11, 2, 1, 64, 64) * 4).floor()
    out = register_pair(model, {"A": A, "B": B, "A_paths": ["", ""], "B_paths": ["", ""]}, label)
    with torch.no_grad():
        tb = st.netG(B)
        ys, flow = st.netR(A, B, registration=True)
        lab = O.spatial_transform(label, flow, 'nearest')
    close(out["translated_B"], tb, what="G(real_B)")
    close(out["flow"], flow, what="flow"); close(out["warped_A"], ys, what="warped A")
    frac = float((out["warped_label"].cpu() != lab).float().mean())
    assert frac < 0.01, "nearest label warp differs on %.3f of pixels (ties only)" % frac
    assert set(np.unique(out["warped_label"].cpu().numpy())) <= {0.0, 1.0, 2.0, 3.0}


@pytest.mark.gpu
def test_training_driver_two_epochs(tmp_path):
    """python -m dfmir_amd.train end to end on a synthetic folder dataset (ngf 8, 64x64)."""
    from dfmir_amd import train
    root = os.path.join(str(tmp_path), "data")
    _make_folders(root, n=4, size=72)
    ck = os.path.join(str(tmp_path), "ck")
    train.main(["--dataroot", root, "--name", "t", "--checkpoints_dir", ck, "--batch_size", "2", "--ngf", "8",
                "--load_size", "64", "--crop_size", "64", "--n_epochs", "1", "--n_epochs_decay", "1",
                "--print_freq", "2", "--save_epoch_freq", "1", "--num_threads", "0"])
    log = open(os.path.join(ck, "t", "loss_log.txt")).read()
    assert "(epoch: 2," in log and "NCE_Y:" in log and "nan" not in log.lower()
    for nm in ("G", "F", "R"):
        assert os.path.exists(os.path.join(ck, "t", "2_net_%s.pth" % nm))


@pytest.mark.gpu
def test_inference_driver_writes_the_reference_outputs(tmp_path):
    """python -m dfmir_amd.test (reference test.py:14-91): train one epoch on a synthetic folder, then the test driver
    with isTrain=False loads <epoch>_net_{G,R}.pth, registers every pair and writes deform_trainA/ and deform_label/ in
    torchvision.utils.save_image's format; the label warp is checked against the oracle's nearest-mode SpatialTransformer
    on the driver's own flow, the moved image against its own tensor."""
    from dfmir_amd import test as test_driver
    from dfmir_amd import train
    from oracle import dfmir_oracle as O
    root = os.path.join(str(tmp_path), "data")
    _make_folders(root, n=3, size=64)
    os.makedirs(os.path.join(root, "trainA_label"))
    rng = np.random.RandomState(5)
    for i in range(3):
        lab = np.zeros((64, 64), dtype=np.uint8)                                 # the flow's size (test.py:80 hard-codes 256)
        lab[16:48, 20:52] = 255
        lab[rng.randint(0, 64, 40), rng.randint(0, 64, 40)] = 128
        Image.fromarray(lab, mode="L").save(os.path.join(root, "trainA_label", "trainA_%02d.png" % i))
    ck = os.path.join(str(tmp_path), "ck")
    train.main(["--dataroot", root, "--name", "t", "--checkpoints_dir", ck, "--batch_size", "1", "--ngf", "8",
                "--load_size", "64", "--crop_size", "64", "--n_epochs", "1", "--n_epochs_decay", "0",
                "--print_freq", "2", "--save_epoch_freq", "1", "--num_threads", "0"])
    assert not os.path.exists(os.path.join(ck, "t", "1_net_D.pth"))
    recs = test_driver.main(["--dataroot", root, "--name", "t", "--checkpoints_dir", ck, "--ngf", "8", "--load_size", "64",
                             "--crop_size", "64", "--phase", "train", "--epoch", "1", "--num_test", "2"])
    assert [r["name"] for r in recs] == ["trainA_00.png", "trainA_01.png"]       # num_test bounds the loop (test.py:46)
    for r in recs:
        moved = np.asarray(Image.open(r["moved"]))
        assert moved.shape == (64, 64, 3) and moved.dtype == np.uint8            # save_image: grey replicated to RGB
        want = (r["warped_A"][0, 0] / 2 + 0.5).mul(255).add(0.5).clamp(0, 255).to("cpu", torch.uint8).numpy()
        assert np.array_equal(moved[:, :, 0], want) and np.array_equal(moved[:, :, 1], want)
        # the label: nearest-mode oracle (SpatialTransformer(mode='nearest'), test.py:80-81) on the driver's own flow
        lab_in = test_driver.read_label(os.path.join(root, "trainA_label", r["name"]))
        ref = O.spatial_transform(lab_in, r["flow"].cpu(), mode='nearest')
        want = ref[0, 0].mul(255).add(0.5).clamp(0, 255).to(torch.uint8).numpy()
        got = np.asarray(Image.open(r["label"]))
        assert got.shape == (64, 64, 3)
        assert float((got[:, :, 0] != want).mean()) < 2e-3                      # (a tie at exactly .5 voxels may round either way)


@pytest.mark.gpu
def test_captured_training_run_stays_finite_and_learns():
    """150 steps of the default step in capture mode (two eager steps, then one hipGraph replay per step) on synthetic
    slices: every loss stays finite, fresh device-drawn patch ids and inputs reach every replay, and the contrastive
    terms leave their chance level log(257) * lambda_NCE = 1.387 -- the captured graph really trains the networks."""
    from dfmir_amd.registration_model import REGISTRATIONModel
    from tests.golden import common as C
    size, B = 64, 4
    opt = default_options(batch_size=B, crop_size=size, load_size=size, ngf=8, gpu_ids=[0], checkpoints_dir="/tmp/dfmir_ckpt",
                          name="long", capture_step=True)
    torch.manual_seed(11)
    model = REGISTRATIONModel(opt)
    batches = [tuple(t.to("cuda") for t in C.image_pair(400 + 2 * i, B, size, size)) for i in range(8)]
    feed = lambda i: {"A": batches[i % 8][0], "B": batches[i % 8][1], "A_paths": [""] * B, "B_paths": [""] * B}
    model.data_dependent_initialize(feed(0))
    model.setup(opt)
    model.parallelize()
    hist = []
    for it in range(150):
        model.set_input(feed(it))
        model.optimize_parameters()
        if it % 10 == 9 or it < 3:
            ls = model.get_current_losses()
            assert all(np.isfinite(v) for v in ls.values()), (it, ls)
            hist.append(ls)
    assert model._graph['graph'] is not None and model._graph['eager_steps'] == 2
    first, last = hist[0], hist[-1]
    assert last["NCE"] < first["NCE"] - 0.05 and last["NCE_Y"] < first["NCE_Y"] - 0.05, (first, last)
    assert last["R"] < first["R"], (first, last)
    for nm in ("G", "F", "R"):
        sd = getattr(model, "net" + nm).state_dict()
        assert all(torch.isfinite(v).all() for v in sd.values() if v.dtype.is_floating_point), nm
