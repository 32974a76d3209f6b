"""Inference driver with the reference's semantics (test.py:14-91) on the MI355X path.

    python -m dfmir_amd.test --dataroot DATA --name exp [--epoch latest] [--num_test 50] [--phase test]

Like the reference it builds the model with `isTrain=False` (model_names G, R; registration_model.py:88), loads
`<checkpoints_dir>/<name>/<epoch>_net_{G,R}.pth` in `setup`, then per image pair: `model.test()`, the translation of B
(`netG(real_B)`, test.py:77), `netR(real_A, real_B, registration=True)` -> (moved A, flow) (test.py:78), the label of A
warped by the same flow with nearest-neighbour sampling (test.py:80-81) -- on the device, the reference moves it to the
CPU -- and writes `<dataroot>/deform_label/<name>` and `<dataroot>/deform_trainA/<name>` (test.py:84-90), in the format
`torchvision.utils.save_image` produces for one image (x * 255 + 0.5, clamped, 8-bit, grey replicated to RGB).
Labels are read from `<dataroot>/trainA_label/<name>` with the names of `<dataroot>/<phase>A` in sorted order, as the
reference does; a pair without a label file gets no `deform_label` output (the reference raises).
"""
import argparse
import os

import numpy as np
import torch
from PIL import Image

from .data import create_dataset
from .infer import register_pair
from .options import default_options
from .registration_model import REGISTRATIONModel


def parse(argv=None):
    d = default_options()
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataroot', required=True)
    ap.add_argument('--phase', default='test')                     # options/test_options.py:13
    ap.add_argument('--results_dir', default='./results/')         # options/test_options.py:12 (web pages: not written)
    ap.add_argument('--num_test', type=int, default=50)            # options/test_options.py:17
    ap.add_argument('--eval', action='store_true')
    ap.add_argument('--max_dataset_size', type=float, default=float("inf"))
    ap.add_argument('--label_dir', default='trainA_label', help="sub-folder of dataroot with A's label maps (test.py:66)")
    for k, v in vars(d).items():
        if k in ('gpu_ids', 'isTrain', 'capture_step'):
            continue
        if isinstance(v, bool):
            ap.add_argument('--' + k, type=lambda s: s.lower() in ('1', 'true', 'yes'), default=v)
        elif v is None:
            ap.add_argument('--' + k, default=None)
        else:
            ap.add_argument('--' + k, type=type(v), default=v)
    opt = ap.parse_args(argv)
    # test.py:16-21: hard-coded for the test phase
    opt.isTrain = False
    opt.num_threads = 0
    opt.batch_size = 1
    opt.serial_batches = True
    opt.no_flip = True
    opt.capture_step = False
    opt.gpu_ids = [0]
    return opt


def save_image(t, path):
    """torchvision.utils.save_image for ONE image [1,C,H,W] or [C,H,W] in [0,1]: mul(255).add(0.5).clamp(0,255) -> uint8,
    a single channel replicated to RGB (make_grid does that), written by PIL (test.py:85,90)."""
    if t.dim() == 4:
        t = t[0]
    if t.shape[0] == 1:
        t = t.expand(3, -1, -1)
    a = t.detach().float().mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to('cpu', torch.uint8).numpy()
    Image.fromarray(a).save(path)


def read_label(path):
    """transforms.ToTensor() of the label image (test.py:66-72): 8-bit modes scaled to [0,1], [1,C,H,W]."""
    img = Image.open(path)
    a = np.asarray(img)
    if a.dtype == np.uint8:
        t = torch.from_numpy(a.astype(np.float32) / 255.0)
    else:                                                           # I / I;16 / F: ToTensor keeps the values
        t = torch.from_numpy(a.astype(np.float32))
    t = t[None] if t.dim() == 2 else t.permute(2, 0, 1)
    return t[None].contiguous()


def main(argv=None):
    opt = parse(argv)
    dataset = create_dataset(opt)
    model = REGISTRATIONModel(opt)
    phase_dir = os.path.join(opt.dataroot, opt.phase + 'A')
    if not os.path.isdir(phase_dir) and os.path.isdir(os.path.join(opt.dataroot, 'valA')):
        phase_dir = os.path.join(opt.dataroot, 'valA')
    names = sorted(os.listdir(phase_dir))                          # test.py:31-32
    out_label = os.path.join(opt.dataroot, 'deform_label')
    out_moved = os.path.join(opt.dataroot, 'deform_trainA')
    written = []
    for i, data in enumerate(dataset):
        if i == 0:
            model.data_dependent_initialize(data)
            model.setup(opt)                                       # loads <epoch>_net_{G,R}.pth (isTrain False)
            model.parallelize()
            model.eval()
        if i >= opt.num_test:
            break
        print(i)
        print(data["A_paths"][0])
        label_path = os.path.join(str(opt.dataroot), opt.label_dir, str(names[i]))
        label = read_label(label_path) if os.path.exists(label_path) else None
        out = register_pair(model, data, label)
        os.makedirs(out_moved, exist_ok=True)
        save_image(out['warped_A'] / 2 + 0.5, os.path.join(out_moved, str(names[i])))      # test.py:88-90
        rec = {"name": str(names[i]), "moved": os.path.join(out_moved, str(names[i])), "flow": out['flow'],
               "warped_A": out['warped_A']}
        if label is not None:
            os.makedirs(out_label, exist_ok=True)
            save_image(out['warped_label'], os.path.join(out_label, str(names[i])))        # test.py:83-85
            rec["label"] = os.path.join(out_label, str(names[i]))
        written.append(rec)
    return written


if __name__ == '__main__':
    main()
