"""Unpaired slice dataset -- the step right BEFORE the hot path.

Mirrors the semantics of the reference's `data/unaligned_dataset.py:20-87` + `data/base_dataset.py:82-145`
+ `data/image_folder.py:13-40` + `data/__init__.py:60-98` for the options the registration model uses
(`--preprocess resize_and_crop`, grayscale): folders `<dataroot>/<phase>A`, `<phase>B`; sorted paths;
B index = A index modulo |B|; joint random left-right flip of the pair in training; then per image
Grayscale -> bicubic Resize(load_size) -> RandomCrop(crop_size) -> round to a multiple of 4 ->
RandomHorizontalFlip (unless --no_flip) -> ToTensor -> Normalize(0.5, 0.5).  PIL + numpy only; the two
torchvision random transforms draw from the torch RNG in torchvision's order, the pair flip from `random`
(data/unaligned_dataset.py:73-76).  Pinned by tests/golden/dataset.npz = the reference's own dataset class run
over seeded image folders.
"""
import os
import random

import numpy as np
import torch
from PIL import Image

IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP',
                  '.tif', '.TIF', '.tiff', '.TIFF']


def is_image_file(filename):
    return any(filename.endswith(ext) for ext in IMG_EXTENSIONS)


def make_dataset(dir, max_dataset_size=float("inf")):
    assert os.path.isdir(dir) or os.path.islink(dir), '%s is not a valid directory' % dir
    images = []
    for root, _, fnames in sorted(os.walk(dir, followlinks=True)):
        for fname in fnames:
            if is_image_file(fname):
                images.append(os.path.join(root, fname))
    return images[:int(min(max_dataset_size, len(images)))]


def slice_transform(img, opt, load_size=None):
    """PIL image -> float32 tensor [1, H, W] in [-1, 1] (get_transform(opt, grayscale=True))."""
    load_size = opt.load_size if load_size is None else load_size
    img = img.convert('L')
    pre = getattr(opt, 'preprocess', 'resize_and_crop')
    if 'resize' in pre:
        img = img.resize((load_size, load_size), Image.BICUBIC)
    if 'crop' in pre:
        w, h = img.size
        cs = opt.crop_size
        if w < cs or h < cs:
            raise ValueError("crop_size %d larger than the %dx%d image" % (cs, w, h))
        if (w, h) != (cs, cs):      # torchvision RandomCrop: top row, then left column, from the torch RNG
            y = int(torch.randint(0, h - cs + 1, size=(1,)).item())
            x = int(torch.randint(0, w - cs + 1, size=(1,)).item())
            img = img.crop((x, y, x + cs, y + cs))
    ow, oh = img.size
    h4, w4 = int(round(oh / 4) * 4), int(round(ow / 4) * 4)
    if (h4, w4) != (oh, ow):
        img = img.resize((w4, h4), Image.BICUBIC)
    if not getattr(opt, 'no_flip', False) and float(torch.rand(1)) < 0.5:   # torchvision RandomHorizontalFlip
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    a = np.asarray(img, dtype=np.float32) / 255.0
    return ((torch.from_numpy(a)[None] - 0.5) / 0.5).contiguous()


class UnalignedDataset(torch.utils.data.Dataset):
    def __init__(self, opt):
        self.opt = opt
        phase = getattr(opt, 'phase', 'train')
        self.dir_A = os.path.join(opt.dataroot, phase + 'A')
        self.dir_B = os.path.join(opt.dataroot, phase + 'B')
        if phase == "test" and not os.path.exists(self.dir_A) and os.path.exists(os.path.join(opt.dataroot, "valA")):
            self.dir_A = os.path.join(opt.dataroot, "valA")
            self.dir_B = os.path.join(opt.dataroot, "valB")
        mds = getattr(opt, 'max_dataset_size', float("inf"))
        self.A_paths = sorted(make_dataset(self.dir_A, mds))
        self.B_paths = sorted(make_dataset(self.dir_B, mds))
        self.A_size, self.B_size = len(self.A_paths), len(self.B_paths)
        self.isTrain = opt.isTrain
        self.current_epoch = 0

    def __getitem__(self, index):
        A_path = self.A_paths[index % self.A_size]
        B_path = self.B_paths[index % self.B_size]
        A_img, B_img = Image.open(A_path), Image.open(B_path)
        finetune = self.opt.isTrain and self.current_epoch > self.opt.n_epochs
        ls = self.opt.crop_size if finetune else self.opt.load_size
        if self.isTrain and random.random() > 0.5:
            A_img = A_img.transpose(Image.FLIP_LEFT_RIGHT)
            B_img = B_img.transpose(Image.FLIP_LEFT_RIGHT)
        return {'A': slice_transform(A_img, self.opt, ls), 'B': slice_transform(B_img, self.opt, ls),
                'A_paths': A_path, 'B_paths': B_path}

    def __len__(self):
        return max(self.A_size, self.B_size)


class CustomDatasetDataLoader(object):
    """data/__init__.py:60-98, plus rank-sharding (DistributedSampler) when torch.distributed is up."""

    def __init__(self, opt):
        self.opt = opt
        self.dataset = UnalignedDataset(opt)
        sampler = None
        shuffle = not getattr(opt, 'serial_batches', False)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self.dataset, shuffle=shuffle)
            shuffle = False
        self.sampler = sampler
        self.dataloader = torch.utils.data.DataLoader(
            self.dataset, batch_size=opt.batch_size, shuffle=shuffle, sampler=sampler,
            num_workers=int(getattr(opt, 'num_threads', 0)), drop_last=bool(opt.isTrain), pin_memory=True)

    def set_epoch(self, epoch):
        self.dataset.current_epoch = epoch
        if self.sampler is not None:
            self.sampler.set_epoch(epoch)

    def load_data(self):
        return self

    def __len__(self):
        return int(min(len(self.dataset), getattr(self.opt, 'max_dataset_size', float("inf"))))

    def __iter__(self):
        mds = getattr(self.opt, 'max_dataset_size', float("inf"))
        for i, data in enumerate(self.dataloader):
            if i * self.opt.batch_size >= mds:
                break
            yield data


def create_dataset(opt):
    return CustomDatasetDataLoader(opt).load_data()
