"""Training driver with the reference's loop semantics (train.py:9-79) on the MI355X path.

    python -m dfmir_amd.train --dataroot DATA --name exp --batch_size 16            # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m dfmir_amd.train ...   # 8 GPUs

One process per GPU; `--batch_size` is the per-rank batch.  Same call order as the reference:
data_dependent_initialize -> setup -> parallelize on the first batch, then set_input + optimize_parameters.
"""
import argparse
import os
import time

import torch

from .data import create_dataset
from .options import default_options
from .registration_model import REGISTRATIONModel
from .visualizer import Visualizer


def parse(argv=None):
    d = default_options()
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataroot', required=True)
    ap.add_argument('--phase', default='train')
    ap.add_argument('--no_flip', action='store_true')
    ap.add_argument('--serial_batches', action='store_true')
    ap.add_argument('--num_threads', type=int, default=4)
    ap.add_argument('--max_dataset_size', type=float, default=float("inf"))
    ap.add_argument('--print_freq', type=int, default=100)
    ap.add_argument('--save_latest_freq', type=int, default=5000)
    ap.add_argument('--save_epoch_freq', type=int, default=5)
    for k, v in vars(d).items():
        if k in ('gpu_ids', 'isTrain'):
            continue
        if isinstance(v, bool):
            ap.add_argument('--' + k, type=lambda s: s.lower() in ('1', 'true', 'yes'), default=v)
        elif v is None:
            ap.add_argument('--' + k, default=None)
        else:
            ap.add_argument('--' + k, type=type(v), default=v)
    opt = ap.parse_args(argv)
    opt.isTrain = True
    return opt


def main(argv=None):
    opt = parse(argv)
    from . import distributed as dfdist
    rank, world, local_rank = dfdist.init_from_env("nccl")
    opt.gpu_ids = [local_rank]
    dataset = create_dataset(opt)
    model = REGISTRATIONModel(opt)
    vis = Visualizer(opt) if rank == 0 else None
    total_iters, optimize_time = 0, 0.1
    for epoch in range(opt.epoch_count, opt.n_epochs + opt.n_epochs_decay + 1):
        t_epoch = time.time()
        t_data0 = time.time()
        dataset.set_epoch(epoch)
        epoch_iter = 0
        for i, data in enumerate(dataset):
            t_iter = time.time()
            t_data = t_iter - t_data0
            bs = data["A"].size(0)
            total_iters += bs
            epoch_iter += bs
            torch.cuda.synchronize()
            t0 = time.time()
            if epoch == opt.epoch_count and i == 0:
                model.data_dependent_initialize(data)
                model.setup(opt)
                model.parallelize()
            model.set_input(data)
            model.optimize_parameters()
            torch.cuda.synchronize()
            optimize_time = (time.time() - t0) / bs * 0.005 + 0.995 * optimize_time
            if vis is not None and total_iters % opt.print_freq < bs:
                vis.print_current_losses(epoch, epoch_iter, model.get_current_losses(), optimize_time, t_data)
            if rank == 0 and total_iters % opt.save_latest_freq < bs:
                model.save_networks('latest')
            t_data0 = time.time()
        if rank == 0 and epoch % opt.save_epoch_freq == 0:
            model.save_networks('latest')
            model.save_networks(epoch)
        if rank == 0:
            print('End of epoch %d / %d \t Time Taken: %d sec' % (epoch, opt.n_epochs + opt.n_epochs_decay, time.time() - t_epoch))
        model.update_learning_rate()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
