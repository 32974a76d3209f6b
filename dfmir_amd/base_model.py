"""BaseModel with the reference's surface (models/base_model.py:8-258): the methods train.py /
test.py call, in the same order and with the same side effects, minus DataParallel (replaced by
one-process-per-GPU data parallelism, see dfmir_amd/distributed.py)."""
import os
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch

from . import distributed as dfdist
from . import networks


class BaseModel(ABC):
    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.device = torch.device('cuda:{}'.format(self.gpu_ids[0])) if self.gpu_ids else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names = []
        self.model_names = []
        self.visual_names = []
        self.optimizers = []
        self.image_paths = []
        self.metric = 0

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @abstractmethod
    def set_input(self, input):
        pass

    @abstractmethod
    def forward(self):
        pass

    @abstractmethod
    def optimize_parameters(self):
        pass

    def setup(self, opt):
        """Create schedulers, optionally load networks (base_model.py:89-101)."""
        if self.isTrain:
            self.schedulers = [networks.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
        if not self.isTrain or opt.continue_train:
            self.load_networks(opt.epoch)
        self.print_networks(opt.verbose)

    def parallelize(self):
        """Reference: wrap nets in nn.DataParallel (base_model.py:103-107).  Here: when
        torch.distributed is initialised, broadcast rank 0's weights once and switch the
        optimisers to all-reduced, world-averaged gradients."""
        self._ddp = dfdist.is_distributed()
        if self._ddp:
            for opt_ in self.optimizers:
                if hasattr(opt_, 'flat_p'):
                    dfdist.broadcast_arena(opt_.flat_p, src=0)
                    opt_.grad_scale = 1.0 / dfdist.world_size()
            from . import ops
            ops.bump_weights_epoch()

    def sync_gradients(self):
        if getattr(self, '_ddp', False):
            dfdist.allreduce_arenas([o.flat_g for o in self.optimizers if hasattr(o, 'flat_g')])

    def data_dependent_initialize(self, data):
        pass

    def eval(self):
        for name in self.model_names:
            if isinstance(name, str):
                getattr(self, 'net' + name).eval()

    def test(self):
        with torch.no_grad():
            self.forward()
            self.compute_visuals()

    def compute_visuals(self):
        pass

    def get_image_paths(self):
        return self.image_paths

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            if self.opt.lr_policy == 'plateau':
                scheduler.step(self.metric)
            else:
                scheduler.step()
        lr = self.optimizers[0].param_groups[0]['lr']
        print('learning rate = %.7f' % lr)

    def get_current_visuals(self):
        visual_ret = OrderedDict()
        for name in self.visual_names:
            if isinstance(name, str):
                visual_ret[name] = getattr(self, name)
        return visual_ret

    def get_current_losses(self):
        errors_ret = OrderedDict()
        for name in self.loss_names:
            if isinstance(name, str):
                v = getattr(self, 'loss_' + name)
                errors_ret[name] = float(v.detach()) if torch.is_tensor(v) else float(v)
        return errors_ret

    def save_networks(self, epoch):
        """<save_dir>/<epoch>_net_<name>.pth = plain state_dict (base_model.py:164-180)."""
        os.makedirs(self.save_dir, exist_ok=True)
        for name in self.model_names:
            if isinstance(name, str):
                save_path = os.path.join(self.save_dir, '%s_net_%s.pth' % (epoch, name))
                net = getattr(self, 'net' + name)
                torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, save_path)

    def load_networks(self, epoch):
        for name in self.model_names:
            if isinstance(name, str):
                load_filename = '%s_net_%s.pth' % (epoch, name)
                if self.opt.isTrain and getattr(self.opt, 'pretrained_name', None) is not None:
                    load_dir = os.path.join(self.opt.checkpoints_dir, self.opt.pretrained_name)
                else:
                    load_dir = self.save_dir
                load_path = os.path.join(load_dir, load_filename)
                net = getattr(self, 'net' + name)
                print('loading the model from %s' % load_path)
                state_dict = torch.load(load_path, map_location=str(self.device))
                if hasattr(state_dict, '_metadata'):
                    del state_dict._metadata
                net.load_state_dict(state_dict)

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name in self.model_names:
            if isinstance(name, str):
                net = getattr(self, 'net' + name)
                num_params = sum(p.numel() for p in net.parameters())
                if verbose:
                    print(net)
                print('[Network %s] Total number of parameters : %.3f M' % (name, num_params / 1e6))
        print('-----------------------------------------------')

    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad

    def generate_visuals_for_evaluation(self, data, mode):
        return {}
