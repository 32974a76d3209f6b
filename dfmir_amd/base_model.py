"""Plugin base class: the surface `train.py` / `test.py` drive (reference
models/base_model.py:70-258 fixes the method NAMES and their observable effects, not their bodies).

What a driver may rely on:
  construction      opt.gpu_ids / isTrain / checkpoints_dir / name  ->  self.device, self.save_dir
  per-run           data_dependent_initialize(data) -> setup(opt) -> parallelize()
  per-step          set_input(data) -> optimize_parameters()
  read-outs         get_current_losses() -> {name: float} in loss_names order
                    get_current_visuals() -> {name: tensor} in visual_names order
                    get_image_paths(), compute_visuals()
  persistence       save_networks(tag) / load_networks(tag): <save_dir>/<tag>_net_<X>.pth = state_dict of net<X>
  schedule          update_learning_rate()
  inference         eval(), test()

Multi-GPU differs from the reference by design: one process per GPU (dfmir_amd.distributed), so `parallelize()`
broadcasts rank 0's flat weight arenas instead of wrapping modules in nn.DataParallel, and `sync_gradients()`
all-reduces each network's flat gradient arena once after backward.
"""
import os
from collections import OrderedDict

import torch

from . import distributed as dfdist
from . import networks


class _JoinedWork(object):
    """wait() for the two halves of one arena's exchange."""

    def __init__(self, *works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


class BaseModel(object):
    def __init__(self, opt):
        self.opt = opt
        self.isTrain = opt.isTrain
        self.gpu_ids = opt.gpu_ids
        self.device = torch.device('cuda', self.gpu_ids[0]) if len(self.gpu_ids) else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names, self.model_names, self.visual_names = [], [], []
        self.optimizers, self.schedulers = [], []
        self.image_paths = []
        self.metric = 0          # what a 'plateau' schedule watches
        self._ddp = False

    # ---- hooks a plugin overrides ---------------------------------------------------------------
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def set_input(self, input):
        raise NotImplementedError

    def forward(self):
        raise NotImplementedError

    def optimize_parameters(self):
        raise NotImplementedError

    def data_dependent_initialize(self, data):
        return None

    def compute_visuals(self):
        return None

    # ---- helpers ----------------------------------------------------------------------------------
    def _networks(self):
        """(short name, module) for every entry of model_names."""
        return [(n, getattr(self, 'net' + n)) for n in self.model_names if isinstance(n, str)]

    def _arena_optimizers(self):
        return [o for o in self.optimizers if hasattr(o, 'flat_p')]

    def _checkpoint_path(self, tag, name, directory=None):
        return os.path.join(directory or self.save_dir, '{}_net_{}.pth'.format(tag, name))

    # ---- run set-up ---------------------------------------------------------------------------------
    def setup(self, opt):
        """One LR scheduler per optimizer (training); weights from disk for inference / --continue_train."""
        if self.isTrain:
            self.schedulers = [networks.get_scheduler(o, opt) for o in self.optimizers]
        if not self.isTrain or opt.continue_train:      # reference order (base_model.py:97): TestOptions has no continue_train
            self.load_networks(opt.epoch)
        self.print_networks(opt.verbose)

    def parallelize(self):
        """Called once, after data_dependent_initialize (netF exists only from then on)."""
        self._ddp = dfdist.is_distributed()
        if not self._ddp:
            return
        for o in self._arena_optimizers():
            dfdist.broadcast_arena(o.flat_p, src=0)
            o.grad_scale = 1.0 / dfdist.world_size()      # the average is folded into the Adam kernel
        from . import ops
        ops.bump_weights_epoch()                          # packed-weight caches: rank 0's weights now
        self._setup_gradient_buckets()

    def _setup_gradient_buckets(self):
        """Subclasses may split an arena into depth buckets whose exchange starts inside backward."""
        self._early = {}

    def sync_gradients(self, async_op=False):
        """All-reduce every network's flat gradient arena (in self.optimizers order).  async_op=True returns one
        work handle (or None: already complete) per arena.  An arena whose LATE part [off, n) already left from inside
        backward (self._early[id(optimizer)] = (off, work)) exchanges only its remaining head [0, off) here; the handle
        returned for it waits for both."""
        if not self._ddp:
            return []
        early = getattr(self, '_early', {})
        flats, joins = [], []
        for o in self._arena_optimizers():
            e = early.pop(id(o), None)
            flats.append(o.flat_g if e is None else o.flat_g[:e[0]])
            joins.append(None if e is None else e[1])
        works = dfdist.allreduce_arenas(flats, async_op=async_op)
        if not async_op:
            for j in joins:
                if j is not None:
                    j.wait()
            return []
        return [w if j is None else _JoinedWork(w, j) for w, j in zip(works, joins)]

    # ---- inference -----------------------------------------------------------------------------------
    def eval(self):
        for _, net in self._networks():
            net.eval()

    def test(self):
        with torch.no_grad():
            self.forward()
            self.compute_visuals()

    def get_image_paths(self):
        return self.image_paths

    # ---- read-outs -------------------------------------------------------------------------------------
    def get_current_visuals(self):
        return OrderedDict((n, getattr(self, n)) for n in self.visual_names if isinstance(n, str))

    def get_current_losses(self):
        """float() of every loss_<name> (a device->host sync, as in the reference)."""
        out = OrderedDict()
        for n in self.loss_names:
            if isinstance(n, str):
                v = getattr(self, 'loss_' + n)
                out[n] = float(v.detach()) if torch.is_tensor(v) else float(v)
        return out

    def update_learning_rate(self):
        plateau = self.opt.lr_policy == 'plateau'
        for s in self.schedulers:
            s.step(self.metric) if plateau else s.step()
        print('learning rate = %.7f' % self.optimizers[0].param_groups[0]['lr'])

    # ---- persistence -------------------------------------------------------------------------------------
    def save_networks(self, epoch):
        """Plain state_dicts on the host, key names as the reference's modules produce them (row N2)."""
        os.makedirs(self.save_dir, exist_ok=True)
        for name, net in self._networks():
            host = OrderedDict((k, v.detach().to('cpu')) for k, v in net.state_dict().items())
            torch.save(host, self._checkpoint_path(epoch, name))

    def load_networks(self, epoch):
        pretrained = getattr(self.opt, 'pretrained_name', None) if self.opt.isTrain else None
        directory = os.path.join(self.opt.checkpoints_dir, pretrained) if pretrained is not None else self.save_dir
        for name, net in self._networks():
            path = self._checkpoint_path(epoch, name, directory)
            print('loading the model from %s' % path)
            state = torch.load(path, map_location=str(self.device))
            state = OrderedDict(state.items())            # drops a pickled _metadata attribute, if any
            net.load_state_dict(state)
        from . import ops
        ops.bump_weights_epoch()

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name, net in self._networks():
            if verbose:
                print(net)
            count = sum(p.numel() for p in net.parameters())
            print('[Network %s] Total number of parameters : %.3f M' % (name, count / 1e6))
        print('-----------------------------------------------')
