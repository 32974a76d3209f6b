"""REGISTRATIONModel -- the `--model registration` plugin (reference
models/registration_model.py:34-263) on the MI355X-native kernels.

Same class name, option setter, attribute names (netG/netF/netR, loss_*, visuals), call order and
loss algebra as the reference, including its quirks (double lambda_NCE on loss_local,
PatchNCELoss grouping by opt.batch_size, NCE layer 0 = the reflection-padded input).  Deliberate
differences, all outside the arithmetic of the losses:
  * `.cuda()` hard-codes become `self.device`;
  * the `dvf` checkerboard is decoded once and expanded to the batch (the reference re-reads
    ./deform256.jpg every step and crashes for batch_size > 1, registration_model.py:148-149);
  * Adam runs as one fused launch per network over a flat parameter arena (dfmir_amd.optim);
  * multi-GPU = one process per GPU + one RCCL all-reduce per network after backward.
"""
import os

import numpy as np
import torch

from . import networks, ops
from . import voxelmorph as vxm
from .base_model import BaseModel
from .losses import NCC_Loss, smooothing_loss
from .optim import FlatAdam
from .patchnce import PatchNCELoss
from .voxelmorph import SpatialTransformer


def str2bool(v):
    """util/util.py:13-21."""
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    import argparse
    raise argparse.ArgumentTypeError('Boolean value expected.')


def open_image_to_torch(path, size):
    """CenterCrop(size) + ToTensor + Normalize(0.5, 0.5) of an image file -> [1,C,size,size]
    (registration_model.py:14-23), without torchvision."""
    from PIL import Image
    im = Image.open(path)
    w, h = im.size
    left, top = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
    im = im.crop((left, top, left + size, top + size))
    a = np.asarray(im, dtype=np.float32) / 255.0
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
    return ((t - 0.5) / 0.5).unsqueeze(0)


def synthetic_checkerboard(size, channels=3, cell=16):
    """Stand-in for the reference's deform256.jpg asset when it is not in the working directory."""
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing='ij')
    board = (((yy // cell) + (xx // cell)) % 2).float() * 2.0 - 1.0
    return board[None, None].repeat(1, channels, 1, 1)


class REGISTRATIONModel(BaseModel):
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        """registration_model.py:35-71."""
        parser.add_argument('--CUT_mode', type=str, default="CUT", choices='(CUT, cut, FastCUT, fastcut)')
        parser.add_argument('--lambda_GAN', type=float, default=0.0, help='weight for GAN loss：GAN(G(X))')
        parser.add_argument('--lambda_NCE', type=float, default=0.25, help='weight for NCE loss: NCE(G(X), X)')
        parser.add_argument('--nce_idt', type=str2bool, nargs='?', const=True, default=False)
        parser.add_argument('--nce_layers', type=str, default='0,4,8,12,16')
        parser.add_argument('--nce_includes_all_negatives_from_minibatch', type=str2bool, nargs='?', const=True, default=False)
        parser.add_argument('--netF', type=str, default='mlp_sample', choices=['sample', 'reshape', 'mlp_sample'])
        parser.add_argument('--netF_nc', type=int, default=256)
        parser.add_argument('--nce_T', type=float, default=0.07)
        parser.add_argument('--num_patches', type=int, default=256)
        parser.add_argument('--flip_equivariance', type=str2bool, nargs='?', const=True, default=False)
        parser.set_defaults(pool_size=0)
        opt, _ = parser.parse_known_args()
        if opt.CUT_mode.lower() == "cut":
            parser.set_defaults(nce_idt=True, lambda_NCE=0.25)
        elif opt.CUT_mode.lower() == "fastcut":
            parser.set_defaults(nce_idt=False, lambda_NCE=10.0, flip_equivariance=True, n_epochs=150, n_epochs_decay=50)
        else:
            raise ValueError(opt.CUT_mode)
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        # opt.deterministic_wgrad (build-defined, default False): weight / bias gradients through 64-bit fixed-point
        # accumulation -- bit-identical arenas from run to run (a PROCESS-global switch of dfmir_amd.ops)
        ops.set_deterministic_wgrad(bool(getattr(opt, 'deterministic_wgrad', False)) or ops._env_on('DFMIR_DETERMINISTIC_WGRAD'))
        self.loss_names = ['G', 'NCE', 'R', 'smooth', 'local']
        self.visual_names = ['real_A', 'fake_B', 'real_B', 'dvf', 'registered', 'regA']
        self.nce_layers = [int(i) for i in self.opt.nce_layers.split(',')]
        if opt.nce_idt and self.isTrain:
            self.loss_names += ['NCE_Y']
            self.visual_names += ['idt_B']
        self.model_names = ['G', 'F', 'R'] if self.isTrain else ['G', 'R']

        self.netG = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.normG, not opt.no_dropout,
                                      opt.init_type, opt.init_gain, opt.no_antialias, opt.no_antialias_up, self.gpu_ids, opt)
        self.netF = networks.define_F(opt.input_nc, opt.netF, opt.normG, not opt.no_dropout, opt.init_type,
                                      opt.init_gain, opt.no_antialias, self.gpu_ids, opt)
        nb_features = [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]]
        vol_shape = (opt.crop_size, opt.crop_size)
        self.netR = vxm.networks.VxmDense(vol_shape, nb_features, int_steps=7, bidir=True).to(self.device)
        self.netR.train()
        # registration_model.py:143-147 read y_output[0] and [2] only: the discarded warp(real_B, -flow) branch is not computed
        self.netR.skip_unused_target = bool(getattr(opt, 'skip_unused_target', True))
        self.spatialTransformer = SpatialTransformer(vol_shape).to(self.device)
        self._dvf_image = None

        if self.isTrain:
            self.criterionGAN = networks.GANLoss(opt.gan_mode).to(self.device)
            self.criterionNCE = [PatchNCELoss(opt).to(self.device) for _ in self.nce_layers]
            self.criterionIdt = None  # torch.nn.L1Loss in the reference; constructed, never called
            self.criterionNCC = NCC_Loss(self.device, name='ncc', kernel_var=[9, 9], kernel_type='mean')
            self.optimizer_G = FlatAdam(self.netG.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2))
            self.optimizer_R = FlatAdam(self.netR.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2))
            self.optimizers.append(self.optimizer_G)
            self.optimizers.append(self.optimizer_R)

    # -- registration_model.py:119-136
    def data_dependent_initialize(self, data):
        if getattr(self.opt, 'capture_step', False) and self.isTrain and not getattr(self, '_in_side_stream', False):
            # a model that will capture its step lives on ONE side stream from its first backward on (autograd binds
            # each parameter's AccumulateGrad node to the stream of its first use; see _optimize_parameters_graphed)
            side, cur = self._graph_state()['stream'], torch.cuda.current_stream()
            side.wait_stream(cur)
            self._in_side_stream = True
            try:
                with torch.cuda.stream(side):
                    self.data_dependent_initialize(data)
            finally:
                self._in_side_stream = False
            cur.wait_stream(side)
            return
        self.set_input(data)
        bs_per_gpu = self.real_A.size(0) // max(len(self.opt.gpu_ids), 1)
        self.real_A = self.real_A[:bs_per_gpu]
        self.real_B = self.real_B[:bs_per_gpu]
        self.forward()
        if self.opt.isTrain:
            self.compute_G_loss().backward()
            if self.opt.lambda_NCE > 0.0:
                self.optimizer_F = FlatAdam(self.netF.parameters(), lr=self.opt.lr, betas=(self.opt.beta1, self.opt.beta2))
                self.optimizers.append(self.optimizer_F)

    def set_dvf_image(self, img):
        """Install the test pattern `dvf` warps: a [1,C,H,W] tensor in [-1,1] at the model's crop size."""
        if img.dim() != 4 or img.shape[0] != 1 or tuple(img.shape[2:]) != (self.opt.crop_size,) * 2:
            raise ValueError("dvf image must be [1,C,%d,%d], got %s" % (self.opt.crop_size, self.opt.crop_size, tuple(img.shape)))
        self._dvf_image = img.to(self.device, torch.float32).contiguous()

    def _checkerboard(self, batch):
        """The visual-only test pattern warped into `dvf` (registration_model.py:148-149): `./deform256.jpg` decoded
        with CenterCrop(256) like the reference (for crop_size < 256: its top-left crop_size window), once.  A missing
        file raises, as the reference does; `opt.dvf_image = 'synthetic'` asks for a generated checkerboard instead
        (benchmarks / tests on a box without the asset), any other string is a path."""
        if self._dvf_image is None:
            size = self.opt.crop_size
            choice = getattr(self.opt, 'dvf_image', None)
            if choice == 'synthetic':
                img = synthetic_checkerboard(size)
            else:
                path = choice or "./deform256.jpg"
                if not os.path.exists(path):
                    raise FileNotFoundError("%s (the dvf test pattern, reference registration_model.py:148); "
                                            "set opt.dvf_image to a path or to 'synthetic'" % path)
                if size > 256:
                    raise ValueError("the dvf test pattern is 256x256; crop_size %d needs opt.dvf_image='synthetic'" % size)
                img = open_image_to_torch(path, 256)[:, :, :size, :size]
            self.set_dvf_image(img)
        return self._dvf_image.expand(batch, -1, -1, -1).contiguous()

    # -- registration_model.py:138-171
    def optimize_parameters(self):
        if getattr(self.opt, 'capture_step', False) and self.isTrain and not self.opt.flip_equivariance:
            return self._optimize_parameters_graphed()     # (FastCUT draws its flip on the host, step by step)
        self._forward_backward()
        self._apply_updates()

    def _apply_updates(self):
        """registration_model.py:168-171 (+ the one exchange step of the data-parallel form).  Under RCCL the three
        arena all-reduces are issued back to back (G 45.5 MB first, then R, F) and each network's Adam launch waits
        only for its own: R's and F's exchange overlaps G's optimizer pass instead of following it."""
        opts = [self.optimizer_G, self.optimizer_R] + ([self.optimizer_F] if self.opt.netF == 'mlp_sample' else [])
        b = getattr(self, '_bucket', None)
        if b is not None and b.get('in_graph') and id(self.optimizer_G) not in self._early:
            self._early[id(self.optimizer_G)] = (b['off'], None)   # the replayed graph has already exchanged the tail
        works = self.sync_gradients(async_op=True)
        timing = getattr(self, '_collective_timing', None) if works else None     # bench.py: exposed collective time
        for i, o in enumerate(opts):
            if works and i < len(works) and works[i] is not None:
                if timing is not None:
                    # the compute stream's idle time while it waits for this arena's exchange: event pair around the wait
                    s_ = torch.cuda.Event(enable_timing=True)
                    e_ = torch.cuda.Event(enable_timing=True)
                    s_.record()
                    works[i].wait()
                    e_.record()
                    timing.append((s_, e_))
                else:
                    works[i].wait()
            o.step()

    def _optimize_parameters_graphed(self):
        """The steady-state step as ONE hipGraph launch (opt.capture_step, build-defined): forward, the losses and
        backward -- ~600 kernel launches that the Python / autograd / ctypes stack needs 40-75 ms of one host core to
        enqueue -- are captured once (after two eager steps have built every cache: packed-weight tables, deferred
        weight-gradient buffers, the id generator) and replayed; the gradient all-reduce and the three Adam launches
        stay eager (bias corrections and the learning rate change per step).  Inputs are copied into static buffers;
        losses / visuals are the capture's output tensors, overwritten by every replay.  Patch ids come from the
        device generator, whose counter the graph itself advances.  A change of batch shape re-captures."""
        st = self._graph_state()
        # what the captured launches hold by address or by value: batch geometry and the parameter / gradient arenas
        shape = (tuple(self.real_A.shape), tuple(self.real_B.shape),
                 tuple((o.flat_p.data_ptr(), o.flat_g.data_ptr()) for o in self.optimizers))
        if st['graph'] is not None and st['shape'] != shape:
            st.update(graph=None, eager_steps=0)
        side, cur = st['stream'], torch.cuda.current_stream()

        def eager_step():
            # every non-replayed step of this model runs on the capture's side stream: autograd binds a parameter's
            # AccumulateGrad node to the stream it first ran on and keeps it while any tensor of an old graph lives;
            # reusing such a node from another stream is a cross-stream sync, illegal while capturing
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._forward_backward()
            cur.wait_stream(side)
            self._apply_updates()

        # capture needs two eager steps behind it (caches built) and a step whose random draws live on the device
        # (the batched NCE path; a wrapped netF / small feature maps / pinned ids go through the host generator)
        blocked = not getattr(self, '_nce_on_device', False) or (
            getattr(self.opt, 'global_mask_norm', False) and getattr(self, '_ddp', False))   # a collective inside forward
        if st['force_eager'] or (st['graph'] is None and (st['eager_steps'] < 2 or blocked)):
            if st['eager_steps'] == 2 and not st['force_eager'] and not st.get('warned'):
                st['warned'] = True
                import warnings
                warnings.warn("opt.capture_step is set but this step cannot be captured (patch ids come from the host: "
                              "batch_query_passes=False / nce_sequential_keys / a patch_id_source that is not graph_safe / "
                              "feature maps outside the device draw's sizes -- or opt.global_mask_norm puts a collective "
                              "on the forward path); it stays eager")
            st['eager_steps'] += 1
            return eager_step()
        if st['graph'] is None:
            st['in_A'], st['in_B'] = self.real_A.clone(), self.real_B.clone()
            self.real_A, self.real_B = st['in_A'], st['in_B']
            for k, v in list(vars(self).items()):                     # drop the previous step's autograd graph
                if torch.is_tensor(v) and v.grad_fn is not None:
                    setattr(self, k, v.detach())
            self._key_feats = None
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            ops.begin_graph_capture()
            # Under a process group other threads of this process (RCCL's watchdog) may touch the HIP runtime while the
            # capture is open: "thread_local" keeps their calls legal; single-process runs keep the strict default.
            mode = "thread_local" if getattr(self, '_ddp', False) else "global"
            staged = self._staged_ok()
            try:
                if staged:
                    graph = self._capture_staged(side, mode)
                else:
                    with torch.cuda.graph(graph, stream=side, capture_error_mode=mode):
                        self._forward_backward()
            except Exception as exc:                                   # noqa: BLE001 -- keep training, eagerly
                # nothing of a failed capture has executed: weights, moments and the id counter are where the last
                # eager step left them -- but the host-side caches believe the recorded re-pack / pool zero-fill ran:
                # forget them, then this step and every later one are enqueued eagerly
                ops.invalidate_after_failed_capture()
                import warnings
                warnings.warn("capturing the train step failed (%s: %s); the step stays eager" % (type(exc).__name__, exc))
                st.update(graph=None, force_eager=True, capture_error="%s: %s" % (type(exc).__name__, exc))
                torch.cuda.synchronize()
                return eager_step()
            finally:
                ops.end_graph_capture()
            st['outputs'] = {k: v for k, v in vars(self).items()
                             if torch.is_tensor(v) and (k.startswith('loss_') or k in self.visual_names or k in ('fake', 'real'))}
            st['outputs']['_loss_inputs'] = self._loss_inputs
            st.update(graph=graph, shape=shape)
        else:
            if self.real_A is not st['in_A']:
                st['in_A'].copy_(self.real_A, non_blocking=True)
                st['in_B'].copy_(self.real_B, non_blocking=True)
            vars(self).update(st['outputs'])
            self.real_A, self.real_B = st['in_A'], st['in_B']
        if isinstance(st['graph'], list):
            self._replay_staged(st['graph'])                 # eight single-stream graphs, event edges between the launches
        else:
            st['graph'].replay()
        self._apply_updates()

    def _overlap_registration(self):
        return (self.isTrain and self.device.type == 'cuda' and getattr(self.opt, 'overlap_registration', True)
                and not ops._env_on('DFMIR_NO_OVERLAP_R'))

    # -- data-parallel gradient buckets (build-defined; replaces nothing in the reference, whose DataParallel reduces on
    # device 0, base_model.py:103-107).  opt.bucket_allreduce: G's arena leaves in two depth buckets.  The query pass of the
    # NCE terms stops at module max(nce_layers) = 16, so modules 17.. (4 ResNet blocks, the decoder, the head: 45 % of G's
    # parameters) receive their only weight gradients in the first part of the main pass's backward; when the gradient of
    # module 17's input exists they are final: their deferred accumulators are flushed and the arena's tail starts its
    # all-reduce while backward goes on through the other 16 modules (and netR's stream).  The head [0, off) and the
    # other arenas follow after backward as before.
    def _setup_gradient_buckets(self):
        self._early = {}
        self._bucket = None
        self.netG.grad_boundary = None
        if not (self._ddp and getattr(self.opt, 'bucket_allreduce', False) and self.isTrain):
            return
        mods = list(self.netG.model)
        idx = max(self.nce_layers) + 1
        if not (0 < idx < len(mods)):
            return
        late = [p for m in mods[idx:] for p in m.parameters()]
        if not late:
            return
        off = 0
        for p in self.optimizer_G.param_groups[0]['params']:
            if p is late[0]:
                break
            off += p.numel()
        assert off + sum(p.numel() for p in late) == self.optimizer_G.flat_g.numel(), "arena order = module order"
        owners = set(id(sub) for m in mods[idx:] for sub in m.modules() if getattr(sub, 'weight', None) is not None)
        self._bucket = dict(idx=idx, off=off, owners=owners, fired=0)
        self.netG.grad_boundary = (idx, self._bucket_ready)

    def _bucket_ready(self):
        b = self._bucket
        if b is None or id(self.optimizer_G) in self._early:
            return
        if getattr(self.opt, 'capture_step', False) and not ops._env_on('DFMIR_BUCKET_IN_GRAPH'):
            # A model that captures its step keeps the whole exchange behind the replay -- in the eager warm-up steps and
            # after a failed capture as well: an early flush there would build the deferred-gradient job tables for a
            # different item set than the capture flushes (a table upload inside the capture), and a rank whose capture
            # failed would issue tail + head all-reduces against its peers' single one.
            return
        from . import distributed as dfdist
        ops.flush_deferred_subset(b['owners'], 'late')
        works = dfdist.allreduce_arenas([self.optimizer_G.flat_g[b['off']:]], async_op=True)
        self._early[id(self.optimizer_G)] = (b['off'], works[0] if works else None)
        b['fired'] += 1
        b['in_graph'] = bool(torch.cuda.is_current_stream_capturing())   # experimental: every replay then repeats it

    def _graph_state(self):
        return self.__dict__.setdefault('_graph', {'eager_steps': 0, 'graph': None, 'shape': None, 'force_eager': False,
                                                   'stream': torch.cuda.Stream(device=self.device)})

    def _forward_backward(self):
        if getattr(self, '_early', None):
            # a previous backward fired the bucket hook and never reached sync_gradients (it raised, or the caller ran
            # _forward_backward twice): finish those exchanges and forget them, or this step's hook would see a stale entry
            # and the arena's tail would never be reduced
            for _, w in self._early.values():
                if w is not None:
                    w.wait()
            self._early.clear()
        if getattr(self, '_bucket', None) is not None:
            self._bucket['in_graph'] = False
        if self._staged_ok():
            return self._run_staged_eager()
        if self._overlap_registration():
            # netR reads only the two input images (registration_model.py:146): its forward runs on a second stream beside
            # the generator's, and autograd then runs its backward there as well, beside the generator's backward -- a
            # few hundred small-grid launches under the big convolutions instead of in front of them.  Captured, the
            # fork / join become parallel branches of the hipGraph.
            cur, rs = torch.cuda.current_stream(), self._graph_state().setdefault('r_stream', torch.cuda.Stream(device=self.device))
            ops.prepare_step(self.device)
            rs.wait_stream(cur)
            with torch.cuda.stream(rs):
                y_output = self.netR(self.real_A, self.real_B)
                r_done = torch.cuda.Event()
                r_done.record(rs)
            self.forward()
            cur.wait_event(r_done)                         # regA exists: the NCE query pass below reads it
            if not torch.cuda.is_current_stream_capturing():
                for t in y_output:
                    if torch.is_tensor(t):
                        t.record_stream(cur)
        else:
            rs = None
            self.forward()
            y_output = self.netR(self.real_A, self.real_B)
        self.regA = y_output[0]
        self.pos_flow = y_output[2]                        # the deformation field (registration_model.py:144 keeps it in a local)

        def registration_losses():
            # registration_model.py:147-166: the warp of the translated image, the checkerboard visual, the two masked-L1
            # terms (masks evaluated inside the fused kernel: mask = (real_B > -0.95) | (registered > -0.95);
            # mask2 = (idt_B > -0.95) | (registered > -0.95)) and the flow smoothness
            y_pred = [self.spatialTransformer(self.fake_B, y_output[2]), y_output[2]]
            self.registered = y_pred[0]
            with torch.no_grad():
                self.dvf = self.spatialTransformer(self._checkerboard(self.real_A.size(0)), y_pred[1].detach())
            l1_reg = self.calculate_L1_loss(y_pred[0], self.real_B, mask='threshold')
            l1_idt = self.calculate_L1_loss(self.idt_B, y_pred[0], mask='threshold')
            if getattr(self.opt, 'global_mask_norm', False) and getattr(self, '_ddp', False):
                l1_reg, l1_idt = self._global_mask_norm(l1_reg, l1_idt)
            return l1_reg, l1_idt, smooothing_loss(y_pred[1])

        # ... which follow netR on ITS stream, beside the NCE query pass (not when the masked-L1 normalisation is a collective)
        side_losses = (rs is not None and not ops._env_on('DFMIR_NO_SIDE_LOSSES')
                       and not (getattr(self.opt, 'global_mask_norm', False) and getattr(self, '_ddp', False)))
        if side_losses:
            rs.wait_stream(cur)                            # fake_B / idt_B exist
            with torch.cuda.stream(rs):
                l1_reg, l1_idt, smooth = registration_losses()
            if not torch.cuda.is_current_stream_capturing():
                self.fake.record_stream(rs)
                for t in (l1_reg, l1_idt, smooth, self.registered, self.dvf):
                    t.record_stream(cur)
        elif rs is not None:
            cur.wait_stream(rs)

        self.optimizer_G.zero_grad()
        self.optimizer_R.zero_grad()
        if self.opt.netF == 'mlp_sample':
            self.optimizer_F.zero_grad()

        self._nce_terms = None
        idt = bool(self.opt.nce_idt)
        if (getattr(self.opt, 'batch_query_passes', True) and not ops._env_on('DFMIR_NO_STACKED_Q')
                and self.opt.lambda_NCE > 0.0 and self.opt.lambda_GAN <= 0.0):
            # The three NCE terms each run G's encoder on their own query batch (fake_B, idt_B, regA) with the same
            # weights: one pass over the three stacked along the batch instead (per-sample kernels; same random
            # draws in the same order -- only the key side draws).  Terms in the reference's order; without nce_idt
            # (FastCUT) the identity term is not computed: two terms, queries cat(fake_B, regA).
            self._nce_terms = self.calculate_NCE_losses_stacked(
                ((self.real_A, None), (self.real_B, None), (self.real_B, y_output[0])) if idt
                else ((self.real_A, self.fake_B), (self.real_B, y_output[0])))
        stacked = self._nce_terms is not None
        if stacked:
            self.loss_G_GAN = 0.0
            if idt:
                self.loss_NCE, self.loss_NCE_Y, nce_local = self._nce_terms
            else:
                (self.loss_NCE, nce_local), self.loss_NCE_Y = self._nce_terms, 0.0
            self._nce_terms = None
        else:
            self.loss_G = self.compute_G_loss()
            nce_local = self.calculate_NCE_loss(self.real_B, y_output[0])

        if side_losses:
            cur.wait_stream(rs)
        else:
            l1_reg, l1_idt, smooth = registration_losses()
        self._loss_inputs = (l1_reg, l1_idt, smooth)
        if stacked:
            # registration_model.py:163-166,230-234 as ONE launch (and one for its gradient):
            #   loss_G = (NCE + NCE_Y) * 0.5;  loss_local = nce_local * 0.25;  loss_R = l1_reg + l1_idt + loss_local
            #   loss_smooth = smooth * 0.20;   total = loss_R + loss_G + loss_smooth
            if idt:
                out = ops.scalar_combine(
                    [[0.5, 0.5, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.25, 1.0, 1.0, 0.0], [0.0, 0.0, 0.25, 0.0, 0.0, 0.0],
                     [0.0, 0.0, 0.0, 0.0, 0.0, 0.20], [0.5, 0.5, 0.25, 1.0, 1.0, 0.20]],
                    [self.loss_NCE, self.loss_NCE_Y, nce_local, l1_reg, l1_idt, smooth])
            else:                                          # loss_G = NCE (registration_model.py:228-232)
                out = ops.scalar_combine(
                    [[1.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.25, 1.0, 1.0, 0.0], [0.0, 0.25, 0.0, 0.0, 0.0],
                     [0.0, 0.0, 0.0, 0.0, 0.20], [1.0, 0.25, 1.0, 1.0, 0.20]],
                    [self.loss_NCE, nce_local, l1_reg, l1_idt, smooth])
            self.loss_G, self.loss_R, self.loss_local, self.loss_smooth, all_G_loss = out.unbind(0)
        else:
            self.loss_local = nce_local * 0.25
            self.loss_R = l1_reg * 1.0 + l1_idt * 1.0 + self.loss_local * 1.0
            self.loss_smooth = smooth * 0.20
            all_G_loss = self.loss_R + self.loss_G + self.loss_smooth
        with ops.deferred_weight_grads():
            all_G_loss.backward()
            if rs is not None:
                # netR's weight-gradient kernels ran on rs; the flush on exit and optimizer_R.step() run here.  An
                # explicit join (legal under capture: it becomes the graph's join edge) instead of relying on which
                # stream autograd bound netR's AccumulateGrad nodes to
                cur.wait_stream(rs)

    # -- the two-stream step in single-stream PIECES (build-defined; opt.staged_step, default on).  With netR on a second
    # stream the captured step used to be ONE hipGraph with parallel branches, and hipGraphLaunch (ROCm 7.2) returns from
    # such a graph only when its second branch has been handed to the GPU -- 39 ms into a 74 ms step, a host core per rank
    # doing nothing.  Here the same kernels are issued as eight pieces, each on ONE stream, with every cross-stream
    # dependency BETWEEN pieces: captured, each piece is a linear hipGraph whose launch returns at once, and the edges are
    # event waits between the launches.  What makes the pieces separable is a backward in four autograd calls instead of
    # one: the generator's output and netR's outputs enter the losses through detached leaves, each loss group is
    # back-propagated on the stream it was computed on with its coefficient of registration_model.py:163-166,230-234 as the
    # seed, and the leaves' gradients are handed to `fake.backward` / netR's backward -- the same sums in the same order
    # (every leaf collects exactly two contributions, and a + b = b + a).
    #
    #   generator stream:  prep | g_fwd ........ | q_pass (NCE query pass, fwd + bwd) | g_bwd ............ | finish
    #   netR's stream:          | r_fwd | r_loss (warp, masked L1, smoothness: fwd + bwd) |  r_bwd        |
    #   edges: prep -> r_fwd;  g_fwd -> r_loss;  r_fwd -> q_pass;  q_pass -> r_bwd;  r_loss -> g_bwd;  r_bwd -> finish
    def _staged_ok(self):
        gmn = getattr(self.opt, 'global_mask_norm', False) and getattr(self, '_ddp', False)
        return (getattr(self.opt, 'staged_step', True) and not ops._env_on('DFMIR_NO_STAGED') and self._overlap_registration()
                and not ops._env_on('DFMIR_NO_SIDE_LOSSES') and not ops._env_on('DFMIR_BUCKET_IN_GRAPH') and not gmn
                and getattr(self.opt, 'batch_query_passes', True) and not ops._env_on('DFMIR_NO_STACKED_Q')
                and self.opt.lambda_NCE > 0.0 and self.opt.lambda_GAN <= 0.0 and self.opt.netF == 'mlp_sample')

    def _loss_seeds(self, idt):
        """d total / d (NCE terms..., l1_reg, l1_idt, smooth): the last row of the step's scalar algebra, as device scalars
        (made once, outside any capture)."""
        key = (bool(idt), float(self.opt.lambda_NCE))
        c = self.__dict__.setdefault('_seed_cache', {})
        if key not in c:
            vals = [0.5, 0.5, 0.25, 1.0, 1.0, 0.20] if idt else [1.0, 0.25, 1.0, 1.0, 0.20]
            c[key] = list(torch.tensor(vals, dtype=torch.float32).to(self.device).unbind(0))
        return c[key]

    def _step_pieces(self):
        """[(name, stream 'g' | 'r', fn, names of the pieces on the OTHER stream it waits for)] -- see the comment above."""
        idt = bool(self.opt.nce_idt)
        seeds = self._loss_seeds(idt)                       # (built before anything is captured)
        n_nce = 3 if idt else 2
        S = self.__dict__.setdefault('_stage', {})
        S.clear()

        def prep():
            ops.prepare_step(self.device)
            self.optimizer_G.zero_grad()
            self.optimizer_R.zero_grad()
            self.optimizer_F.zero_grad()

        def r_fwd():
            y = self.netR(self.real_A, self.real_B)
            S['y'] = y
            self.regA, self.pos_flow = y[0], y[2]
            S['regA_q'] = y[0].detach().requires_grad_()     # the NCE query pass reads regA through this leaf ...
            S['flow_l'] = y[2].detach().requires_grad_()     # ... the warp of fake_B and the smoothness term the field

        def g_fwd():
            self.forward()
            S['fake_q'] = self.fake.detach().requires_grad_()
            S['fake_l'] = self.fake.detach().requires_grad_()

        def r_loss():
            nb = self.real_A.size(0)
            fl = S['fake_l']
            self.registered = self.spatialTransformer(fl[:nb], S['flow_l'])
            with torch.no_grad():
                self.dvf = self.spatialTransformer(self._checkerboard(nb), S['flow_l'].detach())
            l1_reg = self.calculate_L1_loss(self.registered, self.real_B, mask='threshold')
            l1_idt = self.calculate_L1_loss(fl[nb:], self.registered, mask='threshold')
            smooth = smooothing_loss(S['flow_l'])
            self._loss_inputs = (l1_reg, l1_idt, smooth)
            torch.autograd.backward([l1_reg, l1_idt, smooth], seeds[n_nce:])

        def q_pass():
            nb = self.real_A.size(0)
            fq = S['fake_q']
            if idt:
                terms = ((self.real_A, None), (self.real_B, None), (self.real_B, S['regA_q']))
                tgt = ops.cat_batch(fq, S['regA_q'])
            else:
                terms = ((self.real_A, fq[:nb]), (self.real_B, S['regA_q']))
                tgt = ops.cat_batch(fq[:nb], S['regA_q'])
            nce = self.calculate_NCE_losses_stacked(terms, tgt=tgt)
            S['nce'] = nce
            torch.autograd.backward(list(nce), seeds[:n_nce])

        def r_bwd():
            y = S['y']
            torch.autograd.backward([y[0], y[2]], [S['regA_q'].grad, S['flow_l'].grad])

        def g_bwd():
            # (FastCUT: the query pass saw fake_B only -- the slice's backward leaves zeros in idt_B's half)
            self.fake.backward(S['fake_q'].grad + S['fake_l'].grad)

        def finish():
            ops.end_deferred()
            with torch.no_grad():
                l1_reg, l1_idt, smooth = (t.detach() for t in self._loss_inputs)
                nce = [t.detach() for t in S['nce']]
                self.loss_G_GAN = 0.0
                if idt:
                    self.loss_NCE, self.loss_NCE_Y, nce_local = nce
                    out = ops.scalar_combine(
                        [[0.5, 0.5, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.25, 1.0, 1.0, 0.0], [0.0, 0.0, 0.25, 0.0, 0.0, 0.0],
                         [0.0, 0.0, 0.0, 0.0, 0.0, 0.20]], [self.loss_NCE, self.loss_NCE_Y, nce_local, l1_reg, l1_idt, smooth])
                else:
                    (self.loss_NCE, nce_local), self.loss_NCE_Y = nce, 0.0
                    out = ops.scalar_combine(
                        [[1.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.25, 1.0, 1.0, 0.0], [0.0, 0.25, 0.0, 0.0, 0.0],
                         [0.0, 0.0, 0.0, 0.0, 0.20]], [self.loss_NCE, nce_local, l1_reg, l1_idt, smooth])
                self.loss_G, self.loss_R, self.loss_local, self.loss_smooth = out.unbind(0)

        return [('prep', 'g', prep, ()), ('r_fwd', 'r', r_fwd, ('prep',)), ('g_fwd', 'g', g_fwd, ()),
                ('r_loss', 'r', r_loss, ('g_fwd',)), ('q_pass', 'g', q_pass, ('r_fwd',)), ('r_bwd', 'r', r_bwd, ('q_pass',)),
                ('g_bwd', 'g', g_bwd, ('r_loss',)), ('finish', 'g', finish, ('r_bwd',))]

    def _r_stream(self):
        return self._graph_state().setdefault('r_stream', torch.cuda.Stream(device=self.device))

    def _run_staged_eager(self):
        """The pieces enqueued one after the other on the current stream ('g') and netR's stream ('r')."""
        cur, rs = torch.cuda.current_stream(), self._r_stream()
        streams = {'g': cur, 'r': rs}
        done = {}
        ops.begin_deferred()
        try:
            for name, sk, fn, after in self._step_pieces():
                for dep in after:
                    streams[sk].wait_event(done[dep])
                with torch.cuda.stream(streams[sk]):
                    fn()
                done[name] = torch.cuda.Event()
                done[name].record(streams[sk])
        except BaseException:
            cur.wait_stream(rs)
            ops.end_deferred(failed=True)
            raise
        # tensors that crossed streams stay referenced in self._stage until the next step (no record_stream bookkeeping)

    def _capture_staged(self, side, mode):
        """Every piece into its own hipGraph (nothing executes); one private memory pool per stream."""
        rs = self._r_stream()
        streams = {'g': side, 'r': rs}
        pools = {'g': torch.cuda.graph_pool_handle(), 'r': torch.cuda.graph_pool_handle()}
        graphs = []
        ops.begin_deferred()
        try:
            for name, sk, fn, after in self._step_pieces():
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pools[sk], stream=streams[sk], capture_error_mode=mode):
                    fn()
                graphs.append((name, sk, g, after))
        except BaseException:
            ops._DEFER["on"] = False
            ops._DEFER["pending"] = {}
            raise
        return graphs

    def _replay_staged(self, graphs):
        cur, rs = torch.cuda.current_stream(), self._r_stream()
        streams = {'g': cur, 'r': rs}
        evs = self._graph_state().setdefault('events', {})
        for name, sk, g, after in graphs:
            s = streams[sk]
            for dep in after:
                s.wait_event(evs[dep])
            if s is cur:
                g.replay()
            else:
                with torch.cuda.stream(s):
                    g.replay()
            if name not in evs:
                evs[name] = torch.cuda.Event()
            evs[name].record(s)

    def _global_mask_norm(self, *terms):
        """opt.global_mask_norm (build-defined, data-parallel runs only): the reference's DataParallel evaluates
        calculate_L1_loss on the gathered GLOBAL batch, i.e. sum_ranks(S_r) / sum_ranks(M_r) with S = sum |a-b| m and
        M = sum m (registration_model.py:160-166,262); per-rank losses averaged by the gradient all-reduce give
        mean_ranks(S_r / M_r) instead.  With this option each rank's term is rescaled by world * M_r / M_total (one
        all-reduce of the two mask sums), so that the averaged gradient is exactly the global-batch one.  A collective
        on the forward path: the step is then not captured into a hipGraph."""
        import torch.distributed as dist
        from . import distributed as dfdist
        m = torch.stack([t._df_mask_sum for t in terms])
        tot = m.clone()
        if dfdist._staged(tot):
            h = tot.cpu()
            dist.all_reduce(h)
            tot = h.to(m.device)
        else:
            dist.all_reduce(tot)
        f = torch.where(tot > 0, m * float(dfdist.world_size()) / tot.clamp_min(1e-30), torch.ones_like(m))
        return tuple(t * f[i] for i, t in enumerate(terms))

    # -- registration_model.py:174-183
    def set_input(self, input):
        AtoB = self.opt.direction == 'AtoB'
        self.real_A = input['A' if AtoB else 'B'].to(self.device, non_blocking=True)
        self.real_B = input['B' if AtoB else 'A'].to(self.device, non_blocking=True)
        self.image_paths = input['A_paths' if AtoB else 'B_paths']

    # -- registration_model.py:185-196
    def forward(self):
        self.real = ops.cat_batch(self.real_A, self.real_B)
        self.flipped_for_equivariance = False
        if self.opt.flip_equivariance:
            # FastCUT (registration_model.py:188-191): with probability 1/2 the generator sees the batch mirrored along W;
            # calculate_NCE_loss mirrors the query features back.  `flip_draw` (build-defined hook, default the reference's
            # `np.random.random() < 0.5`) lets tests force the branch.  A host-side draw per step: such a step is not captured.
            draw = getattr(self, 'flip_draw', None)
            self.flipped_for_equivariance = self.isTrain and bool(draw() if draw is not None else np.random.random() < 0.5)
            if self.flipped_for_equivariance:
                self.real = torch.flip(self.real, [3])
        nb = self.real_A.size(0)
        self._key_feats = None
        if (self.isTrain and self.opt.lambda_NCE > 0.0 and getattr(self.opt, 'reuse_key_features', True)
                and not self.flipped_for_equivariance):    # (the tapped features would be those of the MIRRORED inputs)
            # The reference re-runs G's encoder on real_A / real_B for the (detached) key side of every
            # NCE term (registration_model.py:244) with the weights this very pass used.  Every
            # kernel on the path is per-sample and batch-size independent, so those activations are
            # bit-identical to the ones this pass produces: tap them here instead of recomputing
            # (ResnetGenerator.forward(layers, encode_only=False) returns them, networks.py:1028-1047).
            self.fake, feats = self.netG(self.real, self.nce_layers, encode_only=False)
            self._key_feats = ((self.real_A, [f[:nb].detach() for f in feats]),
                               (self.real_B, [f[nb:].detach() for f in feats]))
        else:
            self.fake = self.netG(self.real)
        self.fake_B = self.fake[:nb]
        self.idt_B = self.fake[nb:]

    def _encode_keys(self, src):
        for owner, feats in (self._key_feats or ()):
            if owner is src:
                return feats
        return self.netG(src, self.nce_layers, encode_only=True)

    # -- registration_model.py:213-235
    def compute_G_loss(self):
        if self.opt.lambda_GAN > 0.0:
            raise NotImplementedError("the registration model is discriminator-free (lambda_GAN must be 0)")
        self.loss_G_GAN = 0.0
        if self.opt.lambda_NCE > 0.0:
            self.loss_NCE = self.calculate_NCE_loss(self.real_A, self.fake_B)
        else:
            self.loss_NCE, self.loss_NCE_bd = 0.0, 0.0
        if self.opt.nce_idt and self.opt.lambda_NCE > 0.0:
            self.loss_NCE_Y = self.calculate_NCE_loss(self.real_B, self.idt_B)
            loss_NCE_both = (self.loss_NCE + self.loss_NCE_Y) * 0.5
        else:
            loss_NCE_both = self.loss_NCE
        self.loss_G = self.loss_G_GAN + loss_NCE_both
        return self.loss_G

    # -- registration_model.py:237-253
    def calculate_NCE_loss(self, src, tgt):
        n_layers = len(self.nce_layers)
        feat_q = self.netG(tgt, self.nce_layers, encode_only=True)
        if self.opt.flip_equivariance and self.flipped_for_equivariance:     # registration_model.py:241-242
            feat_q = [torch.flip(fq, [3]) for fq in feat_q]
        with torch.no_grad():  # feat_k is detached inside PatchNCELoss (patchnce.py:17): forward only
            feat_k = self._encode_keys(src)
            pinned = None
            if getattr(self, 'patch_id_source', None) is not None:      # ids pinned by the caller (tests, replays of a log)
                sets = self._patch_id_sets([f.shape[2] * f.shape[3] for f in feat_k], 1, self.opt.num_patches, feat_k[0].device)
                pinned = [sets[l][0] for l in range(len(feat_k))]
            feat_k_pool, sample_ids = self.netF(feat_k, self.opt.num_patches, pinned)
        feat_q_pool, _ = self.netF(feat_q, self.opt.num_patches, sample_ids)
        total_nce_loss = 0.0
        for f_q, f_k, crit, nce_layer in zip(feat_q_pool, feat_k_pool, self.criterionNCE, self.nce_layers):
            loss = crit(f_q, f_k)                      # [B*P], reduction='none'
            total_nce_loss += ops.mean(loss) * self.opt.lambda_NCE
        return total_nce_loss / n_layers

    def calculate_NCE_losses_stacked(self, terms, tgt=None):
        """calculate_NCE_loss(real_A, fake_B), (real_B, idt_B), (real_B, regA) with ONE query-side encoder pass over
        cat(fake (= [fake_B; idt_B]), regA).  terms = ((src, tgt or None), ...): the first two targets are the halves
        of self.fake.

        Default: everything per layer happens once for the three terms -- one launch draws the 15 patch-id sets on
        the device, the key rows are gathered from the tapped features of all three sources at once, the MLP runs
        once over all key rows and once over all query rows, one PatchNCE launch per layer covers the three terms and
        one reduction yields the three losses.  When `netF.forward` has been replaced on the instance (tests pin the
        ids by wrapping it with the reference's signature) the key side goes through it term by term, in the
        reference's order, as before."""
        T = len(terms)
        n_layers = len(self.nce_layers)
        if tgt is None:                                     # (the staged step hands in the stack of its detached leaves)
            tgt = ops.cat_batch(self.fake, terms[2][1]) if T == 3 else ops.cat_batch(terms[0][1], terms[1][1])
        feat_q = self.netG(tgt, self.nce_layers, encode_only=True)
        if self.opt.flip_equivariance and self.flipped_for_equivariance:     # registration_model.py:241-242, every term
            feat_q = [torch.flip(fq, [3]) for fq in feat_q]
        sizes = [f.shape[2] * f.shape[3] for f in feat_q]
        P = self.opt.num_patches
        per_term_groups = 1 if self.opt.nce_includes_all_negatives_from_minibatch else self.opt.batch_size
        # opt.nce_sequential_keys (build-defined, default False): the key side term by term through netF.forward, in the
        # reference's call order with host-drawn ids (torch.randperm).  A netF.forward replaced on the instance (how older
        # tests pin ids) implies it; `model.patch_id_source` is the hook that pins ids on the default path.
        sequential = (getattr(self.opt, 'nce_sequential_keys', False) or 'forward' in vars(self.netF)
                      or ops._env_on('DFMIR_NCE_SEQUENTIAL_KEYS'))
        if not sequential and self.opt.netF == 'mlp_sample' and all(S >= 2 * P or P <= S <= 4096 for S in sizes):
            id_src = getattr(self, 'patch_id_source', None)
            # a captured step may hold the ids only by device address: the device generator, or a source that declares
            # `graph_safe` (same tensor every call, refreshed by its owner between steps)
            self._nce_on_device = id_src is None or bool(getattr(id_src, 'graph_safe', False))
            ids = self._patch_id_sets(sizes, T, P, feat_q[0].device)           # L x [T, P]
            with torch.no_grad():
                keys = {}
                for src, _ in terms:
                    if id(src) not in keys:
                        keys[id(src)] = self._encode_keys(src)
                k_cm = [self.netF.sample_project_multi(l, [keys[id(src)][l] for src, _ in terms], ids[l])
                        for l in range(n_layers)]
            q_cm = [self.netF.sample_project(l, feat_q[l], ids[l], T) for l in range(n_layers)]
            losses = ops.nce_terms(q_cm, k_cm, T * per_term_groups, self.opt.nce_T,
                                   self.opt.lambda_NCE / n_layers, T)
            return list(losses.unbind(0))
        self._nce_on_device = False                       # torch.randperm: host-side generator state
        pools_k, ids_t = [], []
        with torch.no_grad():
            for src, _ in terms:                          # the only random draws, in the reference's order
                pool, ids = self.netF(self._encode_keys(src), self.opt.num_patches, None)
                pools_k.append(pool)
                ids_t.append(ids)
        ids_stacked = [torch.stack([ids_t[t][l] for t in range(T)]) for l in range(n_layers)]
        for l in range(n_layers):
            if all(getattr(ids_t[t][l], "_df_distinct", None) is not None for t in range(T)):
                ops.mark_distinct(ids_stacked[l])
        fq_pool, _ = self.netF(feat_q, self.opt.num_patches, ids_stacked)   # [T,P] ids: grouped sampling
        losses = []
        for t in range(T):
            total = 0.0
            for l, crit in enumerate(self.criterionNCE):
                f_q = fq_pool[l]
                rows = f_q.shape[0] // T
                total += ops.mean(crit(f_q[t * rows:(t + 1) * rows], pools_k[t][l])) * self.opt.lambda_NCE
            losses.append(total / n_layers)
        return losses

    def _patch_id_sets(self, sizes, n_sets, P, device):
        """Per layer l a [n_sets, P] tensor of patch positions in [0, sizes[l]).  Default: one device launch draws
        them all (ops.draw_patch_ids; seeded by ops.seed_patch_ids, NOT by torch.manual_seed after first use).
        `self.patch_id_source(sizes, n_sets, P)` -> [L, n_sets, P] (or a list of L tensors) overrides the draw: tests
        and replays of a recorded run pin ids with it.  Attributes of the source: `distinct` (every set is a P-subset:
        the scatter of the backward may skip atomics; otherwise checked / accumulated safely), `graph_safe`."""
        src = getattr(self, 'patch_id_source', None)
        if src is None:
            ids = ops.draw_patch_ids(sizes, n_sets, P, device)                # P-subsets by construction
            return [ops.mark_distinct(ids[l]) for l in range(len(sizes))]
        ids = src(sizes, n_sets, P)
        out = [ids[l].to(device) for l in range(len(sizes))]
        if getattr(src, 'distinct', False):
            out = [ops.mark_distinct(t) for t in out]
        return out

    # -- registration_model.py:255-263
    def calculate_L1_loss(self, src, tgt, mask):
        if mask is None:
            return ops.masked_l1(src, tgt, torch.ones_like(src, dtype=torch.bool))
        if isinstance(mask, str):  # fused (src > -0.95) | (tgt > -0.95)
            return ops.masked_l1(src, tgt, None, -0.95)
        return ops.masked_l1(src, tgt, mask)
