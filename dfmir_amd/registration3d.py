"""3-D registration step for BASELINE configs 4/5.

The reference has no 3-D entry point (`vol_shape = (crop_size, crop_size)` is hard-wired,
models/registration_model.py:97); SURVEY.md section 8 row A13 defines the composition from the pieces the
reference does ship: `VxmDense(ndims=3, int_steps=7, bidir=True)`
(models/voxelmorph/torchvoxelmorph/networks.py:1028-1145) + `NCC_Loss(kernel_var=[9,9,9], 'mean')`
(util/losses.py:132-261) + lambda * `Grad_Loss(dim=3, 'l2')` (util/losses.py:81-130), Adam(2e-4,
(0.5, 0.999)).  Oracle counterpart: oracle/dfmir_oracle.py::Registration3DStep.
"""
import torch

from . import distributed as dfdist
from . import ops
from .losses import Grad_Loss, NCC_Loss
from .optim import FlatAdam
from .voxelmorph import VxmDense


class Registration3DModel(object):
    def __init__(self, shape, features=None, lam=1.0, lr=2e-4, betas=(0.5, 0.999), win=9, device="cuda",
                 capture_step=False, deterministic_wgrad=None):
        """capture_step (build-defined, as REGISTRATIONModel's opt.capture_step): after two eager steps forward +
        losses + backward are captured into ONE hipGraph and replayed; Adam and the gradient all-reduce stay eager.
        Small volumes are host-bound otherwise (128^3: 3.7 ms of Python / autograd / ctypes per 5.2 ms step)."""
        self.device = torch.device(device)
        if deterministic_wgrad is not None:     # (process-global switch of dfmir_amd.ops, as REGISTRATIONModel's opt.deterministic_wgrad)
            ops.set_deterministic_wgrad(deterministic_wgrad)
        self.netR = VxmDense(tuple(shape), features, int_steps=7, bidir=True).to(self.device)
        self.netR.skip_unused_target = True      # the step reads (y_source, flow) only
        self.optimizer_R = FlatAdam(self.netR.parameters(), lr=lr, betas=betas)
        self.criterionNCC = NCC_Loss(self.device, kernel_var=[win] * len(shape), kernel_type='mean')
        self.criterionGrad = Grad_Loss(dim=len(shape), penalty='l2')
        self.lam = lam
        self._ddp = False
        self.capture_step = bool(capture_step) and self.device.type == "cuda"
        self._graph = {'eager_steps': 0, 'graph': None, 'shape': None, 'force_eager': False,
                       'stream': torch.cuda.Stream(device=self.device) if self.capture_step else None}

    def parallelize(self):
        self._ddp = dfdist.is_distributed()
        if self._ddp:
            dfdist.broadcast_arena(self.optimizer_R.flat_p, src=0)
            self.optimizer_R.grad_scale = 1.0 / dfdist.world_size()

    def set_input(self, data):
        self.real_A = data['A'].to(self.device, non_blocking=True)
        self.real_B = data['B'].to(self.device, non_blocking=True)

    def _forward_backward(self):
        y_source, y_target, flow = self.netR(self.real_A, self.real_B)
        self.regA, self.flow = y_source, flow
        self.optimizer_R.zero_grad()
        self.loss_ncc = self.criterionNCC(y_source, self.real_B)
        self.loss_grad = self.criterionGrad(flow)
        with ops.deferred_weight_grads():
            (self.loss_ncc + self.loss_grad * self.lam).backward()

    def _apply_updates(self):
        if self._ddp:
            dfdist.allreduce_arenas([self.optimizer_R.flat_g])
        self.optimizer_R.step()

    def optimize_parameters(self):
        if not self.capture_step:
            self._forward_backward()
            return self._apply_updates()
        # The protocol of REGISTRATIONModel._optimize_parameters_graphed: every step of a capturing model runs on ONE
        # side stream (autograd binds a parameter's AccumulateGrad node to the stream of its first use), inputs live in
        # static buffers, the outputs are the capture's tensors; a change of shape or arena re-captures.
        st = self._graph
        side, cur = st['stream'], torch.cuda.current_stream()
        shape = (tuple(self.real_A.shape), tuple(self.real_B.shape), self.optimizer_R.flat_p.data_ptr(),
                 self.optimizer_R.flat_g.data_ptr())
        if st['graph'] is not None and st['shape'] != shape:
            st.update(graph=None, eager_steps=0)
        if st['force_eager'] or (st['graph'] is None and st['eager_steps'] < 2):
            st['eager_steps'] += 1
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._forward_backward()
            cur.wait_stream(side)
            return self._apply_updates()
        if st['graph'] is None:
            st['in_A'], st['in_B'] = self.real_A.clone(), self.real_B.clone()
            self.real_A, self.real_B = st['in_A'], st['in_B']
            for k in ('regA', 'flow', 'loss_ncc', 'loss_grad'):          # drop the previous step's autograd graph
                v = getattr(self, k, None)
                if torch.is_tensor(v) and v.grad_fn is not None:
                    setattr(self, k, v.detach())
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            ops.begin_graph_capture()
            try:
                with torch.cuda.graph(graph, stream=side):
                    self._forward_backward()
            finally:
                ops.end_graph_capture()
            st['outputs'] = {k: getattr(self, k) for k in ('regA', 'flow', 'loss_ncc', 'loss_grad')}
            st.update(graph=graph, shape=shape)
        else:
            if self.real_A is not st['in_A']:
                st['in_A'].copy_(self.real_A, non_blocking=True)
                st['in_B'].copy_(self.real_B, non_blocking=True)
            vars(self).update(st['outputs'])
            self.real_A, self.real_B = st['in_A'], st['in_B']
        st['graph'].replay()
        self._apply_updates()

    def get_current_losses(self):
        return dict(ncc=float(self.loss_ncc.detach()), grad=float(self.loss_grad.detach()))
