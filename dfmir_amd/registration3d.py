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
    def __init__(self, shape, features=None, lam=1.0, lr=2e-4, betas=(0.5, 0.999), win=9, device="cuda"):
        self.device = torch.device(device)
        self.netR = VxmDense(tuple(shape), features, int_steps=7, bidir=True).to(self.device)
        self.optimizer_R = FlatAdam(self.netR.parameters(), lr=lr, betas=betas)
        self.criterionNCC = NCC_Loss(self.device, kernel_var=[win] * len(shape), kernel_type='mean')
        self.criterionGrad = Grad_Loss(dim=len(shape), penalty='l2')
        self.lam = lam
        self._ddp = False

    def parallelize(self):
        self._ddp = dfdist.is_distributed()
        if self._ddp:
            dfdist.broadcast_arena(self.optimizer_R.flat_p, src=0)
            self.optimizer_R.grad_scale = 1.0 / dfdist.world_size()

    def set_input(self, data):
        self.real_A = data['A'].to(self.device, non_blocking=True)
        self.real_B = data['B'].to(self.device, non_blocking=True)

    def optimize_parameters(self):
        y_source, y_target, flow = self.netR(self.real_A, self.real_B)
        self.regA, self.flow = y_source, flow
        self.optimizer_R.zero_grad()
        self.loss_ncc = self.criterionNCC(y_source, self.real_B)
        self.loss_grad = self.criterionGrad(flow)
        with ops.deferred_weight_grads():
            (self.loss_ncc + self.loss_grad * self.lam).backward()
        if self._ddp:
            dfdist.allreduce_arenas([self.optimizer_R.flat_g])
        self.optimizer_R.step()

    def get_current_losses(self):
        return dict(ncc=float(self.loss_ncc), grad=float(self.loss_grad))
