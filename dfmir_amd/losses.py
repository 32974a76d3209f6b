"""Registration losses with the reference's interfaces: `smooothing_loss`
(models/registration_model.py:25-32), `NCC_Loss` / `Grad_Loss` (util/losses.py:81-261), each one
fused HIP reduction (dfmir_amd.ops)."""
import torch

from . import ops


def smooothing_loss(y_pred):
    """(mean(dx^2) + mean(dy^2)) / 2 of a 2-D flow [B,2,H,W] (sic, three o's)."""
    return ops.flow_smoothness(y_pred)


class _Loss(object):
    def __init__(self, name=None, *args, **kwargs):
        self.name = name

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)


class Grad_Loss(_Loss):
    """util/losses.py:81-130 ('l2' penalty only on this path)."""

    def __init__(self, dim=2, penalty='l2', name=None, loss_mult=None, *args, **kwargs):
        super().__init__(name=name or 'gradient')
        assert dim in [2, 3]
        if penalty != 'l2':
            raise NotImplementedError("only the l2 gradient penalty is on the path")
        self.dim, self.penalty, self.loss_mult = dim, penalty, loss_mult

    def forward(self, prediction, *args, **kwargs):
        if 'mask' in kwargs:
            raise NotImplementedError("masked Grad_Loss is not on the path")
        if prediction.dim() - 2 != self.dim:
            raise ValueError("Grad_Loss(dim=%d) got a %d-D field" % (self.dim, prediction.dim() - 2))
        loss = ops.flow_smoothness(prediction)
        if self.loss_mult is not None:
            loss = ops.scale(loss.view(1), self.loss_mult).view(())
        return loss


class NCC_Loss(_Loss):
    """util/losses.py:132-261 with the 'mean' kernel: -sqrt(mean(cc)) over a win^nd window."""

    def __init__(self, device, kernel_var=None, name=None, kernel_type='mean', eps=1e-5, *args, **kwargs):
        super().__init__(name=name or 'ncc')
        if kernel_type != 'mean':
            raise NotImplementedError("only the 'mean' NCC kernel is on the path")
        self.device, self.kernel_var, self.kernel_type, self.eps = device, kernel_var, kernel_type, eps

    def forward(self, prediction, target, mask=None, *args, **kwargs):
        if mask is not None:
            raise NotImplementedError("masked NCC is not on the path")
        nd = prediction.dim() - 2
        kv = self.kernel_var if self.kernel_var is not None else [9] * nd
        if len(set(kv)) != 1 or len(kv) != nd:
            raise NotImplementedError("NCC window must be cubic and match the tensor rank")
        return ops.ncc_loss(prediction, target, int(kv[0]), self.eps)
