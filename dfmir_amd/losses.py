"""Registration losses with the reference's interfaces: `smooothing_loss`
(models/registration_model.py:25-32), `NCC_Loss` / `Grad_Loss` (util/losses.py:81-261) and vxm `NCC` / `Grad`
(models/voxelmorph/torchvoxelmorph/losses.py:7-67,93-117; also reachable as `dfmir_amd.voxelmorph.losses`), each one
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
    """util/losses.py:81-130: mean over axes of mean(|forward difference|) ('l1', i.e. any penalty other than 'l2' in the
    reference) or of its square ('l2'); `mask=` multiplies the field first (:119-121); `loss_mult` scales the result."""

    def __init__(self, dim=2, penalty='l2', name=None, loss_mult=None, *args, **kwargs):
        super().__init__(name=name or 'gradient')
        assert dim in [2, 3]
        self.dim, self.penalty, self.loss_mult = dim, penalty, loss_mult

    def forward(self, prediction, *args, **kwargs):
        if prediction.dim() - 2 != self.dim:
            raise ValueError("Grad_Loss(dim=%d) got a %d-D field" % (self.dim, prediction.dim() - 2))
        if 'mask' in kwargs:
            m = kwargs['mask'].to(device=prediction.device, dtype=torch.float32).expand_as(prediction).contiguous()
            prediction = ops.mul(prediction, m)
        loss = ops.flow_smoothness(prediction, 'l2' if self.penalty == 'l2' else 'l1')
        if self.loss_mult is not None:
            loss = ops.scale(loss.view(1), self.loss_mult).view(())
        return loss


class NCC_Loss(_Loss):
    """util/losses.py:132-261 with the 'mean' kernel: -sqrt(mean(cc)) over a win^nd window; with `mask`
    -sqrt(sum(cc * mask) / sum(mask)) (:257-261; an empty mask gives 0 -- as a device scalar, the reference returns
    `torch.tensor(0)` after a host sync).  'gaussian' / 'linear' kernels are not on the path."""

    def __init__(self, device, kernel_var=None, name=None, kernel_type='mean', eps=1e-5, *args, **kwargs):
        super().__init__(name=name or 'ncc')
        assert kernel_type in ['mean', 'gaussian', 'linear']
        if kernel_type != 'mean':
            raise NotImplementedError("only the 'mean' NCC kernel is on the path")
        self.device, self.kernel_var, self.kernel_type, self.eps = device, kernel_var, kernel_type, eps

    def forward(self, prediction, target, mask=None, *args, **kwargs):
        nd = prediction.dim() - 2
        kv = self.kernel_var if self.kernel_var is not None else [9] * nd
        if len(set(kv)) != 1 or len(kv) != nd:
            raise NotImplementedError("NCC window must be cubic and match the tensor rank")
        return ops.ncc_loss(prediction, target, int(kv[0]), self.eps, mask=mask)


class NCC(object):
    """vxm `NCC(win).loss(y_true, y_pred)` = -mean(cc) (models/voxelmorph/torchvoxelmorph/losses.py:7-67; eps 1e-5; the
    reference builds its filter with `.to("cuda")`, here the tensors' own device).  cc is symmetric in its arguments;
    the gradient goes to y_pred."""

    def __init__(self, win=None):
        self.win = win

    def loss(self, y_true, y_pred):
        nd = y_true.dim() - 2
        win = [9] * nd if self.win is None else list(self.win)
        if len(set(win)) != 1 or len(win) != nd:
            raise NotImplementedError("NCC window must be cubic and match the tensor rank")
        return ops.ncc_loss(y_pred, y_true, int(win[0]), 1e-5, reduction='neg_mean')


class Grad(object):
    """vxm `Grad(penalty, loss_mult).loss(_, y_pred)` (models/voxelmorph/torchvoxelmorph/losses.py:93-117): 3-D fields only
    (the reference indexes five axes); penalty 'l1' (default) or 'l2'."""

    def __init__(self, penalty='l1', loss_mult=None):
        self.penalty = penalty
        self.loss_mult = loss_mult

    def loss(self, _, y_pred):
        if y_pred.dim() != 5:
            raise IndexError("vxm Grad.loss indexes [B, C, D, H, W] fields (got %d dims)" % y_pred.dim())
        grad = ops.flow_smoothness(y_pred, 'l2' if self.penalty == 'l2' else 'l1')
        if self.loss_mult is not None:
            grad = ops.scale(grad.view(1), self.loss_mult).view(())
        return grad
