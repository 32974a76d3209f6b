"""Flat-arena Adam: the parameters of one network live in ONE contiguous HBM buffer (params,
grads, exp_avg, exp_avg_sq), updated by a single fused kernel launch and all-reduced under DDP as a
single RCCL collective.  Drop-in for `torch.optim.Adam(net.parameters(), lr, betas)` as used at
reference models/registration_model.py:114-117,135 (same update rule, eps=1e-8, no weight decay)."""
import torch

from . import ops


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        if len(params) == 0:
            raise ValueError("optimizer got an empty parameter list")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam keeps one flat arena = one param group")
        self._steps = 0
        self.grad_scale = 1.0
        self._build()

    def _build(self):
        ps = self.param_groups[0]['params']
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        self.flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)
                off += k
        ops.bump_weights_epoch()

    def _reattach(self):
        off = 0
        for p in self.param_groups[0]['params']:
            k = p.numel()
            view = self.flat_g[off:off + k].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
            off += k

    def zero_grad(self, set_to_none=False):
        self._reattach()
        if self.flat_g.is_cuda:
            ops.zero_(self.flat_g)       # a kernel, not hipMemsetAsync (graph capture; see ops.zero_)
        else:
            self.flat_g.zero_()          # arena bookkeeping exercised on CPU by the gloo tests

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used on this path")
        self._reattach()
        g = self.param_groups[0]
        self._steps += 1
        ops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0],
                      g['betas'][1], g['eps'], self._steps, self.grad_scale)
