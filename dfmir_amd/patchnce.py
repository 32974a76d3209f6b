"""PatchNCELoss with the reference's interface (models/patchnce.py:6-55), computed by one fused
HIP kernel (logits + diagonal fill + softmax cross-entropy); gradient flows to feat_q only."""
import torch
from torch import nn

from . import ops
from .networks import _as_channel_major


class PatchNCELoss(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.mask_dtype = torch.bool

    def forward(self, feat_q, feat_k):
        """feat_q, feat_k: [B*P, dim] -> per-row loss [B*P] (reduction='none', patchnce.py:52-55)."""
        feat_k = feat_k.detach()
        if self.opt.nce_includes_all_negatives_from_minibatch:
            groups = 1
        else:
            groups = self.opt.batch_size  # sic: the option, not the tensor's batch (patchnce.py:36)
        return ops.patchnce_rows(_as_channel_major(feat_q), _as_channel_major(feat_k), groups, self.opt.nce_T)
