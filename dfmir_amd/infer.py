"""Inference path (reference test.py:37-90): translate B, register A onto B, and
warp a label map with nearest-neighbour sampling -- all on the HIP device (the reference moves the label
warp to the CPU, test.py:80-81)."""
import torch

from .voxelmorph import SpatialTransformer


@torch.no_grad()
def register_pair(model, data, label=None):
    """Returns dict(fake_B, idt_B, translated_B, warped_A, flow[, warped_label]).

    model: a REGISTRATIONModel (train or test mode); data: {'A','B','A_paths','B_paths'};
    label: optional [B,1,H,W] tensor warped with mode='nearest' by the same flow (test.py:80-81)."""
    model.set_input(data)
    model.forward()                                        # model.test() without the visuals hook
    translated_B = model.netG(model.real_B)                # test.py:77
    warped_A, flow = model.netR(model.real_A, model.real_B, registration=True)   # test.py:78
    out = dict(fake_B=model.fake_B, idt_B=model.idt_B, translated_B=translated_B, warped_A=warped_A, flow=flow)
    if label is not None:
        st = SpatialTransformer(tuple(flow.shape[2:]), mode='nearest').to(flow.device)
        out['warped_label'] = st(label.to(flow.device).float(), flow)
    return out
