"""``mse_with_mask_loss`` — drop-in for the reference training loss, on the HIP kernels.

Same signature and value as reference ``src/utils/utils.py:45-56`` (called at ``train.py:122``):
``((output.squeeze(2) - targets) * scale) ** 2`` with ``scale = 0`` on masked frames, averaged (``reduction="avg"``)
or summed over ALL ``B*T`` entries.  Forward (two-stage fixed-order reduction) and backward are kernels of
``libvsscore.so`` (``include/vs_train.h``: ``vs_mse_mask_loss_forward`` / ``_backward``); HIP tensors only.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


class _MseMask(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, output: Tensor, targets: Tensor, mask, mean: bool):
        lib = _lib.load()
        o, t = output.contiguous(), targets.contiguous().float()
        if o.numel() != t.numel():
            raise RuntimeError("output %s and targets %s differ in size" % (tuple(output.shape), tuple(targets.shape)))
        m = None
        if mask is not None:
            m = mask.contiguous()
            m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
            if m.numel() != o.numel():
                raise RuntimeError("mask %s does not match output %s" % (tuple(mask.shape), tuple(output.shape)))
        loss = torch.empty((), dtype=torch.float32, device=o.device)
        with torch.cuda.device(o.device):
            scratch = torch.empty((256,), dtype=torch.float32, device=o.device)
            stream = torch.cuda.current_stream(o.device).cuda_stream
            _lib.check(lib.vs_mse_mask_loss_forward(o.data_ptr(), t.data_ptr(), None if m is None else m.data_ptr(),
                                                    o.numel(), int(mean), scratch.data_ptr(), loss.data_ptr(), stream))
        ctx.save_for_backward(o, t, m)
        ctx.mean, ctx.shape = bool(mean), output.shape
        return loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_loss):
        lib = _lib.load()
        o, t, m = ctx.saved_tensors
        d_out = torch.empty_like(o)
        g = d_loss.contiguous().float()
        with torch.cuda.device(o.device):
            stream = torch.cuda.current_stream(o.device).cuda_stream
            _lib.check(lib.vs_mse_mask_loss_backward(o.data_ptr(), t.data_ptr(), None if m is None else m.data_ptr(),
                                                     g.data_ptr(), o.numel(), int(ctx.mean), d_out.data_ptr(), stream))
        return d_out.view(ctx.shape), None, None, None


def mse_with_mask_loss(output: Tensor, targets: Tensor, mask, reduction: str = "avg") -> Tensor:
    """output [B,T,1] (the scorer's logits), targets [B,T], mask bool [B,T] (True = padded frame) -> scalar loss."""
    if not output.is_cuda:
        raise RuntimeError("mse_with_mask_loss runs on the MI355X HIP kernels only (HIP tensors)")
    mk = mask if isinstance(mask, Tensor) else None
    return _MseMask.apply(output, targets, mk, reduction == "avg")
