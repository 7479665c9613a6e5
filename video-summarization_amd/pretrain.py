"""Counterpart of the reference's ``model.PretrainModel`` (src/model/simnet_pretrain.py:12-100), the consumer of
the scorer's hidden output in ``pretrain.py``: same constructor, same parameter names
(``encoder.*`` = SimNet, ``video_transform.{weight,bias}``), same ``forward`` signature and return triple
``(distillation loss, centering loss, repelling loss)``.

``forward`` runs on the HIP kernels end to end: the encoder is the HIP training path (``simnet._TrainForward``) and
everything below it - ``video_transform``, the repelling loss, the masked score-softmax pooling, the centering penalty
and the soft cross-entropy, forward and backward - is ``_PretrainHead``, a ``torch.autograd.Function`` over
``vs_pretrain_head_forward / _backward`` (``include/vs_train.h``, kernels ``csrc/vs_pretrain_kernels.hip``).  There is
no PyTorch fallback: CPU tensors raise.

One algorithmic change.  The reference materialises the [B,T,T] cosine-similarity tensor to average its off-diagonal
(simnet_pretrain.py:56-69); the same number is

    sum_{i != j} x^_i . x^_j = || sum_i x^_i ||^2 - sum_i || x^_i ||^2        (x^ = masked, normalised rows)

which costs O(T d) instead of O(T^2 d) time and memory (T = 2000, d = 512: 16 MB of [T,T] per video avoided in
the forward and again in the backward), and differs from the reference only by fp32 rounding.  The methods
``cross_entropy_loss`` / ``entropy`` / ``repelling_loss`` are kept with the reference's signatures for callers that use
them on their own tensors; ``forward`` does not go through them.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _lib
from .simnet import SimNet


class _PretrainHead(torch.autograd.Function):
    """(hidden [B,T,d], logits [B,T,1], vid [B,F], mask, video_transform.weight, .bias) -> losses [3]."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)        # pretrain.py:59 calls under autocast
    def forward(ctx, hidden, logits, vid, mask, weight, bias, temp, entropy):
        lib = _lib.load()
        B, T, d = hidden.shape
        Fo = weight.shape[0]
        hidden, logits, vid = hidden.contiguous(), logits.contiguous(), vid.contiguous()
        w, bvec = weight.detach().contiguous(), bias.detach().contiguous()
        m = None
        if mask is not None:
            m = mask.contiguous()
            m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
        dev = hidden.device
        feats = torch.empty((B, T, Fo), dtype=torch.float32, device=dev)
        losses = torch.empty((3,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            state = torch.empty((lib.vs_pretrain_head_state_bytes(B, T, Fo),), dtype=torch.uint8, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.vs_pretrain_head_forward(hidden.data_ptr(), logits.data_ptr(), None if m is None else m.data_ptr(),
                                                    vid.data_ptr(), w.data_ptr(), bvec.data_ptr(), B, T, d, Fo, float(temp),
                                                    int(entropy), feats.data_ptr(), state.data_ptr(), losses.data_ptr(), stream))
        ctx.save_for_backward(hidden, logits, vid, m, w, feats, state)
        ctx.cfg = (float(temp), int(entropy), logits.shape)
        return losses

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_losses):
        lib = _lib.load()
        hidden, logits, vid, m, w, feats, state = ctx.saved_tensors
        temp, entropy, lshape = ctx.cfg
        B, T, d = hidden.shape
        Fo = w.shape[0]
        dev = hidden.device
        g = d_losses.contiguous().float()
        d_hidden, d_logits = torch.empty_like(hidden), torch.empty((B, T), dtype=torch.float32, device=dev)
        d_w, d_b = torch.empty_like(w), torch.empty((Fo,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = torch.empty((lib.vs_pretrain_head_workspace_bytes(B, T, d, Fo),), dtype=torch.uint8, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.vs_pretrain_head_backward(hidden.data_ptr(), logits.data_ptr(), None if m is None else m.data_ptr(),
                                                     vid.data_ptr(), w.data_ptr(), feats.data_ptr(), state.data_ptr(),
                                                     g.data_ptr(), B, T, d, Fo, temp, entropy, d_hidden.data_ptr(),
                                                     d_logits.data_ptr(), d_w.data_ptr(), d_b.data_ptr(), ws.data_ptr(),
                                                     ws.numel(), stream))
        return d_hidden, d_logits.view(lshape), None, None, d_w, d_b, None, None


class PretrainModel(nn.Module):
    def __init__(self, feature_dim: int = 256, sparsity: float = 0.0, sharpening_t=0.4, **kwargs):
        super().__init__()
        self.feature_dim = feature_dim                      # simnet_pretrain.py:24-26
        self.sparsity = sparsity
        self.sharpening_t = sharpening_t
        # the encoder never gets `sparsity` (hard-wired 0., simnet_pretrain.py:30)
        self.encoder = SimNet(sparsity=0., use_cls=False, d_model=feature_dim, **kwargs)
        self.video_transform = nn.Linear(feature_dim, 512)  # simnet_pretrain.py:33

    def cross_entropy_loss(self, x1: Tensor, x2: Tensor) -> Tensor:
        """simnet_pretrain.py:35-41: mean over ALL elements of -softmax(x2) * log softmax(x1)."""
        return (-F.softmax(x2, dim=1) * F.log_softmax(x1, dim=1)).mean()

    def entropy(self, x: Tensor, mask=None) -> Tensor:
        """simnet_pretrain.py:43-47 (x log x, masked positions zeroed, mean over frames then over the rest)."""
        x = x * torch.log(x)
        if isinstance(mask, Tensor):
            x = x.masked_fill(mask, 0.)
        return x.mean(dim=1).mean()

    def repelling_loss(self, x: Tensor, mask) -> Tensor:
        """simnet_pretrain.py:49-71: average cosine similarity between different frames (masked frames count as
        zero rows but stay in the T^2 denominator), without the [T,T] tensor - see the module docstring."""
        n_frames = x.size(1)
        if isinstance(mask, Tensor):
            x = x * (mask == False).unsqueeze(2)            # noqa: E712  (the reference's own spelling)
        x = x / (x.norm(dim=2, keepdim=True) + 1e-9)
        total = x.sum(dim=1).pow(2).sum(dim=1)              # || sum_i x^_i ||^2     [B]
        diag = x.pow(2).sum(dim=(1, 2))                     # sum_i || x^_i ||^2     [B]
        return ((total - diag) / float(n_frames * n_frames)).mean()

    def forward(self, x: Tensor, video_representation: Tensor, mask=None, visualize_attention=None,
                pen_met: str = "entropy"):
        # `visualize_attention` would make the reference unpack a 2-tuple into (out, attention) and crash a line
        # later (simnet_pretrain.py:75-78); no caller passes it, and it is ignored here.
        logits, hidden = self.encoder(x, mask, model_score=True)                                      # :77
        if not hidden.is_cuda:
            raise RuntimeError("PretrainModel runs on the MI355X HIP kernels only (HIP tensors)")
        if logits.size(2) != 1:
            raise RuntimeError("PretrainModel needs num_classes == 1 (one score per frame)")
        losses = _PretrainHead.apply(hidden, logits, video_representation.to(hidden.device), mask if isinstance(mask, Tensor) else None,
                                     self.video_transform.weight, self.video_transform.bias, self.sharpening_t,
                                     pen_met == "entropy")
        return losses[0], losses[1], losses[2]
