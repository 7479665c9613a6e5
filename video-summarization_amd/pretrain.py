"""Counterpart of the reference's ``model.PretrainModel`` (src/model/simnet_pretrain.py:12-100), the consumer of
the scorer's hidden output in ``pretrain.py``: same constructor, same parameter names
(``encoder.*`` = SimNet, ``video_transform.{weight,bias}``), same ``forward`` signature and return triple
``(distillation loss, centering loss, repelling loss)``.

It is a TRAINING loss head: every call site runs it under autograd (pretrain.py:61-66), so it is composed from
torch ops on the module's device (like ``SimNet._forward_autograd``) rather than hand-written kernels - with one
algorithmic change.  The reference materialises the [B,T,T] cosine-similarity tensor to average its off-diagonal
(simnet_pretrain.py:56-69); the same number is

    sum_{i != j} x^_i . x^_j = || sum_i x^_i ||^2 - sum_i || x^_i ||^2        (x^ = masked, normalised rows)

which costs O(T d) instead of O(T^2 d) time and memory (T = 2000, d = 512: 16 MB of [T,T] per video avoided in
the forward and again in the backward), and differs from the reference only by fp32 rounding.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .simnet import SimNet


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
        logits, hidden = self.encoder(x, mask, model_score=True)
        feats = self.video_transform(hidden)                               # [B, T, 512]
        repel = self.repelling_loss(feats, mask)
        key_mask = mask.unsqueeze(2) if isinstance(mask, Tensor) else None
        if key_mask is not None:
            logits = logits.masked_fill(key_mask, float("-inf"))
        weights = F.softmax(logits / self.sharpening_t, dim=1)             # mixture over the frames  [B, T, 1]
        if pen_met == "entropy":
            center = self.entropy(weights + 1e-9, key_mask)               # 1e-9: the reference's stabiliser
        else:
            center = torch.norm(weights, dim=1).mean()
        pooled = torch.matmul(weights.transpose(1, 2), feats).squeeze(1)   # score-weighted video representation
        return self.cross_entropy_loss(pooled, video_representation), center, repel
