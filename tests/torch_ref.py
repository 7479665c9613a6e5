"""Test infrastructure (not product code): a plain-torch, autograd-capable restatement of ``SimNet.forward`` in TRAIN
mode (reference src/model/simnet.py:32-45, 105-114, 138-164, 180-183) that takes every dropout mask as an explicit
argument.  The HIP training path cannot share torch's RNG stream, so its tests read the keep masks the library draws
(``vs_train_dropout_mask_*``), run this model in float64 on the CPU with exactly those masks and compare values and
gradients.  With ``masks=None`` it is the dropout-free train-mode forward."""
import math

import torch
import torch.nn.functional as F


def _drop(t, keep, p):
    if keep is None or p <= 0.0:
        return t
    return t * keep.to(t.dtype) / (1.0 - p)


def forward_with_masks(params, x, mask, num_heads, p=0.0, p_embed=0.0, masks=None, stats=None, gates=None):
    """params: dict name -> tensor (reference state_dict names, any float dtype, may require grad).
    masks: dict with 'embed' [B,T,d] and per layer l 'attn%d' [B,H,T,T], 'drop1_%d' [B,T,d], 'mlp%d' [B,T,4d],
    'drop2_%d' [B,T,d] (bool/uint8, 1 = kept).  Returns (logits [B,T,nc], hidden [B,T,d]).
    stats (optional dict): receives 'min_abs_fc1' = the smallest |fc1 pre-activation| of the run - a ReLU whose input
    is within fp32 rounding of zero may legitimately switch the other way in an fp32 implementation.
    gates (optional dict 'gate%d' -> bool [B,T,4d]): the implementation's own ReLU-and-dropout gate of layer l (the sign
    pattern of its saved MLP activation); when given it replaces relu + mlp dropout, so checker and implementation
    differentiate the SAME piecewise-linear function even where a ReLU input is within rounding of zero."""
    masks = masks or {}
    B, T, _ = x.shape
    d = params["embedding_layer.feature_transform.weight"].shape[0]
    H, dh = num_heads, d // num_heads
    h = F.linear(x, params["embedding_layer.feature_transform.weight"], params["embedding_layer.feature_transform.bias"])
    pe = params.get("embedding_layer.positional_encoding.pos_embedding")
    if pe is not None:
        h = _drop(h + pe[:, :T].to(h.dtype), masks.get("embed"), p_embed)
    L = 0
    while "encoder.module_list.%d.sa.q.weight" % L in params:
        L += 1
    scale = d ** -0.5                                                                  # simnet.py:126
    for l in range(L):
        pre = "encoder.module_list.%d." % l
        lin = lambda t, n: F.linear(t, params[pre + n + ".weight"], params[pre + n + ".bias"])     # noqa: E731
        q = lin(h, "sa.q").view(B, T, H, dh).permute(0, 2, 1, 3)
        k = lin(h, "sa.k").view(B, T, H, dh).permute(0, 2, 1, 3)
        v = lin(h, "sa.v").view(B, T, H, dh).permute(0, 2, 1, 3)
        s = torch.matmul(q, k.transpose(2, 3)) * scale
        if mask is not None:
            s = s.masked_fill(mask.view(B, 1, 1, T), float("-inf"))
        w = _drop(F.softmax(s, dim=3), masks.get("attn%d" % l), p)
        o = torch.matmul(w, v).permute(0, 2, 1, 3).contiguous().view(B, T, d)
        o = lin(o, "sa.feature_projection")
        h = F.layer_norm(_drop(o, masks.get("drop1_%d" % l), p) + h, (d,), params[pre + "norm1.weight"], params[pre + "norm1.bias"], 1e-5)
        pre_act = lin(h, "mlp.fc1")
        if stats is not None:
            stats["min_abs_fc1"] = min(stats.get("min_abs_fc1", float("inf")), pre_act.detach().abs().min().item())
        if gates is not None:
            f = pre_act * gates["gate%d" % l].to(pre_act.dtype) / (1.0 - p)
        else:
            f = _drop(F.relu(pre_act), masks.get("mlp%d" % l), p)
        f = lin(f, "mlp.fc2")
        h = F.layer_norm(_drop(f, masks.get("drop2_%d" % l), p) + h, (d,), params[pre + "norm2.weight"], params[pre + "norm2.bias"], 1e-5)
    return F.linear(h, params["final_layer.weight"], params["final_layer.bias"]), h


def attention_with_mask(q, k, v, key_mask, scale, keep=None, p=0.0):
    """q,k,v [B,H,T,dh] -> out [B,T,H*dh]; keep [B,H,T,T] or None."""
    B, H, T, dh = q.shape
    s = torch.matmul(q, k.transpose(2, 3)) * scale
    if key_mask is not None:
        s = s.masked_fill(key_mask.view(B, 1, 1, T), float("-inf"))
    lse2 = torch.logsumexp(s, dim=3) / math.log(2.0)
    w = _drop(F.softmax(s, dim=3), keep, p)
    return torch.matmul(w, v).permute(0, 2, 1, 3).reshape(B, T, H * dh), lse2
