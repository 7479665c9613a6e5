"""ORACLE — test infrastructure, not product code.

CPU restatement of the reference frame-importance scorer (``SimNet.forward`` and
everything below it, reference ``src/model/simnet.py``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file;
the product path (``video-summarization_amd/``) never does and fails loudly when its HIP
library is missing.

Pinning: the reference ships no tests or golden vectors for this path, so this
restatement is pinned by outputs of the reference itself, imported on CPU in the build
container by ``tests/golden/make_golden.py``; the resulting vectors are committed under
``tests/golden/`` and checked by ``tests/test_oracle.py``.  Arithmetic is torch/ATen fp32 on
CPU — the same third-party arithmetic the reference uses (unpinned there; torch 2.10.0 here).

The op sequence mirrors the reference one-for-one (including the materialised [B,H,T,T]
logits tensor and the separate scale multiply) so that timing it is a fair "port" CPU baseline.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


def _linear(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def oracle_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor,
                   mask: Optional[torch.Tensor], num_heads: int,
                   dtype: torch.dtype = torch.float32
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval-mode forward.  Returns (logits [B,T,num_classes], hidden [B,T,d]).

    sd    reference-keyed state dict
    x     [B,T,in_features]
    mask  bool [B,T], True = key is padding, or None / non-Tensor (ignored, simnet.py:38)
    """
    sd = {k: v.to(dtype) for k, v in sd.items()}
    x = x.to(dtype)
    B, T, _ = x.shape
    d = sd["embedding_layer.feature_transform.weight"].shape[0]
    H = num_heads
    dh = d // H
    # Embedding.forward simnet.py:208-217
    h = _linear(x, sd, "embedding_layer.feature_transform")                      # :211
    pe_key = "embedding_layer.positional_encoding.pos_embedding"
    if pe_key in sd:
        h = h + sd[pe_key][:, :T]                                                # :237-238 (dropout: eval identity)
    use_cls = "embedding_layer.cls_token" in sd
    if use_cls:                                                                  # :214-216: token prepended after the PE
        h = torch.cat([sd["embedding_layer.cls_token"].expand(B, 1, d), h], dim=1)
        T = T + 1
    # SimNet.process_mask simnet.py:47-56
    kmask = None
    if isinstance(mask, torch.Tensor):
        if use_cls:                                                              # :48-51 (the token is never padding)
            mask = torch.cat([torch.zeros((B, 1), dtype=mask.dtype), mask], dim=1)
        kmask = mask.view(B, 1, 1, T).expand(B, H, T, T)
    scale = d ** -0.5                                                            # :126  (d_model, NOT head_dim)
    L = 0
    while ("encoder.module_list.%d.sa.q.weight" % L) in sd:
        L += 1
    for l in range(L):                                                           # Encoder.forward :78
        p = "encoder.module_list.%d." % l
        # MultiAttentionNetwork.forward :138-164
        q = _linear(h, sd, p + "sa.q").view(B, T, H, dh).permute(0, 2, 1, 3)     # :148
        k = _linear(h, sd, p + "sa.k").view(B, T, H, dh).permute(0, 2, 1, 3)     # :150
        v = _linear(h, sd, p + "sa.v").view(B, T, H, dh).permute(0, 2, 1, 3)     # :152
        s = torch.matmul(q, k.transpose(2, 3)) * scale                           # :155
        if kmask is not None:
            s = s.masked_fill(kmask, float("-inf"))                              # :157
        w = F.softmax(s, dim=3)                                                  # :158
        o = torch.matmul(w, v).permute(0, 2, 1, 3).contiguous().view(B, T, d)    # :160-161
        o = _linear(o, sd, p + "sa.feature_projection")                          # :163
        # EncoderBlock.forward :105-114  (post-LN)
        h = F.layer_norm(o + h, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)   # :107
        m = _linear(F.relu(_linear(h, sd, p + "mlp.fc1")), sd, p + "mlp.fc2")    # :181-182
        h = F.layer_norm(m + h, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)   # :110
    logits = _linear(h, sd, "final_layer")                                       # simnet.py:42
    return logits, h


def oracle_scores(sd, x, mask, num_heads):
    """Caller-side head of val_step (reference train.py:143-144): sigmoid of the logits, [B,T]."""
    logits, _ = oracle_forward(sd, x, mask, num_heads)
    return torch.sigmoid(logits.squeeze(-1))


def usable_cpus() -> int:
    """CPUs this process may really use: min(affinity, cgroup quota, cpu_count)."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _time_once(sd, num_heads, x, iters):
    import time
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(iters):
            oracle_forward(sd, x, None, num_heads)
        return time.perf_counter() - t0


def time_cpu_baseline(sd, num_heads: int, B: int, T: int, seconds: float = 12.0, seed: int = 1234, iters: int = 0):
    """Times the restatement on host cores.  The thread count is chosen by a short sweep (the most
    favourable to the CPU wins) because a GPU box exposes far more logical CPUs than the share a
    job may use.  ``iters`` > 0 fixes the iteration count (B=1,T=320 x20); otherwise it is
    chosen to fill ``seconds``.  Returns (frames_per_s, threads_used, sample_desc)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, sd["embedding_layer.feature_transform.weight"].shape[1], generator=g)
    cap = usable_cpus()
    cands = sorted({c for c in (4, 8, 16, 32, 64, min(cap, 64)) if c <= cap}) or [1]
    best_n, best_t = cands[0], float("inf")
    for n in cands:
        torch.set_num_threads(n)
        _time_once(sd, num_heads, x[:1], 1)                 # warm the pool
        t = _time_once(sd, num_heads, x, 1)
        if t < best_t:
            best_n, best_t = n, t
    torch.set_num_threads(best_n)
    iters = iters if iters > 0 else max(2, min(60, int(round(seconds / best_t))))
    dt = _time_once(sd, num_heads, x, iters)
    return B * T * iters / dt, best_n, "B=%d,T=%d x%d iters, %d threads (best of %s)" % (B, T, iters, best_n, cands)
