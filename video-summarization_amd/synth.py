"""Deterministic synthetic weights and frame features.

No dataset (.h5) or checkpoint exists offline, so every test, golden fixture and
bench run builds its model weights and inputs from a seed.  numpy's PCG64 stream is
stable across platforms and numpy versions, which lets the golden fixtures under
``tests/golden/`` store only seeds + expected outputs instead of 13.7 MB of weights.

The key set / shapes produced here are exactly the reference ``SimNet.state_dict()``
(reference ``src/model/simnet.py:10-30``; key list in SURVEY.md §8(a) row 1), so the
same dict loads ``strict=True`` into the reference module (done in
``tests/golden/make_golden.py``) and into this package's ``SimNet``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

IN_FEATURES = 1024      # reference simnet.py:22 (hard-coded)
PE_MAX_LEN = 2000       # reference simnet.py:188 (Embedding default; SimNet.max_len is never forwarded)
PAD_VALUE = 1000.0      # reference data/dataset.py:159-160


def positional_table(d_model: int, max_len: int = PE_MAX_LEN, stable: bool = False) -> torch.Tensor:
    """Sinusoidal table [1, max_len, d_model], fp32: pe[p,2i]=sin(p*w_i), pe[p,2i+1]=cos(p*w_i),
    w_i = exp(-2i*ln(10000)/d)  (reference ``simnet.py:226-232``).

    ``stable=False`` (the module's default buffer): the reference's own torch fp32 op sequence, so on one machine
    the buffer is bit-equal to the reference's.  That sequence is NOT reproducible across machines: torch's
    vectorised fp32 ``exp`` differs in the last bit between CPUs (3 of the 128 w_i at d = 256 differ from the
    correctly rounded value in the build container), and one ulp of w_i (6e-8) times p = 2000 is a phase shift of
    1.2e-4 - the table, hence every score, then moves by up to ~1e-4 between the machine that made a golden vector
    and the machine that checks it.  Real deployments are not affected (the table is a ``state_dict`` buffer and
    travels with the checkpoint), but seeded test weights are, so ``make_state_dict`` uses
    ``stable=True``: every fp32 operation of the same formula evaluated in float64 and rounded once (correctly
    rounded exp / sin / cos of the same fp32 arguments) - identical on every IEEE machine, and within 1.3e-4 of the
    reference's buffer on any of them."""
    if stable:
        t = (-torch.arange(0, d_model, 2) * math.log(10000) / d_model).numpy()          # exact fp32 ops (mul, div)
        w = torch.from_numpy(np.exp(t.astype(np.float64)).astype(np.float32))
        arg = (torch.arange(0, max_len).reshape(max_len, 1) * w).numpy().astype(np.float64)   # fp32 product, then widened
        pe = torch.zeros((max_len, d_model))
        pe[:, 0::2] = torch.from_numpy(np.sin(arg).astype(np.float32))
        pe[:, 1::2] = torch.from_numpy(np.cos(arg).astype(np.float32))
        return pe.unsqueeze(0)
    w = torch.exp(-torch.arange(0, d_model, 2) * math.log(10000) / d_model)
    pos = torch.arange(0, max_len).reshape(max_len, 1)
    pe = torch.zeros((max_len, d_model))
    pe[:, 0::2] = torch.sin(pos * w)
    pe[:, 1::2] = torch.cos(pos * w)
    return pe.unsqueeze(0)


def state_dict_keys(d_model: int, num_layers: int, num_classes: int = 1,
                    use_pos: bool = True, in_features: int = IN_FEATURES,
                    max_len: int = PE_MAX_LEN, use_cls: bool = False):
    """(key, shape, kind) in the reference's registration order."""
    d = d_model
    out = [("embedding_layer.cls_token", (1, 1, d), "beta")] if use_cls else []      # simnet.py:205-206: first in the state_dict
    out += [("embedding_layer.feature_transform.weight", (d, in_features), "w"),
            ("embedding_layer.feature_transform.bias", (d,), "b:%d" % in_features)]
    if use_pos:
        out.append(("embedding_layer.positional_encoding.pos_embedding", (1, max_len, d), "pe"))
    for l in range(num_layers):
        p = "encoder.module_list.%d." % l
        for name in ("q", "k", "v"):
            out.append((p + "sa.%s.weight" % name, (d, d), "w"))
            out.append((p + "sa.%s.bias" % name, (d,), "b:%d" % d))
        out.append((p + "sa.feature_projection.weight", (d, d), "w"))
        out.append((p + "sa.feature_projection.bias", (d,), "b:%d" % d))
        out.append((p + "mlp.fc1.weight", (4 * d, d), "w"))
        out.append((p + "mlp.fc1.bias", (4 * d,), "b:%d" % d))
        out.append((p + "mlp.fc2.weight", (d, 4 * d), "w"))
        out.append((p + "mlp.fc2.bias", (d,), "b:%d" % (4 * d)))
        out.append((p + "norm1.weight", (d,), "g"))
        out.append((p + "norm1.bias", (d,), "beta"))
        out.append((p + "norm2.weight", (d,), "g"))
        out.append((p + "norm2.bias", (d,), "beta"))
    out.append(("final_layer.weight", (num_classes, d), "w"))
    out.append(("final_layer.bias", (num_classes,), "b:%d" % d))
    return out


def make_state_dict(d_model: int, num_layers: int, seed: int, num_classes: int = 1,
                    use_pos: bool = True, in_features: int = IN_FEATURES,
                    max_len: int = PE_MAX_LEN, trained_like: bool = True, use_cls: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded weights with nn.Linear's default distribution U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    for weights and biases.  ``trained_like`` perturbs the LayerNorm affine away from (1, 0)
    so the fused LN epilogues are really exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, torch.Tensor] = {}
    cls_token = None
    if use_cls:      # drawn from its own stream so that every other tensor equals the use_cls=False dict of the same seed
        cls_token = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 7919)).standard_normal((1, 1, d_model)).astype(np.float32))
    for key, shape, kind in state_dict_keys(d_model, num_layers, num_classes, use_pos,
                                            in_features, max_len, use_cls):
        if key == "embedding_layer.cls_token":
            sd[key] = cls_token
            continue
        if kind == "pe":
            sd[key] = positional_table(d_model, max_len, stable=True)     # machine-independent (see positional_table)
            continue
        if kind == "w":
            bound = 1.0 / math.sqrt(shape[-1])
            a = rng.uniform(-bound, bound, size=shape)
        elif kind.startswith("b:"):
            bound = 1.0 / math.sqrt(int(kind[2:]))
            a = rng.uniform(-bound, bound, size=shape)
        elif kind == "g":
            a = 1.0 + (0.1 * rng.standard_normal(size=shape) if trained_like else 0.0) * np.ones(shape)
        else:  # beta
            a = (0.1 * rng.standard_normal(size=shape) if trained_like else 0.0) * np.ones(shape)
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def make_features(B: int, T: int, seed: int, kind: str = "randn",
                  lengths: Optional[Sequence[int]] = None,
                  in_features: int = IN_FEATURES) -> torch.Tensor:
    """Frame features [B, T, in_features] fp32.  ``randn`` = N(0,1);
    ``pool5`` = non-negative, GoogLeNet-pool5-like magnitudes.  ``lengths`` right-pads each
    video with PAD_VALUE in every feature, like ``collate_fn_train`` (dataset.py:157-161)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.standard_normal(size=(B, T, in_features)).astype(np.float32)
    if kind == "pool5":
        a = np.abs(a) * 0.5
    elif kind != "randn":
        raise ValueError(kind)
    if lengths is not None:
        assert len(lengths) == B
        for b, n in enumerate(lengths):
            a[b, n:, :] = PAD_VALUE
    return torch.from_numpy(a)


def padding_mask(x: torch.Tensor) -> torch.Tensor:
    """Caller-side mask construction of reference train.py:118 (True = padded frame)."""
    return x[:, :, 0] == PAD_VALUE


def random_mask(B: int, T: int, seed: int, p: float = 0.3) -> torch.Tensor:
    """Arbitrary (non-suffix) bool key mask; never masks key 0 so no row is fully masked."""
    rng = np.random.Generator(np.random.PCG64(seed))
    m = rng.random(size=(B, T)) < p
    m[:, 0] = False
    return torch.from_numpy(m)
