#!/usr/bin/env python3
"""Generates the golden vectors in this directory by IMPORTING the reference on CPU.

Run in the build container only (the reference tree is not available on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

For every case in CASES it builds seeded weights/inputs with ``synth.py`` (this repo),
loads them ``strict=True`` into the reference ``model.SimNet`` (reference
``src/model/simnet.py:8``), runs ``forward`` in eval mode on CPU fp32 and stores the raw
logits plus a strided sample of the hidden state.  Fixtures hold data only: seeds, shapes and
expected outputs — never reference source.
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("VS_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True

synth = importlib.import_module("video-summarization_amd.synth")

# name, model cfg, input cfg
CASES = [
    dict(name="ma_t320", H=4, d=256, L=4, B=1, T=320, wseed=11, xseed=101, kind="randn"),
    dict(name="ma_t320_pad", H=4, d=256, L=4, B=2, T=320, wseed=11, xseed=102, kind="pool5",
         lengths=[320, 211]),
    dict(name="mb_t320", H=4, d=512, L=3, B=1, T=320, wseed=12, xseed=103, kind="randn"),
    dict(name="ma_t1024_b2", H=4, d=256, L=4, B=2, T=1024, wseed=11, xseed=104, kind="randn"),
    dict(name="ma_ragged37", H=4, d=256, L=4, B=3, T=37, wseed=13, xseed=105, kind="randn",
         lengths=[37, 20, 5]),
    dict(name="ctor_default_t100", H=8, d=512, L=4, B=1, T=100, wseed=14, xseed=106, kind="pool5"),
    dict(name="dh32_t65", H=8, d=256, L=2, B=2, T=65, wseed=15, xseed=107, kind="randn"),
    dict(name="ma_randmask_t96", H=4, d=256, L=4, B=2, T=96, wseed=11, xseed=108, kind="randn",
         randmask=7),
    dict(name="mb_pad_t150", H=4, d=512, L=3, B=2, T=150, wseed=12, xseed=109, kind="pool5",
         lengths=[150, 97]),
    dict(name="ma_t2000", H=4, d=256, L=4, B=1, T=2000, wseed=11, xseed=110, kind="randn"),
    dict(name="ma_nopos_t129", H=4, d=256, L=4, B=1, T=129, wseed=16, xseed=111, kind="randn",
         use_pos=False),
    dict(name="ma_nc3_t64", H=4, d=256, L=1, B=1, T=64, wseed=17, xseed=112, kind="randn",
         num_classes=3),
    # round 3: the reference's envelope is any d_model % num_heads == 0 (simnet.py:10-13, 123); wider models through the
    # plain GEMM + row LayerNorm path: d_model 768 / 1024, head dim 64 and 128
    dict(name="d768_h12_t200_pad", H=12, d=768, L=2, B=2, T=200, wseed=21, xseed=131, kind="pool5", lengths=[200, 133]),
    dict(name="d768_h6_t130", H=6, d=768, L=1, B=1, T=130, wseed=22, xseed=132, kind="randn"),
    dict(name="d1024_h8_t150", H=8, d=1024, L=2, B=1, T=150, wseed=23, xseed=133, kind="randn"),
    dict(name="d1024_h16_randmask_t96", H=16, d=1024, L=1, B=2, T=96, wseed=24, xseed=134, kind="randn", randmask=5),
    # round 4: shapes outside the kernels' own envelope (d_model not a multiple of 64, head dim not 32 / 64 / 128), scored
    # EMBEDDED in the next supported shape (simnet.embedding_plan): head dim 16 -> 32, 40 -> 64, 36 -> 64, 32 with three
    # heads (d_model 96 -> 192); plus two shapes the VERDICT named that are native after all (one head; five heads of 64)
    dict(name="d128_h8_t150_pad", H=8, d=128, L=2, B=2, T=150, wseed=31, xseed=141, kind="pool5", lengths=[150, 97]),
    dict(name="d200_h5_t130", H=5, d=200, L=2, B=1, T=130, wseed=32, xseed=142, kind="randn"),
    dict(name="d72_h2_randmask_t96", H=2, d=72, L=1, B=2, T=96, wseed=33, xseed=143, kind="randn", randmask=4),
    dict(name="d96_h3_nc3_t70", H=3, d=96, L=2, B=1, T=70, wseed=34, xseed=144, kind="randn", num_classes=3),
    dict(name="d128_h1_t100", H=1, d=128, L=2, B=1, T=100, wseed=35, xseed=145, kind="randn"),
    dict(name="d320_h5_t200_pad", H=5, d=320, L=2, B=2, T=200, wseed=36, xseed=146, kind="pool5", lengths=[200, 120]),
    # head dim 256 (num_heads=1, d_model=256; two heads of 256) and head dim 200 embedded in 256
    dict(name="d256_h1_t150_pad", H=1, d=256, L=2, B=2, T=150, wseed=41, xseed=147, kind="pool5", lengths=[150, 91]),
    dict(name="d512_h2_randmask_t96", H=2, d=512, L=1, B=2, T=96, wseed=42, xseed=148, kind="randn", randmask=3),
    dict(name="d200_h1_t130", H=1, d=200, L=2, B=1, T=130, wseed=43, xseed=149, kind="randn"),
]
# VS_GOLDEN_ONLY=name1,name2: (re)generate only these cases, keep every other fixture file as it is
ONLY = [n for n in os.environ.get("VS_GOLDEN_ONLY", "").split(",") if n]
HIDDEN_STRIDE = 7


def build_inputs(c):
    x = synth.make_features(c["B"], c["T"], c["xseed"], c["kind"], c.get("lengths"))
    mask = None
    if c.get("lengths") is not None:
        mask = synth.padding_mask(x)
    if c.get("randmask") is not None:
        mask = synth.random_mask(c["B"], c["T"], c["randmask"])
    return x, mask


def main():
    from model import SimNet          # the reference (reference src/model/__init__.py:2)
    torch.set_num_threads(os.cpu_count() or 1)
    index = []
    for c in CASES:
        if ONLY and c["name"] not in ONLY:
            index.append(c)
            continue
        nc = c.get("num_classes", 1)
        use_pos = c.get("use_pos", True)
        ref = SimNet(num_heads=c["H"], d_model=c["d"], num_layers=c["L"], sparsity=0.0,
                     dropout=0.3, num_classes=nc, use_pos=use_pos).eval()
        sd = synth.make_state_dict(c["d"], c["L"], c["wseed"], num_classes=nc, use_pos=use_pos)
        assert list(sd.keys()) == list(ref.state_dict().keys()), "state_dict key set differs"
        if use_pos:
            k = "embedding_layer.positional_encoding.pos_embedding"
            # the seeded table is the machine-independent evaluation of the reference's formula (synth.positional_table):
            # it sits within 1.3e-4 of the reference's own (machine-dependent, see there) torch fp32 buffer, which the
            # module's DEFAULT buffer reproduces bit for bit on the same machine
            assert (sd[k] - ref.state_dict()[k]).abs().max().item() < 1.3e-4, "positional table deviates"
            assert torch.equal(synth.positional_table(c["d"], sd[k].shape[1]), ref.state_dict()[k]), "default table not bit-equal"
        ref.load_state_dict(sd, strict=True)
        x, mask = build_inputs(c)
        with torch.no_grad():
            logits, hidden = ref(x, mask)
            logits2, inter = ref(x, mask, model_score=True)
        assert torch.equal(logits, logits2) and torch.equal(hidden, inter)
        rows = np.arange(0, c["T"], HIDDEN_STRIDE)
        np.savez_compressed(
            os.path.join(HERE, c["name"] + ".npz"),
            cfg=json.dumps(c),
            logits=logits.numpy(),
            hidden_rows=rows,
            hidden=hidden[:, rows].numpy(),
            mask=(mask.numpy() if mask is not None else np.zeros((0,), dtype=bool)),
        )
        index.append(c)
        print("%-20s logits %s |max| %.4f  hidden |max| %.4f" % (
            c["name"], tuple(logits.shape), logits.abs().max().item(), hidden.abs().max().item()))
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump({"torch": torch.__version__, "hidden_stride": HIDDEN_STRIDE, "cases": index}, f, indent=1)


if __name__ == "__main__":
    main()
