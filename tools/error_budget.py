#!/usr/bin/env python3
"""Where does the scoring path's distance to a float64 evaluation come from?  Runs the model layer by layer in float64
on the CPU and, beside it, every HIP kernel ON THE FLOAT64 PATH'S INPUTS (rounded to fp32), so each kernel's own error
is seen in isolation, then the end-to-end error.   python tools/error_budget.py [T]"""
import importlib
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd")
lib = pkg._lib.load()
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
H, d, L = 4, 256, 4
sd = pkg.synth.make_state_dict(d, L, 11)
x = pkg.synth.make_features(1, T, 110, "randn")
st = torch.cuda.current_stream().cuda_stream
P = {k: v.double() for k, v in sd.items()}


def dev32(t):
    return t.float().to(dev).contiguous()


def err(name, got, want):
    e = (got.cpu().double() - want).abs()
    print("  %-34s max err %.2e  rms err %.2e  (max |want| %.2e)" % (name, e.max().item(), e.pow(2).mean().sqrt().item(), want.abs().max().item()))


def linear(A, W, b, relu=0, pe=None, Tpe=0):
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device=dev)
    a, w, bb = dev32(A), dev32(W), dev32(b)
    pp = dev32(pe) if pe is not None else None
    pkg._lib.check(lib.vs_linear_f32(a.data_ptr(), w.data_ptr(), bb.data_ptr(), out.data_ptr(), M, N, K, relu,
                                     pp.data_ptr() if pp is not None else None, Tpe, st))
    return out


def lin_ln(A, W, b, res, g, be):
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device=dev)
    t = [dev32(v) for v in (A, W, b, res, g, be)]
    pkg._lib.check(lib.vs_linear_residual_layernorm_f32(*[v.data_ptr() for v in t], out.data_ptr(), M, N, K, None, None, 0, 0, None, st))
    return out


for pin in (-1, 0):
    pkg._lib.set_option("VS_SKINNY_ROWS", pin)
    print("== T=%d, %s kernels" % (T, "default (skinny below 16384 rows)" if pin < 0 else "tiled"))
    h = F.linear(x.double()[0], P["embedding_layer.feature_transform.weight"], P["embedding_layer.feature_transform.bias"]) + P["embedding_layer.positional_encoding.pos_embedding"][0, :T]
    err("embed + pe", linear(x[0], sd["embedding_layer.feature_transform.weight"], sd["embedding_layer.feature_transform.bias"], 0,
                             sd["embedding_layer.positional_encoding.pos_embedding"][0, :T], T), h)
    for l in range(L):
        p = "encoder.module_list.%d." % l
        Wqkv = torch.cat([P[p + "sa.%s.weight" % n] for n in "qkv"]); bqkv = torch.cat([P[p + "sa.%s.bias" % n] for n in "qkv"])
        qkv = F.linear(h, Wqkv, bqkv).view(T, 3, H, d // H).permute(1, 2, 0, 3)          # [3,H,T,dh]
        out = torch.empty(3, 1, H, T, d // H, device=dev)
        hh, ww, bb = dev32(h), dev32(Wqkv), dev32(bqkv)
        pkg._lib.check(lib.vs_qkv_proj_f32(hh.data_ptr(), ww.data_ptr(), bb.data_ptr(), out.data_ptr(), 1, T, d, H, st))
        err("L%d qkv" % l, out[:, 0], qkv)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = torch.matmul(q, k.transpose(1, 2)) * d ** -0.5
        o = torch.matmul(torch.softmax(s, dim=2), v).permute(1, 0, 2).reshape(T, d)
        att = torch.empty(1, T, d, device=dev)
        qq, kk, vv = dev32(q[None]), dev32(k[None]), dev32(v[None])
        pkg._lib.check(lib.vs_attention_f32(qq.data_ptr(), kk.data_ptr(), vv.data_ptr(), None, att.data_ptr(), 1, H, T, d // H, d ** -0.5, st))
        err("L%d attention" % l, att[0], o)
        a = F.linear(o, P[p + "sa.feature_projection.weight"], P[p + "sa.feature_projection.bias"])
        h1 = F.layer_norm(a + h, (d,), P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-5)
        err("L%d out-proj + LN1" % l, lin_ln(o, P[p + "sa.feature_projection.weight"], P[p + "sa.feature_projection.bias"], h, P[p + "norm1.weight"], P[p + "norm1.bias"]), h1)
        f = F.relu(F.linear(h1, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"]))
        err("L%d fc1 + relu" % l, linear(h1, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"], 1), f)
        h2 = F.layer_norm(F.linear(f, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"]) + h1, (d,), P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-5)
        err("L%d fc2 + LN2" % l, lin_ln(f, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"], h1, P[p + "norm2.weight"], P[p + "norm2.bias"]), h2)
        h = h2
    m = pkg.SimNet(num_heads=H, d_model=d, num_layers=L, sparsity=0.0, dropout=0.3)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    with torch.no_grad():
        lg, hd = m(x.to(dev))
    err("END TO END hidden", hd[0], h)
pkg._lib.set_option("VS_SKINNY_ROWS", -1)
