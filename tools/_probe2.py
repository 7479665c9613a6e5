import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-summarization_amd"); lib = pkg._lib.load(); dev = torch.device("cuda:0")
def run(q16, k16, v16, w64, checked=0):
    B, H, T, dh = q16.shape
    pkg._lib.set_option("VS_ATTN_W64", w64); pkg._lib.set_option("VS_ATTN_W64_CHECKED", checked)
    out = torch.full((B, T, H * dh), float("nan"), device=dev, dtype=torch.bfloat16)
    pkg._lib.check(lib.vs_attention_bf16_stored(q16.data_ptr(), k16.data_ptr(), v16.data_ptr(), None, out.data_ptr(), B, H, T, dh, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); pkg._lib.set_option("VS_ATTN_W64", -1); pkg._lib.set_option("VS_ATTN_W64_CHECKED", -1)
    return out
g = torch.Generator().manual_seed(1)
for T, pos in ((64, 10), (64, 40), (128, 10), (128, 100)):
    for (qa, kb) in ((1.0, 20000.0), (2.0, 10000.0), (100.0, 200.0), (200.0, 100.0), (1.0, 17000.0), (1.0, 16500.0), (1.0, 16300.0)):
        q = torch.randn(1, 1, T, 64, generator=g) * 0.1; k = torch.randn(1, 1, T, 64, generator=g) * 0.1; v = torch.randn(1, 1, T, 64, generator=g)
        q[..., 0] = qa; k[..., 0] = 0.0; k[:, :, pos, 0] = kb
        q16, k16, v16 = (t.to(torch.bfloat16).to(dev) for t in (q, k, v))
        r = [int((~torch.isfinite(run(q16, k16, v16, w, c))).sum()) for (w, c) in ((1, 0), (1, 1), (0, 0))]
        print("T=%d spike key %d: q %g x k %g = %g: non-finite w64 optimistic-first %d, w64 checked %d, 8-wave %d" % (T, pos, qa, kb, qa * kb, r[0], r[1], r[2]), flush=True)
