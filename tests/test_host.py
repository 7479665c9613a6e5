"""CPU (no GPU): host logic, the C-ABI surface, and the module's drop-in contract."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT


def test_library_builds_and_exports_every_declared_symbol(vsa):
    """libvsscore.so loads without a GPU and exports exactly what include/vs_scorer.h declares."""
    vsa._lib.build()
    lib = vsa._lib.load()
    hdr = open(os.path.join(ROOT, "include", "vs_scorer.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(vsa._lib.EXPORTS), declared ^ set(vsa._lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vs_abi_version() == vsa._lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "vs_eval.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vs_eval_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(vsa._lib.EVAL_EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    hdr = open(os.path.join(ROOT, "include", "vs_train.h")).read()          # the training C ABI
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(vsa._lib.TRAIN_EXPORTS), declared ^ set(vsa._lib.TRAIN_EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_c_abi_argument_checks_need_no_gpu(vsa):
    lib = vsa._lib.load()
    out = C.c_void_p()
    P = vsa._lib.ModelParams()
    for desc, word in ((vsa._lib.ModelDesc(100, 4, 1, 1024, 2000, 1), b"d_model"),
                       (vsa._lib.ModelDesc(256, 3, 1, 1024, 2000, 1), b"num_heads"),
                       (vsa._lib.ModelDesc(256, 16, 1, 1024, 2000, 1), b"head_dim"),
                       (vsa._lib.ModelDesc(256, 4, 0, 1024, 2000, 1), b"num_layers"),
                       (vsa._lib.ModelDesc(256, 4, 1, 1000, 2000, 1), b"in_features")):
        assert lib.vs_weights_pack(C.byref(desc), C.byref(P), None, C.byref(out)) == vsa._lib.VS_ERR_INVALID
        assert word in lib.vs_last_error()
    assert lib.vs_scorer_forward(None, None, None, 1, 1, 0, None, None, None, 0, None) == vsa._lib.VS_ERR_INVALID


def test_state_dict_key_set_matches_reference_layout(vsa):
    """Key names/shapes of SURVEY.md §8(a) row 1; loads strict=True."""
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=4, sparsity=0.0, dropout=0.3)
    sd = vsa.synth.make_state_dict(256, 4, 1)
    assert list(m.state_dict().keys()) == list(sd.keys())
    assert all(m.state_dict()[k].shape == v.shape for k, v in sd.items())
    m.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 3421697      # M-A parameter count
    assert m.state_dict()["embedding_layer.positional_encoding.pos_embedding"].shape == (1, 2000, 256)
    assert len(m.encoder.module_score) == 0


def test_ctor_defaults_and_attributes(vsa):
    m = vsa.SimNet()
    assert (m.num_heads, m.d_model, m.num_layers, m.num_classes, m.in_features, m.max_len) == (8, 512, 4, 1, 1024, 2500)
    cls = vsa.SimNet(num_heads=4, d_model=256, num_layers=1, use_cls=True)      # simnet.py:205-206: token first in the state_dict
    assert list(cls.state_dict())[0] == "embedding_layer.cls_token" and cls.state_dict()["embedding_layer.cls_token"].shape == (1, 1, 256)
    assert cls.process_mask(torch.zeros(2, 7, dtype=torch.bool)).shape == (2, 4, 8, 8)       # simnet.py:48-51
    with pytest.raises(AssertionError):
        vsa.SimNet(num_heads=3, d_model=256)
    mask = torch.zeros(2, 7, dtype=torch.bool)
    assert vsa.SimNet(num_heads=4, d_model=256, num_layers=1).process_mask(mask).shape == (2, 4, 7, 7)


def test_scoring_has_no_cpu_fallback(vsa):
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=1).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="HIP"):
        m(torch.zeros(1, 8, 1024))


def test_training_has_no_cpu_fallback_either(vsa):
    """train.py:111-131 (forward under autograd + backward) runs on the HIP training kernels (tests/test_hip_train.py);
    like scoring it refuses CPU tensors instead of falling back to composed torch ops."""
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=1, sparsity=0.0, dropout=0.0).train()
    with pytest.raises(RuntimeError, match="HIP"):
        m(torch.zeros(1, 8, 1024), torch.zeros(1, 8, dtype=torch.bool))


def test_default_positional_buffer_is_the_torch_formula_and_seeded_weights_use_the_stable_one(vsa):
    """The module's default buffer is the reference's own torch fp32 op sequence (bit-equal on one machine, asserted
    against the reference in tests/golden/make_golden.py); seeded test weights carry the machine-independent
    evaluation (synth.positional_table(stable=True)), within 1.3e-4 of it."""
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=1)
    buf = m.state_dict()["embedding_layer.positional_encoding.pos_embedding"]
    assert torch.equal(buf, vsa.synth.positional_table(256, 2000))
    st = vsa.synth.make_state_dict(256, 1, 1)["embedding_layer.positional_encoding.pos_embedding"]
    assert torch.equal(st, vsa.synth.positional_table(256, 2000, stable=True))
    assert (st - buf).abs().max().item() < 1.3e-4


def test_positional_table_formula(vsa):
    pe = vsa.synth.positional_table(256, 2000)
    assert pe.shape == (1, 2000, 256) and pe[0, 0, 0] == 0 and pe[0, 0, 1] == 1
    assert abs(pe[0, 3, 0].item() - float(torch.sin(torch.tensor(3.0)))) < 1e-6


def test_attention_dtype_is_validated(vsa):
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=1)
    assert m.attention_dtype == "fp32"
    m.attention_dtype = "bf16"
    with pytest.raises(ValueError):
        m.attention_dtype = "fp16"
    wide = vsa.SimNet(num_heads=4, d_model=512, num_layers=1)      # head_dim 128: exact and bf16 attention kernels only
    wide.attention_dtype = "bf16"
    with pytest.raises(ValueError):
        wide.attention_dtype = "fp16x3"
    huge = vsa.SimNet(num_heads=2, d_model=512, num_layers=1)      # head_dim 256: the exact attention only
    with pytest.raises(ValueError):
        huge.attention_dtype = "bf16"
    assert huge.set_compute_dtype("bf16").attention_dtype == "fp32" and huge.linear_dtype == "bf16"


def test_embedding_plan_covers_the_reference_envelope_up_to_head_dim_128(vsa):
    """simnet.py:123 accepts any d_model % num_heads == 0; shapes outside the kernels' own envelope run embedded in the next
    supported shape (zero-padded parameters, true LayerNorm width declared to the library)."""
    plan = vsa.simnet.embedding_plan
    for d, H in [(256, 4), (512, 4), (512, 8), (320, 5), (128, 1), (64, 1), (1024, 8), (768, 12), (256, 1), (512, 2), (1024, 4)]:
        assert plan(d, H) is None, (d, H)                      # native
    assert plan(128, 8) == (256, 32)                            # head dim 16 -> 32
    assert plan(200, 5) == (320, 64)                            # head dim 40 -> 64
    assert plan(96, 3) == (192, 64)                             # three heads of 32 would be d_model 96: not a multiple of 64
    assert plan(72, 2) == (128, 64)
    assert plan(640, 5) is None and plan(520, 5) == (640, 128)
    assert plan(200, 1) == (256, 256) and plan(400, 2) == (512, 256)      # head dim 200 -> 256
    for d, H in [(512, 1), (1032, 8), (126, 3), (1280, 5)]:     # head dim > 256, too wide, d_model % 4
        with pytest.raises(NotImplementedError):
            plan(d, H)
    m = vsa.SimNet(num_heads=8, d_model=128, num_layers=1)
    assert (m._lib_d, m._lib_dh) == (256, 32) and m.final_layer.weight.shape == (1, 128)     # the state_dict keeps the true shapes
    # the padding maps: residual-stream axes keep their index, head-structured axes move feature (h, j) to h * 32 + j
    w = torch.arange(128 * 128, dtype=torch.float32).reshape(128, 128)
    wp = m._pad(w, ("head", "res"))
    assert wp.shape == (256, 256) and torch.equal(wp[32 * 3 + 5, :128], w[16 * 3 + 5]) and wp[32 * 3 + 16:32 * 4].abs().sum() == 0
    assert torch.equal(m._unpad(wp, ("head", "res")), w) and wp[:, 128:].abs().sum() == 0
    assert m._padded_shape((512, 128), ("hid", "res")) == (1024, 256) and m._padded_shape((1, 128), (None, "res")) == (1, 256)
    axes = list(m._tensor_axes())
    assert len(axes) == len(list(m._tensors()))
    for t, ax in zip(m._tensors(), axes):
        assert m._unpad(m._pad(t.detach(), ax), ax).shape == t.shape


def test_compute_dtype_switches_are_validated(vsa):
    m = vsa.SimNet(num_heads=4, d_model=256, num_layers=1)
    for mode in ("fp16x3", "bf16", "fp32"):
        assert m.set_compute_dtype(mode) is m
        assert m.attention_dtype == mode and m.linear_dtype == mode
    with pytest.raises(ValueError):
        m.set_compute_dtype("fp8")
    with pytest.raises(ValueError):
        m.linear_dtype = "tf32"
    wide = vsa.SimNet(num_heads=4, d_model=512, num_layers=1)          # M-B: head dim 128
    wide.set_compute_dtype("fp16x3")                                   # plain projections emulated, the rest exact
    assert wide.attention_dtype == "fp32" and wide.linear_dtype == "fp16x3"
    wide.set_compute_dtype("bf16")                                     # head dim 128 has a bf16 attention; plain bf16 GEMMs
    assert wide.attention_dtype == "bf16" and wide.linear_dtype == "bf16"
    with pytest.raises(ValueError):
        wide.attention_dtype = "fp16x3"
    flags = vsa._lib
    assert flags.VS_FLAG_F16X3_LINEAR == 8 and flags.VS_FLAG_F16X3_ATTENTION == 16 and flags.VS_FLAG_BF16_LINEAR == 4


def test_header_is_plain_c_and_the_c_client_compiles(vsa):
    """include/vs_scorer.h + include/vs_eval.h compile as C99 (no C++, no torch types), and the plain-C client
    (tests/cabi/score_demo.c, run on the GPU box by tests/test_cabi_c.py) type-checks against them."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    assert gcc
    inc = os.path.join(ROOT, "include")
    for hdr in ("vs_scorer.h", "vs_eval.h", "vs_train.h"):
        r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, hdr)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    r = subprocess.run([gcc, "-std=gnu99", "-fsyntax-only", "-I" + inc, "-I" + os.path.join(rocm, "include"),
                        "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "cabi", "score_demo.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([gcc, "-std=gnu99", "-fsyntax-only", "-I" + inc, "-I" + os.path.join(rocm, "include"),
                        "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "cabi", "train_demo.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_fused_mlp_chunk_loop_has_only_its_counted_dma_on_the_vector_memory_counter(vsa, tmp_path):
    """The fused bf16 MLP kernel keeps one chunk of weights in flight across its per-chunk barrier with a COUNTED
    `s_waitcnt vmcnt(5)`: that count is only right if the 5 LDS-DMA copies are the loop's only vector-memory
    instructions.  A register spill reloaded inside the loop (scratch_load) or a hoisted global load would silently
    let the barrier pass before the weights have landed - so the ISA of the loop is checked at build time."""
    import subprocess
    csrc = os.path.join(ROOT, "video-summarization_amd", "csrc")
    asm = str(tmp_path / "mlp.s")
    r = subprocess.run([vsa._lib.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + csrc,
                        "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        os.path.join(csrc, "vs_mlp_fused.hip"), "-o", asm], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(asm).read()
    for nwv in (8, 4):                              # 8-wave blocks (256-row tiles) and 4-wave blocks (128-row tiles)
        pieces = 40 // nwv
        for tail in (0, 1):                         # MLP block alone; out-projection + norm1 + MLP block
            name = "mlp_fused_bf16ILb%dELi%dELi0E" % (tail, nwv)
            body = text[text.index(name, text.index(name) + 1):]
            body = body[:body.index(".Lfunc_end")]
            # every barrier-to-barrier segment that multiplies is one chunk of 32 MFMAs and ends in the COUNTED wait with
            # exactly its DMA pieces as the only LOADS on the counter (loads retire in order among themselves, so
            # "<= pieces outstanding" proves that the chunk issued one iteration earlier has landed; the QKV epilogue's
            # stores may sit anywhere - they can only make the wait longer, not shorter).  No spill reloads, no other loads.
            chunks = [seg for seg in body.split("s_barrier") if "v_mfma" in seg]
            assert len(chunks) >= (6 if tail else 2)        # out-projection x4 (unrolled), MLP loop, QKV epilogue loop
            for idx, seg in enumerate(chunks):
                assert len(re.findall(r"\bv_mfma_f32_32x32x16_bf16\b", seg)) == 32
                seg = seg[seg.index("global_load_lds_dwordx4"):]            # (what precedes is the previous wait's tail)
                assert len(re.findall(r"\bglobal_load_lds_dwordx4\b", seg)) == pieces
                assert str(pieces) in re.findall(r"s_waitcnt[^\n]*vmcnt\((\d+)\)", seg)
                if idx < (5 if tail else 1):            # out-projection and MLP chunks (99 % of the MFMAs): nothing else at all
                    vm = re.findall(r"\b(scratch_\w+|buffer_\w+|flat_\w+|global_(?!load_lds_dwordx4)\w+)\b", seg)
                    assert vm == [], vm
                    assert re.findall(r"s_waitcnt[^\n]*vmcnt\((\d+)\)", seg) == [str(pieces)]
        # the embedding kernel: per chunk the DMA pieces + this lane's 8 x-loads (two chunks ahead), 32 MFMAs, counted wait
        name = "embed_qkv_bf16ILi%dE" % nwv
        body = text[text.index(name, text.index(name) + 1):]
        body = body[:body.index(".Lfunc_end")]
        embed = [seg for seg in body.split("s_barrier") if "v_mfma" in seg and re.findall(r"s_waitcnt[^\n]*vmcnt\(%d\)" % (pieces + 8), seg)]
        full = 0
        for seg in embed:
            seg = seg[seg.index("global_load_lds_dwordx4"):]
            ops = re.findall(r"\b(global_load_lds_dwordx4|global_load_dwordx4|scratch_\w+|buffer_\w+|flat_\w+|global_store\w+|global_load_dword\b)", seg)
            assert len(re.findall(r"\bv_mfma_f32_32x32x16_bf16\b", seg)) == 32
            assert ops.count("global_load_lds_dwordx4") == pieces and set(ops) <= {"global_load_lds_dwordx4", "global_load_dwordx4"}, ops
            full += ops.count("global_load_dwordx4") == 8          # (the peeled last chunk of an odd count loads no x)
        assert full >= 2


def _device_functions(vsa, tmp_path):
    """{kernel name: [(address, instruction text, branch-target offset or None)]} of every gfx950 code object in the built
    library (llvm-objdump --offloading unbundles them next to a COPY of the .so)."""
    import shutil
    import subprocess
    od = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(od):
        pytest.skip("llvm-objdump not found")
    vsa._lib.build()
    lib = str(tmp_path / "lib.so")
    shutil.copy(vsa._lib.LIB_PATH, lib)
    subprocess.run([od, "--offloading", lib], capture_output=True, check=True)
    funcs = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" not in f:
            continue
        txt = subprocess.run([od, "-d", "--no-show-raw-insn", str(tmp_path / f)], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
            if m:
                cur = funcs.setdefault(m.group(1), [])
                continue
            m = re.match(r"^\s+(\S.*?)\s*// ([0-9A-Fa-f]+):(.*)$", line)
            if m and cur is not None:
                t = re.search(r"\+0x([0-9a-f]+)>", m.group(3))
                cur.append((int(m.group(2), 16), m.group(1), int(t.group(1), 16) if t else None))
    assert len(funcs) > 50
    return funcs


def test_no_kernel_touches_scratch_inside_its_innermost_mfma_loop(vsa, tmp_path):
    """a register spill inside a k-loop costs a memory round trip per MFMA step and falsifies counted
    `vmcnt` waits.  Every backward branch of every kernel of the built library is a loop; the innermost loops that hold
    MFMAs must hold no scratch_ instruction.  (Known and kept, outside the k-loops, once per output tile: the 64-wide-k
    bf16 GEMM instantiations park their store addresses around the k-loop - DESIGN section 17.)"""
    funcs = _device_functions(vsa, tmp_path)
    offenders, outer = [], {}
    seen_mfma_kernels = 0
    # the one accepted exception (round 4): the head-dim-256 instantiations of the fp32 training attention kernels, a
    # correctness-first form for SimNet(num_heads=1, d_model=256) and the like - one wave per SIMD, their
    # accumulators alone exceed the register file (dK + dV = 256 registers beside K, V and the two score tiles)
    dh256 = lambda n: any(k in n for k in ("attn_fwd_trainILi256E", "attn_bwd_dqILi256E", "attn_bwd_dkdvILi256E"))
    for name, ins in funcs.items():
        if not any(i.startswith("v_mfma") for _, i, _ in ins):
            continue
        seen_mfma_kernels += 1
        if dh256(name):
            continue
        base = ins[0][0]
        loops = [(base + t, a) for a, i, t in ins if (i.startswith("s_cbranch") or i.startswith("s_branch")) and t is not None and base + t <= a]
        count = lambda lo, hi, pfx: sum(1 for a, i, _ in ins if lo <= a <= hi and i.startswith(pfx))
        mf = [l for l in loops if count(l[0], l[1], "v_mfma") > 0]
        inner = [l for l in mf if not any(o != l and o[0] >= l[0] and o[1] <= l[1] for o in mf)]
        for lo, hi in inner:
            if count(lo, hi, "scratch_"):
                offenders.append((name, count(lo, hi, "v_mfma"), count(lo, hi, "scratch_")))
        tot = count(ins[0][0], ins[-1][0], "scratch_")
        if tot:
            outer[name] = tot
    assert seen_mfma_kernels > 100
    assert not offenders, offenders
    # the spills that exist are the documented ones: gemm_nt_128's 64-wide-k forms (per output tile,
    # around the k-loop), gemm_rows16<512> (prologue: its 128 registers of A) and the fused layer-tail kernel (2 / 6 in
    # the prologue); a NEW kernel with scratch fails here and has to be looked at
    assert all(("gemm_nt_128" in n or "gemm_rows16ILi512" in n or "mlp_fused_bf16ILb1" in n) for n in outer), sorted(outer)
    assert max(outer.values()) <= 80, outer


def test_every_documented_switch_is_a_known_option_and_unknown_names_are_refused(vsa):
    """include/vs_scorer.h names the A/B switches vs_set_option accepts; each must be known to the library (value -1 =
    back to the environment / built-in default: a no-op here), a misspelt one must be an error, not a silent no-op."""
    hdr = open(os.path.join(ROOT, "include", "vs_scorer.h")).read()
    block = hdr[hdr.index("A/B and test switches"):hdr.index("int vs_set_option")]
    names = sorted(set(re.findall(r"\bVS_[A-Z0-9_]+\b", block)) - {"VS_SCORER_H"})
    assert "VS_LP_STORE32" in names and "VS_LP_EMBED_UNFUSED" in names and "VS_LP_MIN_ROWS_FUSED" in names and len(names) >= 15
    for n in names:
        vsa._lib.set_option(n, -1)
    with pytest.raises(RuntimeError, match="unknown option"):
        vsa._lib.set_option("VS_LP_STORE_32", 1)
