"""ctypes binding of libvsscore.so (C ABI: ``include/vs_scorer.h``) and its in-tree build.

The library is the product path; nothing here falls back to PyTorch or to the oracle.  If the
shared object is missing or a symbol is absent, ``load()`` raises — loudly — instead.
"""
from __future__ import annotations

import concurrent.futures
import ctypes as C
import os
import subprocess
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_PATH = os.path.join(HERE, "libvsscore.so")
DIAG_LIB_PATH = os.path.join(HERE, "libvsscore_diag.so")
SOURCES = ("vs_kernels.hip", "vs_attention.hip", "vs_attention_w64.hip", "vs_mlp_fused.hip", "vs_gemm_ring.hip", "vs_scorer.cpp", "vs_eval.cpp",
           "vs_train_kernels.hip", "vs_train_attention.hip", "vs_train_attention_bf16.hip", "vs_train_gemm_rows.hip", "vs_pretrain_kernels.hip",
           "vs_train.cpp")
ABI_VERSION = 3

VS_OK, VS_ERR_INVALID, VS_ERR_WORKSPACE, VS_ERR_HIP = 0, 1, 2, 3
VS_FLAG_SIGMOID = 1
VS_FLAG_BF16_ATTENTION = 2
VS_FLAG_BF16_LINEAR = 4
VS_FLAG_F16X3_LINEAR = 8
VS_FLAG_F16X3_ATTENTION = 16
VS_FLAG_SPLITK = 32              # opt-in latency mode for reference-sized calls (include/vs_scorer.h)
VS_TRAIN_FLAG_BF16_LINEAR = 1
VS_TRAIN_FLAG_BF16_ATTENTION = 2
VS_TRAIN_FLAG_FP16 = 4          # modifier: the 16-bit type is IEEE fp16 (the reference's own autocast type) instead of bf16

# every symbol include/vs_scorer.h declares
EXPORTS = ("vs_abi_version", "vs_last_error", "vs_weights_pack", "vs_weights_free", "vs_weights_update", "vs_weights_set_norm_width", "vs_set_option",
           "vs_scorer_workspace_bytes", "vs_scorer_forward", "vs_scorer_workspace_bytes_packed",
           "vs_scorer_forward_packed", "vs_scorer_workspace_bytes_cls", "vs_scorer_forward_cls", "vs_linear_f32", "vs_qkv_proj_f32",
           "vs_attention_f32", "vs_attention_bf16", "vs_attention_bf16_stored", "vs_attention_qscale", "vs_attention_f16x3", "vs_linear_residual_layernorm_f32",
           "vs_linear_bf16", "vs_linear_residual_layernorm_bf16", "vs_linear_f16x3",
           "vs_linear_residual_layernorm_f16x3", "vs_mlp_block_bf16",
           "vs_linear_bf16_operands", "vs_qkv_proj_bf16_operands", "vs_to_bf16", "vs_linear_bf16_a16",
           "vs_profile_enable", "vs_profile_collect", "vs_stage_name")
# include/vs_eval.h
EVAL_EXPORTS = ("vs_eval_upsample", "vs_eval_knapsack", "vs_eval_generate_summary", "vs_eval_fscore",
                "vs_eval_rank_correlation", "vs_eval_corpus")
# include/vs_train.h
TRAIN_EXPORTS = ("vs_train_prepare", "vs_train_saved_bytes", "vs_train_workspace_bytes", "vs_train_forward", "vs_train_backward",
                 "vs_mse_mask_loss_forward", "vs_mse_mask_loss_backward", "vs_train_attention_forward",
                 "vs_train_attention_backward", "vs_train_attention_dropout_bits_bytes", "vs_train_attention_dropout_bits",
                 "vs_train_attention_forward_bf16", "vs_train_attention_backward_bf16", "vs_train_wgrad_scratch_floats", "vs_train_wgrad", "vs_train_wgrad_bf16",
                 "vs_train_dropout_mask_attention", "vs_train_dropout_mask_rows", "vs_train_dropout_site", "vs_train_saved_field", "vs_train_last_format",
                 "vs_pretrain_head_state_bytes", "vs_pretrain_head_workspace_bytes", "vs_pretrain_head_forward",
                 "vs_pretrain_head_backward")
NUM_STAGES = 6


class ModelDesc(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("num_heads", C.c_int32), ("num_layers", C.c_int32),
                ("in_features", C.c_int32), ("max_len", C.c_int32), ("num_classes", C.c_int32)]


_LAYER_FIELDS = ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b",
                 "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")


class LayerParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_FIELDS]


class ModelParams(C.Structure):
    _fields_ = [("embed_w", C.c_void_p), ("embed_b", C.c_void_p), ("pos_embedding", C.c_void_p),
                ("layers", C.POINTER(LayerParams)), ("final_w", C.c_void_p), ("final_b", C.c_void_p)]


class DropoutCfg(C.Structure):
    _fields_ = [("p_embed", C.c_float), ("p", C.c_float), ("seed", C.c_uint64), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class EvalVideo(C.Structure):   # vs_eval_video (include/vs_eval.h)
    _fields_ = [("scores", C.c_void_p), ("positions", C.c_void_p), ("change_points", C.c_void_p), ("user_summary", C.c_void_p),
                ("user_scores", C.c_void_p)] + [(n, C.c_int32) for n in ("n_scores", "n_positions", "n_frames", "n_shots", "n_users",
                                                                        "user_len", "n_score_users", "use_max", "user_scores_f32")]


LayerGrads = LayerParams        # vs_layer_grads: same field names, destinations instead of sources


class ModelGrads(C.Structure):  # vs_model_grads (no positional table: it is a buffer, not a parameter)
    _fields_ = [("embed_w", C.c_void_p), ("embed_b", C.c_void_p), ("layers", C.POINTER(LayerParams)),
                ("final_w", C.c_void_p), ("final_b", C.c_void_p)]


def hipcc_path() -> str:
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """Cross-compiles the HIP sources for gfx950 into the in-tree ``libvsscore.so``
    (works without a GPU).  Returns the library path.  ``diag=True`` builds ``libvsscore_diag.so`` instead: the same
    sources with ``-DVS_WITH_DIAG`` (stamped GEMM instantiations, the fused-MLP negative result, the non-pipelined
    attention) for ``tools/`` only - the product library does not carry them."""
    if diag:
        return _build(DIAG_LIB_PATH, ["-DVS_WITH_DIAG"], verbose)
    if not force and not needs_build():
        return LIB_PATH
    return _build(LIB_PATH, [], verbose)


def _build(out_path: str, extra, verbose: bool) -> str:
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize",
             "-pthread", "-I" + INCLUDE, "-I" + CSRC] + list(extra)
    # -fno-slp-vectorize: packed f32 VALU (v_pk_mul/add_f32) beside MFMAs costs more than the scalar
    # forms it replaces (MI355X_MICROARCH.md, cycle constants); keep elementwise epilogue/softmax ops scalar
    tmp = out_path + ".tmp.%d" % os.getpid()
    objs = [tmp + "." + os.path.splitext(src)[0] + ".o" for src in SOURCES]

    def compile_one(pair):
        src, obj = pair
        cmd = [hipcc_path()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        return subprocess.run(cmd, capture_output=True, text=True)

    try:
        # one hipcc per translation unit, side by side (the two kernel files dominate the build time)
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
            for r in pool.map(compile_one, zip(SOURCES, objs)):
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed building libvsscore.so:\n" + r.stdout + r.stderr)
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", tmp]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed linking libvsscore.so:\n" + r.stdout + r.stderr)
    finally:
        for o in objs:
            if os.path.exists(o):
                os.remove(o)
    os.replace(tmp, out_path)
    return out_path


_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """dlopen + signature setup.  Raises RuntimeError when the library is missing/stale-ABI."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # VS_LIBRARY: tools/ point this at libvsscore_diag.so (build(diag=True)); never a fallback
        path = os.environ.get("VS_LIBRARY") or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                "libvsscore.so not found at %s — the HIP scorer library is not built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no PyTorch/CPU fallback for the scoring path." % path)
        lib = C.CDLL(path)
        for name in EXPORTS + EVAL_EXPORTS + TRAIN_EXPORTS:
            if not hasattr(lib, name):
                raise RuntimeError("libvsscore.so lacks symbol %s (stale build?)" % name)
        lib.vs_abi_version.restype = C.c_int
        if lib.vs_abi_version() != ABI_VERSION:
            raise RuntimeError("libvsscore.so ABI %d != binding ABI %d" % (lib.vs_abi_version(), ABI_VERSION))
        lib.vs_last_error.restype = C.c_char_p
        lib.vs_weights_pack.restype = C.c_int
        lib.vs_weights_pack.argtypes = [C.POINTER(ModelDesc), C.POINTER(ModelParams), C.c_void_p,
                                        C.POINTER(C.c_void_p)]
        lib.vs_weights_free.restype = None
        lib.vs_weights_free.argtypes = [C.c_void_p]
        lib.vs_weights_update.restype = C.c_int
        lib.vs_weights_update.argtypes = [C.c_void_p, C.POINTER(ModelParams), C.c_void_p]
        lib.vs_weights_set_norm_width.restype = C.c_int
        lib.vs_weights_set_norm_width.argtypes = [C.c_void_p, C.c_int32]
        lib.vs_set_option.restype = C.c_int
        lib.vs_set_option.argtypes = [C.c_char_p, C.c_int32]
        lib.vs_scorer_workspace_bytes.restype = C.c_size_t
        lib.vs_scorer_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.vs_scorer_forward.restype = C.c_int
        lib.vs_scorer_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p]
        lib.vs_scorer_workspace_bytes_cls.restype = C.c_size_t
        lib.vs_scorer_workspace_bytes_cls.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.vs_scorer_forward_cls.restype = C.c_int
        lib.vs_scorer_forward_cls.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                              C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.vs_scorer_workspace_bytes_packed.restype = C.c_size_t
        lib.vs_scorer_workspace_bytes_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        lib.vs_scorer_forward_packed.restype = C.c_int
        lib.vs_scorer_forward_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.vs_linear_f32.restype = C.c_int
        lib.vs_linear_f32.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_void_p]
        lib.vs_qkv_proj_f32.restype = C.c_int
        lib.vs_qkv_proj_f32.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]
        lib.vs_attention_f32.restype = C.c_int
        lib.vs_attention_f32.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_float, C.c_void_p]
        lib.vs_attention_bf16_stored.restype = C.c_int
        lib.vs_attention_bf16_stored.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_void_p]
        lib.vs_attention_qscale.restype = C.c_float
        lib.vs_attention_qscale.argtypes = [C.c_float]
        lib.vs_attention_bf16.restype = C.c_int
        lib.vs_attention_bf16.argtypes = lib.vs_attention_f32.argtypes
        lib.vs_attention_f16x3.restype = C.c_int
        lib.vs_attention_f16x3.argtypes = lib.vs_attention_f32.argtypes
        lib.vs_linear_residual_layernorm_f32.restype = C.c_int
        lib.vs_linear_residual_layernorm_f32.argtypes = ([C.c_void_p] * 7 + [C.c_int32] * 3 + [C.c_void_p] * 2
                                                         + [C.c_int32] * 2 + [C.c_void_p] * 2)
        lib.vs_linear_bf16.restype = C.c_int
        lib.vs_linear_bf16.argtypes = lib.vs_linear_f32.argtypes
        lib.vs_linear_bf16_operands.restype = C.c_int
        lib.vs_linear_bf16_operands.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 5 + [C.c_void_p]
        lib.vs_qkv_proj_bf16_operands.restype = C.c_int
        lib.vs_qkv_proj_bf16_operands.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 5 + [C.c_void_p]
        lib.vs_linear_bf16_a16.restype = C.c_int
        lib.vs_linear_bf16_a16.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p, C.c_void_p]
        lib.vs_to_bf16.restype = C.c_int
        lib.vs_to_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.vs_linear_residual_layernorm_bf16.restype = C.c_int
        lib.vs_linear_residual_layernorm_bf16.argtypes = lib.vs_linear_residual_layernorm_f32.argtypes
        lib.vs_linear_f16x3.restype = C.c_int
        lib.vs_linear_f16x3.argtypes = lib.vs_linear_f32.argtypes
        lib.vs_linear_residual_layernorm_f16x3.restype = C.c_int
        lib.vs_linear_residual_layernorm_f16x3.argtypes = lib.vs_linear_residual_layernorm_f32.argtypes
        lib.vs_mlp_block_bf16.restype = C.c_int
        lib.vs_mlp_block_bf16.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p]
        for name in EVAL_EXPORTS:
            getattr(lib, name).restype = C.c_int
        lib.vs_eval_upsample.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        lib.vs_eval_knapsack.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
        lib.vs_eval_generate_summary.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                 C.c_int32, C.c_void_p, C.c_int32]
        lib.vs_eval_fscore.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(C.c_double)]
        lib.vs_eval_rank_correlation.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_double),
                                                 C.POINTER(C.c_double)]
        lib.vs_train_last_format.restype = C.c_uint32
        lib.vs_train_last_format.argtypes = []
        lib.vs_eval_corpus.restype = C.c_int
        lib.vs_eval_corpus.argtypes = [C.POINTER(EvalVideo), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        # include/vs_train.h
        lib.vs_train_prepare.restype = C.c_int
        lib.vs_train_prepare.argtypes = [C.c_void_p, C.c_void_p]
        lib.vs_train_saved_bytes.restype = C.c_size_t
        lib.vs_train_saved_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.vs_train_workspace_bytes.restype = C.c_size_t
        lib.vs_train_workspace_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.vs_train_forward.restype = C.c_int
        lib.vs_train_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(DropoutCfg),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.vs_train_backward.restype = C.c_int
        lib.vs_train_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(DropoutCfg),
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(ModelGrads), C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p]
        lib.vs_mse_mask_loss_forward.restype = C.c_int
        lib.vs_mse_mask_loss_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
        lib.vs_mse_mask_loss_backward.restype = C.c_int
        lib.vs_mse_mask_loss_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                  C.c_void_p, C.c_void_p]
        lib.vs_train_attention_forward.restype = C.c_int
        lib.vs_train_attention_forward.argtypes = ([C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_float, C.c_uint64, C.c_uint32,
                                                                                       C.c_float, C.c_void_p])
        lib.vs_train_attention_backward.restype = C.c_int
        lib.vs_train_attention_backward.argtypes = ([C.c_void_p] * 9 + [C.c_int32] * 4 + [C.c_float, C.c_uint64, C.c_uint32,
                                                                                        C.c_float, C.c_void_p])
        lib.vs_train_attention_dropout_bits_bytes.restype = C.c_size_t
        lib.vs_train_attention_dropout_bits_bytes.argtypes = [C.c_int32] * 3
        lib.vs_train_attention_dropout_bits.restype = C.c_int
        lib.vs_train_attention_dropout_bits.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32,
                                                        C.c_float, C.c_void_p]
        lib.vs_train_attention_forward_bf16.restype = C.c_int
        lib.vs_train_attention_forward_bf16.argtypes = ([C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_float, C.c_float, C.c_void_p,
                                                                                            C.c_int32, C.c_void_p])
        lib.vs_train_attention_backward_bf16.restype = C.c_int
        lib.vs_train_attention_backward_bf16.argtypes = ([C.c_void_p] * 9 + [C.c_int32] * 4 + [C.c_float, C.c_float, C.c_void_p,
                                                                                             C.c_int32, C.c_void_p])
        lib.vs_train_wgrad_scratch_floats.restype = C.c_size_t
        lib.vs_train_wgrad_scratch_floats.argtypes = [C.c_int32] * 3
        lib.vs_train_wgrad.restype = C.c_int
        lib.vs_train_wgrad.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
        lib.vs_train_wgrad_bf16.restype = C.c_int
        lib.vs_train_wgrad_bf16.argtypes = lib.vs_train_wgrad.argtypes
        lib.vs_train_dropout_mask_attention.restype = C.c_int
        lib.vs_train_dropout_mask_attention.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32,
                                                        C.c_float, C.c_void_p]
        lib.vs_train_dropout_mask_rows.restype = C.c_int
        lib.vs_train_dropout_mask_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32, C.c_float,
                                                   C.c_void_p]
        lib.vs_pretrain_head_state_bytes.restype = C.c_size_t
        lib.vs_pretrain_head_state_bytes.argtypes = [C.c_int32] * 3
        lib.vs_pretrain_head_workspace_bytes.restype = C.c_size_t
        lib.vs_pretrain_head_workspace_bytes.argtypes = [C.c_int32] * 4
        lib.vs_pretrain_head_forward.restype = C.c_int
        lib.vs_pretrain_head_forward.argtypes = ([C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_float, C.c_int32]
                                                 + [C.c_void_p] * 4)
        lib.vs_pretrain_head_backward.restype = C.c_int
        lib.vs_pretrain_head_backward.argtypes = ([C.c_void_p] * 8 + [C.c_int32] * 4 + [C.c_float, C.c_int32]
                                                  + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p])
        lib.vs_train_saved_field.restype = C.c_int
        lib.vs_train_saved_field.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t),
                                             C.POINTER(C.c_size_t)]
        lib.vs_train_dropout_site.restype = C.c_uint32
        lib.vs_train_dropout_site.argtypes = [C.c_int32, C.c_int32]
        lib.vs_profile_enable.restype = C.c_int
        lib.vs_profile_enable.argtypes = [C.c_int32]
        lib.vs_profile_collect.restype = C.c_int
        lib.vs_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.vs_stage_name.restype = C.c_char_p
        lib.vs_stage_name.argtypes = [C.c_int32]
        _lib = lib
    return _lib


def set_option(name: str, value: int) -> None:
    """A/B / test switch of the library (include/vs_scorer.h: vs_set_option); value < 0 restores the default."""
    check(load().vs_set_option(name.encode(), int(value)))


def profile_collect():
    """{stage_name: (ms_sum, launches)} of the stages recorded since vs_profile_enable(1)."""
    lib = load()
    ms = (C.c_double * NUM_STAGES)()
    n = (C.c_int64 * NUM_STAGES)()
    check(lib.vs_profile_collect(ms, n))
    return {lib.vs_stage_name(i).decode(): (ms[i], n[i]) for i in range(NUM_STAGES)}


def check(rc: int) -> None:
    """Non-zero status -> RuntimeError with the library's message (reference: ATen RuntimeError)."""
    if rc != VS_OK:
        msg = load().vs_last_error().decode("utf-8", "replace")
        raise RuntimeError("libvsscore: %s (status %d)" % (msg, rc))
