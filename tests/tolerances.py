"""ONE stated tolerance per quantity, shared by the GPU tests, the soak (tests/fuzz_parity.py, tools/fuzz_train.py),
bench.py's contract test and the docs (DESIGN.md §2 / §8 / §12 quote these names).

The north-star bar (per-frame scores within 1e-4 of the reference's fp32 CPU path) applies to the DEFAULT exact-fp32
path and to the fp16x3 emulation.  The bf16 modes are opt-in and outside that bar; their tolerances are the measured
rounding of 8-bit mantissas with head-room, stated here once.
"""

# ---- exact fp32 path and the fp16x3 emulation: the north-star bar ---------------------------------------------
FP32_TOL = 1e-4                 # max |logit - reference| and max |hidden - reference| on valid frames, absolute

# ---- bf16 attention only (attention_dtype = "bf16", Linear layers exact) --------------------------------------
BF16_ATTN_KERNEL_REL = 1.5e-2   # one attention call against float64 on UNROUNDED operands, relative to max |reference|
BF16_ATTN_LOGIT_TOL = 2e-3      # end to end, max |logit - exact path|          (measured 3.8e-4)
BF16_ATTN_SCORE_TOL = 5e-4      # end to end, max |sigmoid(logit) - exact path| (measured 9.5e-5)

# ---- bf16 mode (every matrix product on the bf16 pipe: set_compute_dtype("bf16")) -----------------------------
BF16_LOGIT_TOL = 3e-2           # max |logit - exact path| on valid frames, ANY architecture / weights the soak draws
                                # (measured: trained-like weights 4.3e-3, soak worst 2.2e-2)
BF16_SCORE_TOL = 7.5e-3         # max |sigmoid(logit) - exact path| = BF16_LOGIT_TOL / 4 (sigmoid slope <= 1/4)
BF16_KERNEL_SHARED_ROUNDING = 2e-3   # one bf16 Linear / MLP-block kernel against float64 that shares its rounding points

# ---- training path (fp32 arithmetic against float64 gradients from the imported reference) --------------------
TRAIN_GRAD_ATOL = 1e-4          # per tensor, max |g - g64| absolute ...
TRAIN_GRAD_RTOL = 1e-3          # ... and relative to the tensor's largest entry (+1e-6 for analytically-zero sums)
# low-precision training (set_train_dtype("bf16" | "fp16"): MFMA operands rounded, fp32 accumulate / softmax / LayerNorm /
# loss) - the counterpart of the reference's fp16 autocast (train.py:120)
TRAIN_LP_GRAD_L2 = 3.5e-2       # per tensor, relative L2 error ||g - g64|| / ||g64|| over the golden's sampled rows
                                # (measured over the seven golden cases: 1.2e-2 ... 2.6e-2; 8-bit mantissas through ~10
                                # GEMMs and two LayerNorm backwards per layer; round 3 allowed 5e-2)
                                # - for mlp.fc1.weight ONLY, one sampled row is set aside when it alone breaks the bound: a ReLU
                                # unit whose pre-activation lies within bf16 rounding of zero flips, and that unit's whole
                                # fc1.weight row moves (measured: 78 % of the squared error in one of 8 sampled rows); every
                                # other tensor meets the bound with all its rows (round 3 allowed the set-aside everywhere)
                                # - the soak (tools/fuzz_train.py bf16) holds the q / k projection gradients relative to the
                                # same layer's v projection gradient: dS = P (dP - delta) is a difference, the bf16 rounding
                                # of dO and V enters at the scale of dP, and with diffuse attention little of dP is left
                                # (measured: 35 % of a q.weight gradient that is itself 1/30 of the layer's others)
TRAIN_LP_QK_L2 = 7e-2           # the SOAK's bound for the q / k projection gradients (relative to the larger of their own norm and the
                                # v projection's, scaled): random cases reach 5.6e-2 (one head of 256, dropout 0.5, T = 320, round 4;
                                # head dim 128 reaches 1.2e-1 of the tensor's OWN norm, which is 1/50 of the v projection's) - dS = P (dP
                                # - delta) is a difference that mostly cancels; the goldens hold these tensors to TRAIN_LP_GRAD_L2
TRAIN_LP_FC1_L2 = 5e-2          # mlp.fc1.weight after the set-aside, and mlp.fc1.bias - the same flipped units' entries (round 3's
                                # bound, kept for these two tensors of a layer alone: the d_model 768 golden has two flipped
                                # units among its sampled rows - 4.7e-2 with one of them set aside, 3.9e-2 in the bias)
TRAIN_LP_GRAD_RTOL = 2e-1       # per tensor, LARGEST element error relative to the tensor's largest entry: a gross-error
                                # bound only - a ReLU unit whose pre-activation lies within bf16 rounding of zero flips and
                                # moves one row of d_fc1 / one entry of its bias, by up to 14 % of the maximum in the
                                # 100-frame golden batches (measured: 0.6e-2 ... 1.4e-1)
TRAIN_LP_ZERO_ATOL = 5e-5       # norm of a gradient that is analytically zero (k.bias: softmax shift invariance) once the
                                # attention backward itself runs on bf16 operands (measured 4.0e-6)
TRAIN_LP_LOSS_RTOL = 2e-3       # |loss - loss64| relative

# ---- fp16 training mode (set_train_dtype("fp16"), VS_TRAIN_FLAG_FP16; round 4): 11 significant bits per operand against
# bf16's 8.  Measured over the seven golden cases under the loss scale the reference trains with (amp.GradScaler()'s initial
# 65 536, train.py:60; 1 024 gives the same figures - nothing underflows in these batches).  The L2 figures are ~3x under
# the bf16 mode's, not 8x: what is left is mostly ReLU units whose pre-activation lies within rounding of zero - eight times
# fewer of them flip, each moving its row as before, so the L2 error falls with the square root.
TRAIN_FP16_LOSS_SCALE = 65536.0
TRAIN_FP16_LOSS_RTOL = 2e-3
TRAIN_FP16_GRAD_L2 = 1e-2         # per tensor, relative L2 over the golden's sampled rows (measured 2.1e-3 ... 6.7e-3; bf16: 1.2e-2 ... 2.6e-2)
TRAIN_FP16_FC1_L2 = 2.5e-2        # mlp.fc1.weight / mlp.fc1.bias: the flipped units' own tensors (measured 1.7e-2 in the d_model 768
                                  # golden, whose 60-frame batch gives one unit 9 % of the bias gradient's maximum; bf16: 3.9e-2)
TRAIN_FP16_GRAD_RTOL = 1.5e-1     # largest element error / the tensor's largest entry: gross-error bound, a flipped unit moves one
                                  # row of d_fc1 / one bias entry as far as in bf16 (measured 0.6e-2 ... 8.7e-2)
TRAIN_FP16_ZERO_ATOL = 5e-5
