#!/usr/bin/env python3
"""Generates video-summarization_amd/csrc/vs_attention_w64_asm.inc: the hand-placed gfx950 instruction stream of the
one-wave-per-SIMD bf16 attention kernel (csrc/vs_attention_w64.hip; algorithm and register plan in that file's header).

    python tools/gen_attn_w64.py            # rewrites the .inc (committed; the build does not run this script)

Why generated assembly: with a 512-register budget hipcc places the S' accumulators in AGPRs and moves them through
v_accvgpr_read/write around every exp2 (1 579 reads + 1 587 writes per loop body in the plain-HIP form of this kernel,
DESIGN section 17); here every register is assigned by hand and each MFMA gap gets its fillers explicitly.

Register plan (asm-owned; v0..v23 and s0..s35 are left to the compiler for the operands it hands in):
  a[0:47]  O_A (d 0..31 | d 32..63 | row sums)   a[48:95] O_B      a[96:111] Q_A fragments   a[112:127] Q_B
  a[128:159] K fragments (n*4 + ks)              a[160:191] V^T fragments ((n*2+s)*2 + d)    a[192:195] bf16 ones
  v[32:63] S'_A (key block 0 | 1)   v[64:95] S'_B   v[96:111] -c_A (x16)   v[112:127] -c_B
  v[128:191] P [tile parity][row block][key block][k step] x 4 dwords
  v[192:199] exp2 results   v200 / v201 row constants c_A / c_B   v[202:205] K read addresses   v[206:207] V read addresses
  v[208:251] rare-path temporaries   v24 16*h   v25 K piece-1 DMA offset   v26/v27 OR accumulators
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "video-summarization_amd", "csrc", "vs_attention_w64_asm.inc")

# ---- registers ----
OA, OB, QA, QB, KF, VF, ONES = 0, 48, 96, 112, 128, 160, 192


def O(x, d):
    return (OA if x == 0 else OB) + 16 * d


def Qf(x, ks):
    return (QA if x == 0 else QB) + 4 * ks


def S(x, n, i=0):
    return 32 + 32 * x + 16 * n + i


def NEGC(x, i=0):
    return 96 + 16 * x + i


def P(par, x, n, ss, j=0):
    return 128 + 4 * (((par * 2 + x) * 2 + n) * 2 + ss) + j


E0 = 192
VM = (200, 201)
VKA, VVA = 202, 206
VH16, VDK1, VOR0, VOR1, VDV1 = 24, 25, 26, 27, 28
RT = 208          # rare-path temporaries v208..v251

# SGPRs
QD, KD, VD, OD = 36, 40, 44, 48
sT, sT1, sNT = 52, 53, 54
sKSO, sVSO = 55, 56                 # soffset of the next K / V DMA
sKB0, sVB0 = 57, 58                 # LDS address of this wave's piece 2w in buffer 0 of K / V
sR0, sR1, sR2 = 59, 60, 61          # ring offsets (t % 3, (t+1) % 3, (t+2) % 3) * 8192
sNINF, sN32K, sP32K = 62, 63, 68
sRAISE = 64                         # s[64:65]
sMSK = 66                           # s[66:67]
sTMP, sTMP2, sMB, sKB, sVB, sC4096, sOB = 69, 70, 71, 72, 73, 74, 75
sLAST, sKB1, sVB1, sR3, sMODE, sWV4, sNPADN = 76, 77, 78, 79, 80, 81, 82


def vr(b, n=1):
    return "v%d" % b if n == 1 else "v[%d:%d]" % (b, b + n - 1)


def ar(b, n=1):
    return "a%d" % b if n == 1 else "a[%d:%d]" % (b, b + n - 1)


def sr(b, n=1):
    return "s%d" % b if n == 1 else "s[%d:%d]" % (b, b + n - 1)


class Gen:
    def __init__(self):
        self.lines = []
        self.tail = []          # out-of-line rare blocks
        self.nsite = 0

    def e(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(name + "_%=:")

    def site(self):
        self.nsite += 1
        return "S%d" % self.nsite


def mfma(dst, a, b, c):
    return "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (dst, a, b, c)


def s_mfma(x, j):
    """The j-th MFMA of S'(., x).  The two key blocks' chains alternate (n0 k0, n1 k0, n0 k1, ...): measured, a chain of
    dependent v_mfma_f32_32x32x16_bf16 with VGPR accumulators issues every ~34.5 cycles, not 32 (MFMA-only ablation:
    1381 cycles per 40 MFMAs).  MFMA (n, ks) takes K fragment n*4 + ks, Q fragment ks; ks = 0 starts from -c."""
    i = (j % 2) * 4 + j // 2
    n, ks = i // 4, i % 4
    c = vr(NEGC(x), 16) if ks == 0 else vr(S(x, n), 16)
    return mfma(vr(S(x, n), 16), ar(KF + 4 * i, 4), ar(Qf(x, ks), 4), c)


def pv_mfma(x, par, i):
    """MFMA i of O_x += V^T P^T: i = 3 g + w, g = n*2 + s; w = 0 / 1 the two d blocks, w = 2 the row sums"""
    g, w = i // 3, i % 3
    p = vr(P(par, x, g >> 1, g & 1), 4)
    if w < 2:
        return mfma(ar(O(x, w), 16), ar(VF + 4 * (2 * g + w), 4), p, ar(O(x, w), 16))
    return mfma(ar(O(x, 2), 16), ar(ONES, 4), p, ar(O(x, 2), 16))


def pair_exp(x, k):
    n, ss, j = k // 8, (k // 4) % 2, k % 4
    e = E0 + 2 * (k % 4)
    s0 = S(x, n, 8 * ss + 2 * j)
    return ["v_exp_f32 %s, %s" % (vr(e), vr(s0)), "v_exp_f32 %s, %s" % (vr(e + 1), vr(s0 + 1))]


def pair_cvt(x, par, k):
    n, ss, j = k // 8, (k // 4) % 2, k % 4
    e = E0 + 2 * (k % 4)
    return "v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(P(par, x, n, ss, j)), vr(e), vr(e + 1))


def preg_of_pair(x, par, k):
    n, ss, j = k // 8, (k // 4) % 2, k % 4
    return P(par, x, n, ss, j)


def rebase_block(g, x, par, name, ret, o_is_zero=False):
    """Out-of-line rare path.  par None: the tile's first look at a row without a constant (no raise request, P not
    computed yet); else: some P reached 2 (lanes in vcc) - raise, rescale, recompute the tile's P into buffer par."""
    t = g.tail
    m = VM[x]
    t.append(name + "_%=:")
    t.append("s_nop 7")
    t.append("s_nop 7")
    if par is not None:
        t.append("v_cndmask_b32_e64 %s, 0, 1, vcc" % vr(RT))
        t.append("v_mov_b32 %s, %s" % (vr(RT + 1), vr(RT)))
        t.append("s_nop 1")
        t.append("v_permlane32_swap_b32 %s, %s" % (vr(RT), vr(RT + 1)))
        t.append("v_or_b32 %s, %s, %s" % (vr(RT), vr(RT), vr(RT + 1)))
        t.append("v_cmp_ne_u32_e64 %s, 0, %s" % (sr(sRAISE, 2), vr(RT)))
    else:
        t.append("s_mov_b64 %s, 0" % sr(sRAISE, 2))
    sregs = [S(x, n, i) for n in range(2) for i in range(16)]
    mx = RT + 2
    t.append("v_max3_f32 %s, %s, %s, %s" % (vr(mx), vr(sregs[0]), vr(sregs[1]), vr(sregs[2])))
    for i in range(3, 31, 2):
        t.append("v_max3_f32 %s, %s, %s, %s" % (vr(mx), vr(mx), vr(sregs[i]), vr(sregs[i + 1])))
    t.append("v_max_f32 %s, %s, %s" % (vr(mx), vr(mx), vr(sregs[31])))
    t.append("v_mov_b32 %s, %s" % (vr(mx + 1), vr(mx)))
    t.append("s_nop 1")
    t.append("v_permlane32_swap_b32 %s, %s" % (vr(mx), vr(mx + 1)))
    t.append("v_max_f32 %s, %s, %s" % (vr(mx), vr(mx), vr(mx + 1)))
    cb, raw, cand, cnew, unew, shift, alpha, tmp = RT + 4, RT + 5, RT + 6, RT + 7, RT + 8, RT + 9, RT + 10, RT + 11
    t.append("v_sub_f32 %s, 0, %s" % (vr(cb), vr(NEGC(x))))
    t.append("v_add_f32 %s, %s, %s" % (vr(raw), vr(mx), vr(cb)))
    # CMARGIN: checked pass max + 6 (P <= 2^-6 after a raise, the next one when a key beats that maximum by 2^7);
    # optimistic pass max - 60 (P = 2^60 at tile 0's maximum: room for keys 2^67 above it and 2^186 below)
    t.append("v_add_f32 %s, %s, %s" % (vr(cand), "0xc2700000" if FAST else "0x40c00000", vr(raw)))
    t.append("v_cmp_eq_f32_e64 %s, %s, %s" % (sr(sMSK, 2), sr(sNINF), vr(m)))
    t.append("s_or_b64 %s, %s, %s" % (sr(sMSK, 2), sr(sMSK, 2), sr(sRAISE, 2)))
    t.append("v_cmp_gt_f32 vcc, %s, %s" % (vr(cand), vr(m)))
    t.append("s_and_b64 vcc, vcc, %s" % sr(sMSK, 2))
    t.append("v_max_f32 %s, %s, %s" % (vr(cand), sr(sN32K), vr(cand)))
    t.append("v_min_f32 %s, %s, %s" % (vr(cand), sr(sP32K), vr(cand)))
    t.append("v_cndmask_b32 %s, %s, %s, vcc" % (vr(cnew), vr(m), vr(cand)))
    t.append("v_cmp_eq_f32 vcc, %s, %s" % (sr(sNINF), vr(cnew)))
    t.append("v_cndmask_b32_e64 %s, %s, 0, vcc" % (vr(unew), vr(cnew)))
    t.append("v_sub_f32 %s, %s, %s" % (vr(shift), vr(cb), vr(unew)))
    t.append("v_sub_f32 %s, %s, %s" % (vr(alpha), vr(m), vr(unew)))
    t.append("v_exp_f32 %s, %s" % (vr(alpha), vr(alpha)))
    t.append("v_mov_b32 %s, %s" % (vr(m), vr(cnew)))
    for r in sregs:
        t.append("v_add_f32 %s, %s, %s" % (vr(r), vr(shift), vr(r)))
    for i in range(16):
        t.append("v_sub_f32 %s, 0, %s" % (vr(NEGC(x, i)), vr(unew)))
    for i in range(0 if o_is_zero else 48):
        a = (OA if x == 0 else OB) + i
        t.append("v_accvgpr_read_b32 %s, %s" % (vr(tmp), ar(a)))
        t.append("s_nop 0")
        t.append("v_mul_f32 %s, %s, %s" % (vr(tmp), vr(alpha), vr(tmp)))
        t.append("v_accvgpr_write_b32 %s, %s" % (ar(a), vr(tmp)))
    if par is not None:
        for k in range(16):
            t.extend(pair_exp(x, k))
            t.append("s_nop 0")
            t.append(pair_cvt(x, par, k))
    t.append("s_nop 7")
    t.append("s_branch %s_%%=" % ret)


def bias_block(g, x, name, ret):
    """Out-of-line: the key bias (0 / -inf) of tile sT1 added to S'_x (masked keys, keys beyond the video's end)."""
    t = g.tail
    t.append(name + "_%=:")
    t.append("s_lshl_b32 %s, %s, 8" % (sr(sTMP2), sr(sT1)))
    t.append("s_add_u32 %s, %s, %s" % (sr(sTMP2), sr(sTMP2), sr(sMB)))
    t.append("v_add_u32 %s, %s, %s" % (vr(RT), sr(sTMP2), vr(VH16)))
    for n in range(2):
        for gg in range(4):
            t.append("ds_read_b128 %s, %s offset:%d" % (vr(RT + 4 + 4 * (4 * n + gg), 4), vr(RT), 128 * n + 32 * gg))
    t.append("s_waitcnt lgkmcnt(0)")
    for n in range(2):
        for gg in range(4):
            for e_ in range(4):
                t.append("v_add_f32 %s, %s, %s" % (vr(S(x, n, 4 * gg + e_)), vr(RT + 4 + 4 * (4 * n + gg) + e_), vr(S(x, n, 4 * gg + e_))))
    t.append("s_branch %s_%%=" % ret)


FAST = False     # the optimistic pass is being generated (see generate())
SFX = ""         # label suffix of the pass being generated
MASKED = False   # which variant is being generated: generic tile flags (key mask) or "only the last tile can be ragged"


class Item:
    """an atomic group of filler instructions with its issue cost (cycles) and, optionally, the out-of-line blocks it needs"""
    def __init__(self, ops, cost=None, blocks=None):
        self.ops = ops
        self.cost = cost if cost is not None else 4 * len([o for o in ops if not o.endswith(":")])
        self.blocks = blocks

    def emit(self, g):
        for op in self.ops:
            if (ABL & 16) and op.startswith("v_exp_f32"):
                op = op.replace("v_exp_f32", "v_mov_b32")              # timing ablation: no transcendental unit
            if (ABL & 32) and (op.startswith("v_cvt_pk") or op.startswith("v_or")):
                continue                                               # timing ablation: no packs, no OR chain
            if (ABL & 128) and (op.startswith("s_cmp_eq_u32") or op.startswith("s_cbranch_scc1 B")):
                continue                                               # timing ablation: no key-bias check
            if (ABL & 256) and (op.startswith("v_cmp_ne_u32 vcc") or op.startswith("s_cbranch_vccnz B") or op.startswith("v_and_b32")):
                continue                                               # timing ablation: no OR test (the chain stays)
            g.e(op)
        if self.blocks is not None:
            self.blocks()


def tile0_prelude(g, x):
    """no-mask variant, pre-iteration only: tile 0's key bias (a video shorter than one tile) and the rows' first constant"""
    sb, rb0 = g.site(), g.site()
    for op in ["s_cmp_eq_u32 %s, %s" % (sr(sT1), sr(sLAST)), "s_cbranch_scc1 B%s_%%=" % sb, "R%s_%%=:" % sb,
               "v_cmp_eq_f32 vcc, %s, %s" % (sr(sNINF), vr(VM[x])), "s_cbranch_vccnz B%s_%%=" % rb0, "R%s_%%=:" % rb0]:
        g.e(op)
    bias_block(g, x, "B" + sb, "R" + sb)
    rebase_block(g, x, None, "B" + rb0, "R" + rb0, o_is_zero=True)


def job_items(g, x, par, tile0):
    """softmax(tile sT1, row block x) -> P[par] as an ordered list of Items: the checks, then per pair of keys two exp2 and
    (one pair behind: the transcendental unit's result needs a wait state) the pack, the running OR, the OR test.
    Key-mask variant: per-tile flag and "row without a constant" checks in every job.  No-mask variant: only the last
    tile can need the key bias, and every row has its constant after tile 0 (tile0_prelude does both for tile 0; the job of
    tile 0 then carries two nops in place of the check so that every job has the same cost profile - the B job is split
    over two iterations and both halves must agree on where)."""
    items = []
    if FAST:
        pass          # nothing is checked per tile (the ragged last tile is corrected in the epilogue; see generate())
    elif not MASKED and tile0:
        items.append(Item(["s_nop 0", "s_nop 0"]))
    else:
        sb = g.site()
        if MASKED:
            ops = ["s_lshr_b32 %s, %s, 5" % (sr(sTMP), sr(sT1)),
                   "v_readlane_b32 %s, %%[flags], %s" % (sr(sTMP2), sr(sTMP)),
                   "s_bitcmp1_b32 %s, %s" % (sr(sTMP2), sr(sT1))]
        else:
            ops = ["s_cmp_eq_u32 %s, %s" % (sr(sT1), sr(sLAST))]
        ops += ["s_cbranch_scc1 B%s_%%=" % sb, "R%s_%%=:" % sb]
        items.append(Item(ops, blocks=lambda: bias_block(g, x, "B" + sb, "R" + sb)))
    if MASKED:
        rb0 = g.site()
        items.append(Item(["v_cmp_eq_f32 vcc, %s, %s" % (sr(sNINF), vr(VM[x])), "s_cbranch_vccnz B%s_%%=" % rb0, "R%s_%%=:" % rb0],
                          blocks=lambda: rebase_block(g, x, None, "B" + rb0, "R" + rb0)))
    acc = VOR0
    for k in range(16):
        e2 = pair_exp(x, k)
        items.append(Item([e2[0]], 8))
        items.append(Item([e2[1]], 8))
        if k > 0:
            items.append(Item([pair_cvt(x, par, k - 1)]))
        if FAST:
            continue
        if k == 4:
            items.append(Item(["v_or3_b32 %s, %s, %s, %s" % (vr(acc), vr(preg_of_pair(x, par, 0)), vr(preg_of_pair(x, par, 1)), vr(preg_of_pair(x, par, 2)))]))
        elif k in (6, 8, 10, 12, 14):
            items.append(Item(["v_or3_b32 %s, %s, %s, %s" % (vr(acc), vr(acc), vr(preg_of_pair(x, par, k - 3)), vr(preg_of_pair(x, par, k - 2)))]))
    items.append(Item([pair_cvt(x, par, 15)]))
    if FAST:
        return items
    rb = g.site()
    items.append(Item(["v_or3_b32 %s, %s, %s, %s" % (vr(acc), vr(acc), vr(preg_of_pair(x, par, 13)), vr(preg_of_pair(x, par, 14))),
                       "v_or_b32 %s, %s, %s" % (vr(acc), vr(acc), vr(preg_of_pair(x, par, 15))),
                       "v_and_b32 %s, 0x40004000, %s" % (vr(acc), vr(acc)),
                       "v_cmp_ne_u32 vcc, 0, %s" % vr(acc),
                       "s_cbranch_vccnz B%s_%%=" % rb,
                       "R%s_%%=:" % rb], blocks=lambda: rebase_block(g, x, par, "B" + rb, "R" + rb)))
    return items


BUDGET = 24      # cycles of filler issue a v_mfma_f32_32x32x16_bf16 gap hides (32 minus the MFMA's own 8)


def distribute_quota(items, quota):
    """items (ordered) over len(quota) MFMA gaps: gap i takes items until it holds quota[i] exp2 (and the non-exp2 items
    that follow them); items before the first exp2 go to gap 0, what is left after the last quota to the next gap."""
    n = len(quota)
    out = [[] for _ in range(n)]
    idx = 0
    last_q = max(i for i in range(n) if quota[i] > 0)
    for sl in range(n):
        taken = 0
        while idx < len(items):
            is_exp = items[idx].ops[0].startswith("v_exp")
            if is_exp and taken >= quota[sl]:
                break
            if not is_exp and sl > last_q:
                pass
            out[sl].append(items[idx])
            taken += 1 if is_exp else 0
            idx += 1
            if sl <= last_q and taken >= quota[sl] and idx < len(items) and items[idx].ops[0].startswith("v_exp"):
                break
        if sl == last_q:
            # the exp2 are placed; the remaining items (last pack, OR test) go to the next gap
            pass
    assert idx == len(items) and sum(quota) == sum(1 for it in items if it.ops[0].startswith("v_exp")), (idx, len(items), sum(quota))
    return out


QUOTA_A = [1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2] + [2, 2, 2, 2, 2, 0, 0, 0]      # steps 2 + 3
QUOTA_B = [1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1] + [2, 2, 2, 2, 2, 0, 0, 0]      # step 4 + the next step 1


def distribute(items, fixed):
    """items (ordered) over len(fixed) MFMA gaps; fixed[i] = issue cycles gap i already carries.  Every gap gets a share
    of what is left in proportion to its room, so that an over-full phase is over-full evenly."""
    n = len(fixed)
    out = [[] for _ in range(n)]
    idx = 0
    remaining = sum(it.cost for it in items)
    rooms = [max(BUDGET - f, 4) for f in fixed]
    for sl in range(n):
        if sl == n - 1:
            out[sl] = items[idx:]
            break
        share = remaining * rooms[sl] / float(sum(rooms[sl:]))
        taken = 0
        while idx < len(items) and taken + items[idx].cost / 2.0 <= share:
            out[sl].append(items[idx])
            taken += items[idx].cost
            idx += 1
        remaining -= taken
    return out


def v_reads(f):
    """the two transposed reads of V^T fragment f = (n*2+s)*2 + d"""
    n, s_, d = f >> 2, (f >> 1) & 1, f & 1
    off = (32 * n + 16 * s_) * 128
    a = VF + 4 * f
    return ["ds_read_b64_tr_b16 %s, %s offset:%d" % (ar(a, 2), vr(VVA + d), off),
            "ds_read_b64_tr_b16 %s, %s offset:%d" % (ar(a + 2, 2), vr(VVA + d), off + 1024)]


def k_read(gi):
    n, ks = gi // 4, gi % 4
    return "ds_read_b128 %s, %s offset:%d" % (ar(KF + 4 * gi, 4), vr(VKA + ks), 4096 * n)


def k_addr_ops(ring):
    return ["v_add_u32 %s, %s, %%[koff]" % (vr(VKA), sr(sTMP)),
            "v_xor_b32 %s, 32, %s" % (vr(VKA + 1), vr(VKA)),
            "v_xor_b32 %s, 64, %s" % (vr(VKA + 2), vr(VKA)),
            "v_xor_b32 %s, 0x60, %s" % (vr(VKA + 3), vr(VKA))]


def dma_piece(is_k, piece, ring):
    """one LDS-DMA piece (1 KiB, 8 key rows) of this wave: K / V piece 0 / 1 into ring slot `ring` (an SGPR offset).
    No instruction offset: it would be added to the LDS address as well as to the source address."""
    base = (sKB0, sKB1, sVB0, sVB1)[(0 if is_k else 2) + piece]
    voff = ("%[dk0]", vr(VDK1), "%[dv0]", vr(VDV1))[(0 if is_k else 2) + piece]
    return ["s_add_u32 m0, %s, %s" % (sr(base), sr(ring)), "s_nop 0",
            "buffer_load_dwordx4 %s, %s, %s offen lds" % (voff, sr(KD if is_k else VD, 4), sr(sKSO if is_k else sVSO))]


DMA_COST = 16

ABL = 0      # timing ablations (diagnostic library only; WRONG results): 1 no softmax work, 2 no LDS-DMA, 4 no fragment reads, 8 no barrier,
             # 16 exp2 -> mov, 32 no packs / OR chain, 64 the proportional filler distribution instead of the exp2 quotas,
             # 512 / 1024 (optimistic pass) no softmax items beside the S' MFMAs / beside the P.V MFMAs


def iteration(g, par, do_pv, do_s):
    """One iteration t (see the kernel header); P(t, .) lives in P[par], softmax(t+1, .) writes P[par ^ 1].
    do_pv False: the pre-iteration t = -1; do_s False: the last tile.
    LDS-DMA: one piece per step, so that the four waves of a block never issue sixteen pieces at once behind the barrier:
    step 4 K(t+4) piece 0; next iteration's steps 1 / 2 / 3: K(t+4) piece 1, V(t+3) piece 0, V(t+3) piece 1 - i.e. THIS
    iteration's steps 1 / 2 / 3 carry K(t+3) piece 1 -> ring slot t % 3, V(t+2) pieces -> ring slot (t+2) % 3."""
    e = g.e
    pre = not do_pv
    late_dma = do_pv and do_s and not (ABL & 2)
    gs = Gen() if (ABL & 1) else g                   # ablation 1: the softmax items and their rare blocks go nowhere

    def emit_items(lst):
        if not (ABL & 1):
            for it in lst:
                it.emit(g)

    # the B job of tile t (second part here, first part in the previous iteration's step 4) and of tile t+1
    fixed4 = [(DMA_COST + 4 if i == 0 else 0) + (4 if i < 8 else 0) + (12 if i == 10 else 0) + (28 if i == 11 else 0) for i in range(12)]
    fixed1 = [8 + (4 if i == 0 else 0) + (DMA_COST if i == 7 else 0) for i in range(8)]
    fixed2 = [(4 if i % 3 == 0 else 0) + (DMA_COST if i == 1 else 0) for i in range(12)]
    fixed3 = [(20 if i == 6 else 0) + (DMA_COST + 8 if i == 7 else 0) for i in range(8)]
    def dist(items, fixed, quota):
        return distribute(items, fixed) if (ABL & 64) else distribute_quota(items, quota)

    distB_prev = dist(job_items(gs, 1, par, False), fixed4 + fixed1, QUOTA_B) if do_pv else None
    # ---------------- step 1: S'(t+1, A) || softmax(t, B) second part; V(t) fragments ----------------
    if do_pv:
        e("s_waitcnt lgkmcnt(0)")                    # K(t+1) fragments (requested in the previous step 4, slots 0..7)
    for i in range(8):
        if do_s:
            e(s_mfma(0, i))
        if do_pv:
            for op in ([] if (ABL & 4) else v_reads(i)):
                e(op)
            if i == 6 and do_s:
                # the ring moves on here (a gap that carries only fragment reads): (t, t+1, t+2) % 3 from now on
                e("s_mov_b32 %s, %s" % (sr(sTMP), sr(sR0)))
                e("s_mov_b32 %s, %s" % (sr(sR0), sr(sR1)))
                e("s_mov_b32 %s, %s" % (sr(sR1), sr(sR2)))
                e("s_mov_b32 %s, %s" % (sr(sR2), sr(sTMP)))
                e("s_add_u32 %s, %s, 0x2000" % (sr(sVSO), sr(sVSO)))
            if i == 7 and late_dma:
                for op in dma_piece(True, 1, sR0):
                    e(op)
            emit_items(distB_prev[12 + i])
    if do_s and pre:
        e("s_nop 15"); e("s_nop 15"); e("s_nop 15")       # pre-iteration: no MFMAs behind which S'(0, A) could settle
        if not MASKED and not (ABL & 1):
            tile0_prelude(g, 0)
    # ---------------- step 2: O_A += V(t)^T P(t, A)^T || softmax(t+1, A) first part ----------------
    distA = dist(job_items(gs, 0, par ^ 1, pre), fixed2 + fixed3, QUOTA_A) if do_s else None
    for i in range(12):
        if do_pv:
            if i % 3 == 0:
                e("s_waitcnt lgkmcnt(%d)" % (12 - 4 * (i // 3)))
            e(pv_mfma(0, par, i))
        if i == 1 and late_dma:
            for op in dma_piece(False, 0, sR2):
                e(op)
        if do_s:
            emit_items(distA[i])
    # ---------------- step 3: S'(t+1, B) || softmax(t+1, A) second part ----------------
    if do_s:
        for i in range(8):
            e(s_mfma(1, i))
            if i == 6:
                e("s_add_u32 %s, %s, %s" % (sr(sTMP), sr(sKB), sr(sR2)))
                for op in k_addr_ops(None):
                    e(op)
            if i == 7 and late_dma:
                for op in dma_piece(False, 1, sR2):
                    e(op)
            emit_items(distA[12 + i])
        if pre:
            e("s_nop 15"); e("s_nop 15"); e("s_nop 15")
            if not MASKED and not (ABL & 1):
                tile0_prelude(g, 1)
        e("s_waitcnt vmcnt(4)")                      # this wave's pieces of K(t+2) / V(t+1) have landed
        if not (ABL & 8):
            e("s_barrier")
    # ---------------- step 4: O_B += V(t)^T P(t, B)^T || softmax(t+1, B) first part; K(t+2) fragments; DMA ----------------
    distB = dist(job_items(gs, 1, par ^ 1, pre), fixed4 + fixed1, QUOTA_B) if do_s else None
    for i in range(12):
        if do_pv:
            e(pv_mfma(1, par, i))
        if do_s:
            if i == 0:
                e("s_add_u32 %s, %s, 0x2000" % (sr(sKSO), sr(sKSO)))
                if not (ABL & 2):
                    for op in dma_piece(True, 0, sR1):
                        e(op)
            if i < 8 and not (ABL & 4):
                e(k_read(i))
            emit_items(distB[i])
            if i == 10:
                # next iteration's V read addresses: ring slot (t+1) % 3
                e("s_add_u32 %s, %s, %s" % (sr(sTMP), sr(sVB), sr(sR1)))
                e("v_add_u32 %s, %s, %%[voff]" % (vr(VVA), sr(sTMP)))
                e("v_xor_b32 %s, 64, %s" % (vr(VVA + 1), vr(VVA)))
            if i == 11:
                e("s_add_u32 %s, %s, 1" % (sr(sT1), sr(sT1)))


LATE_MEM = False     # A/B (measured slower: 0.980 vs 0.956 of the checked pass): inside an MFMA gap vector instructions first, LDS reads / LDS-DMA last
VF2 = 196                            # second V^T fragment buffer (optimistic pass: fragments of tile t+1 are read while tile t's are in use)
KAS, VAS = 244, 202                  # optimistic pass: read addresses of the three ring slots, K v[244:255] (slot * 4 + ks), V v[202:207] (slot * 2 + d)
QUOTA_FA = [2, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2] + [2, 2, 2, 2, 1, 0, 0, 0]      # exp2 per gap, job A: steps 2 + 3
QUOTA_FB = [0, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 0] + [1, 2, 2, 2, 2, 2, 2, 2]      # job B: step 4 + the next step 1


def iteration_fast(g, par, ring, do_pv, do_s):
    """The optimistic pass's iteration t (t % 2 = par, t % 3 = ring: six bodies per loop trip, every ring offset an
    immediate).  Differences from iteration(): nothing is checked; V^T fragments are double-buffered (first half of tile
    t+1's in step 4, second half in the next step 1); the read addresses of the three ring slots sit in registers."""
    e = g.e
    pre = not do_pv
    r0, r1, r2 = ring, (ring + 1) % 3, (ring + 2) % 3
    vfb, vfn = (VF, VF2) if par == 0 else (VF2, VF)          # V^T fragments of tile t / where tile t+1's go

    def emit_items(lst):
        if not (ABL & 1):
            for it in lst:
                it.emit(g)

    def pv(x, i):
        gq, w = i // 3, i % 3
        p = vr(P(par, x, gq >> 1, gq & 1), 4)
        if w < 2:
            return mfma(ar(O(x, w), 16), ar(vfb + 4 * (2 * gq + w), 4), p, ar(O(x, w), 16))
        return mfma(ar(O(x, 2), 16), ar(ONES, 4), p, ar(O(x, 2), 16))

    def vread(buf, slot, f):
        n, s_, d = f >> 2, (f >> 1) & 1, f & 1
        off = (32 * n + 16 * s_) * 128
        a = buf + 4 * f
        return ["ds_read_b64_tr_b16 %s, %s offset:%d" % (ar(a, 2), vr(VAS + 2 * slot + d), off),
                "ds_read_b64_tr_b16 %s, %s offset:%d" % (ar(a + 2, 2), vr(VAS + 2 * slot + d), off + 1024)]

    def kread(slot, gi):
        n, ks = gi // 4, gi % 4
        return "ds_read_b128 %s, %s offset:%d" % (ar(KF + 4 * gi, 4), vr(KAS + 4 * slot + ks), 4096 * n)

    def dma(is_k, piece, slot):
        base = (sKB0, sKB1, sVB0, sVB1)[(0 if is_k else 2) + piece]
        voff = ("%[dk0]", vr(VDK1), "%[dv0]", vr(VDV1))[(0 if is_k else 2) + piece]
        return ["s_add_u32 m0, %s, 0x%x" % (sr(base), slot * 8192), "s_nop 0",
                "buffer_load_dwordx4 %s, %s, %s offen lds" % (voff, sr(KD if is_k else VD, 4), sr(sKSO if is_k else sVSO))]

    def gap_with_dma(dma_ops3, items):
        """M0 write, one of the gap's own vector instructions (the wait state the LDS-DMA needs after an M0 write), the load"""
        e(dma_ops3[0])
        rest = list(items)
        if rest and not (ABL & 1) and len(rest[0].ops) == 1:
            rest.pop(0).emit(g)
        else:
            e(dma_ops3[1])
        e(dma_ops3[2])
        emit_items(rest)

    late_dma = do_pv and do_s
    distB_prev = distribute_quota(job_items(g, 1, par, False), QUOTA_FB) if do_pv else None
    # ---------------- step 1: S'(t+1, A) || softmax(t, B) second part; second half of V(t)'s fragments ----------------
    if do_pv:
        e("s_waitcnt lgkmcnt(8)")                    # K(t+1) fragments (the eight younger reads are V(t)'s first half)
    for i in range(8):
        if do_s:
            e(s_mfma(0, i))
        if do_pv:
            if not (ABL & 512) and not LATE_MEM:
                pass
            if LATE_MEM and not (ABL & 512):
                emit_items(distB_prev[12 + i])
            if i < 4 and not (ABL & 4):
                for op in vread(vfb, r0, 4 + i):
                    e(op)
            if i == 6 and do_s:
                e("s_add_u32 %s, %s, 0x2000" % (sr(sVSO), sr(sVSO)))
            if not LATE_MEM and not (ABL & 512):
                emit_items(distB_prev[12 + i])
    if do_s and pre:
        e("s_nop 15"); e("s_nop 15"); e("s_nop 15")
        tile0_prelude(g, 0)
    # ---------------- step 2: O_A += V(t)^T P(t, A)^T || softmax(t+1, A) first part ----------------
    distA = distribute_quota(job_items(g, 0, par ^ 1, pre), QUOTA_FA) if do_s else None
    for i in range(12):
        if do_pv:
            if i == 0:
                e("s_waitcnt lgkmcnt(0)")            # V(t)^T fragments (second half asked for in step 1's gaps 0..3)
            e(pv(0, i))
        if LATE_MEM and do_s and not (ABL & 1024):
            emit_items(distA[i])
        if not LATE_MEM and do_s and not (ABL & 1024):
            emit_items(distA[i])
    # ---------------- step 3: S'(t+1, B) || softmax(t+1, A) second part ----------------
    if do_s:
        for i in range(8):
            e(s_mfma(1, i))
            if LATE_MEM and not (ABL & 512):
                emit_items(distA[12 + i])
            if i >= 5 and late_dma and not (ABL & 2):
                # the rest of the batch whose first piece went out in the previous step 4: K(t+3) piece 1 -> slot t % 3,
                # V(t+2) pieces -> slot (t+2) % 3 (any time in steps 1..3 is safe; these gaps carry no exp2)
                gap_with_dma(dma(True, 1, r0) if i == 5 else dma(False, i - 6, r2), [] if (ABL & 512) else distA[12 + i])
            elif not LATE_MEM and not (ABL & 512):
                emit_items(distA[12 + i])
        if pre:
            e("s_nop 15"); e("s_nop 15"); e("s_nop 15")
            tile0_prelude(g, 1)
        e("s_waitcnt vmcnt(4)")
        if not (ABL & 8):
            e("s_barrier")
    # ---------------- step 4: O_B += V(t)^T P(t, B)^T || softmax(t+1, B) first part; K(t+2), first half of V(t+1) ----------------
    distB = distribute_quota(job_items(g, 1, par ^ 1, pre), QUOTA_FB) if do_s else None
    for i in range(12):
        if do_pv:
            e(pv(1, i))
        if do_s:
            if LATE_MEM and not (ABL & 1024):
                emit_items(distB[i])
            if i == 0:
                d3 = dma(True, 0, r1)
                if not (ABL & 2):
                    e(d3[0])
                e("s_add_u32 %s, %s, 0x2000" % (sr(sKSO), sr(sKSO)))      # (also the wait state between the M0 write and the load)
                if not (ABL & 2):
                    e(d3[2])
            if not (ABL & 4):
                if i < 8:
                    e(kread(r2, i))
                else:
                    for op in vread(vfn, r1, i - 8):
                        e(op)
            if not LATE_MEM and not (ABL & 1024):
                emit_items(distB[i])
            if i == 11:
                e("s_add_u32 %s, %s, 1" % (sr(sT1), sr(sT1)))


def tile_loop_fast(g):
    e = g.e
    iteration_fast(g, 1, 2, False, True)             # pre-iteration t = -1
    g.label("LOOPF")
    for k in range(6):
        e("s_cmp_ge_i32 %s, %s" % (sr(sT1), sr(sNT)))
        e("s_cbranch_scc1 LASTF%d_%%=" % k)
        iteration_fast(g, k % 2, k % 3, True, True)
    e("s_branch LOOPF_%=")
    for k in range(6):
        g.label("LASTF%d" % k)
        iteration_fast(g, k % 2, k % 3, True, False)
        if k < 5:
            e("s_branch EPIF_%=")
    g.label("EPIF")
    e("s_nop 15"); e("s_nop 15"); e("s_nop 15")
    e("s_waitcnt vmcnt(0)")


def tile_loop(g):
    """pre-iteration, the tile loop (two iterations per trip: the P buffers alternate) and the last tile; ends at the epilogue"""
    e = g.e
    iteration(g, 1, False, True)
    g.label("LOOP" + SFX)
    e("s_cmp_ge_i32 %s, %s" % (sr(sT1), sr(sNT)))    # t + 1 >= ntiles: t (even) is the last tile
    e("s_cbranch_scc1 LAST0%s_%%=" % SFX)
    iteration(g, 0, True, True)
    e("s_cmp_ge_i32 %s, %s" % (sr(sT1), sr(sNT)))
    e("s_cbranch_scc1 LAST1%s_%%=" % SFX)
    iteration(g, 1, True, True)
    e("s_branch LOOP%s_%%=" % SFX)
    g.label("LAST0" + SFX)
    iteration(g, 0, True, False)
    e("s_branch EPI%s_%%=" % SFX)
    g.label("LAST1" + SFX)
    iteration(g, 1, True, False)
    g.label("EPI" + SFX)
    e("s_nop 15"); e("s_nop 15"); e("s_nop 15")      # the last MFMAs' results
    e("s_waitcnt vmcnt(0)")                          # no LDS-DMA piece may land after this block has left (or restarts)


OUTV = 32        # the optimistic pass parks its packed outputs in v[32:63] (S' is dead by then): [row block][d block][g] x 2 dwords


def epilogue(g, check):
    """O / l -> bf16.  check False: stored at once (rows beyond the video are dropped by the buffer bounds).
    check True (optimistic pass): kept in registers, and v27 collects, per lane, whether anything is inf / NaN."""
    e = g.e
    bad = VOR1
    if check:
        e("v_mov_b32 %s, 0" % vr(bad))
    for x in range(2):
        inv = RT
        e("v_accvgpr_read_b32 %s, %s" % (vr(inv), ar(O(x, 2))))
        e("s_nop 0")
        if check:
            # keys beyond the end of the video in a ragged last tile: their K rows are zeros (buffer bounds), so each of
            # them added exactly bf16(exp2(-c)) to l and nothing to O (their V rows are zeros too): take it out again
            e("v_exp_f32 %s, %s" % (vr(RT + 1), vr(NEGC(x))))
            e("s_nop 0")
            e("v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(RT + 1), vr(RT + 1), vr(RT + 1)))
            e("v_lshlrev_b32 %s, 16, %s" % (vr(RT + 1), vr(RT + 1)))
            e("v_fma_f32 %s, %s, %s, %s" % (vr(inv), sr(sNPADN), vr(RT + 1), vr(inv)))
            # l itself: 1 / inf = 0 would hide an overflow - and so would l in [2^126, 2^128): its reciprocal is a denormal, which
            # v_rcp_f32 flushes to 0 (tools/fuzz_attn_w64.py found it: a key 66 above tile 0's maximum, every output of that row
            # zero, nothing non-finite).  l * 2^8 overflows from 2^120 on: such a row sends the block to the checked pass too.
            e("v_mul_f32 %s, 0x43800000, %s" % (vr(RT + 1), vr(inv)))
            e("v_fma_f32 %s, %s, 0, %s" % (vr(bad), vr(RT + 1), vr(bad)))
        e("v_rcp_f32 %s, %s" % (vr(inv), vr(inv)))
        e("s_nop 0")
        for d in range(2):
            for gg in range(4):
                t0 = RT + 2 + 8 * ((4 * d + gg) % 4)
                for e_ in range(4):
                    e("v_accvgpr_read_b32 %s, %s" % (vr(t0 + e_), ar(O(x, d) + 4 * gg + e_)))
                e("s_nop 0")
                for e_ in range(4):
                    e("v_mul_f32 %s, %s, %s" % (vr(t0 + e_), vr(inv), vr(t0 + e_)))
                if check:
                    # x * 0 is 0 for a finite x and NaN for inf / NaN; the NaN survives the additions
                    for e_ in range(4):
                        e("v_fma_f32 %s, %s, 0, %s" % (vr(bad), vr(t0 + e_), vr(bad)))
                    o0 = OUTV + 2 * ((x * 2 + d) * 4 + gg)
                    e("v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(o0), vr(t0), vr(t0 + 1)))
                    e("v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(o0 + 1), vr(t0 + 2), vr(t0 + 3)))
                else:
                    e("v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(t0 + 4), vr(t0), vr(t0 + 1)))
                    e("v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(t0 + 5), vr(t0 + 2), vr(t0 + 3)))
                    so = "0" if x == 0 else sr(sOB)
                    e("buffer_store_dwordx2 %s, %%[ooff], %s, %s offen offset:%d" % (vr(t0 + 4, 2), sr(OD, 4), so, 64 * d + 16 * gg))


def store_parked(g):
    for x in range(2):
        for d in range(2):
            for gg in range(4):
                o0 = OUTV + 2 * ((x * 2 + d) * 4 + gg)
                so = "0" if x == 0 else sr(sOB)
                g.e("buffer_store_dwordx2 %s, %%[ooff], %s, %s offen offset:%d" % (vr(o0, 2), sr(OD, 4), so, 64 * d + 16 * gg))


def generate():
    """Key-mask variant: one pass, every tile checked (flags, rows without a constant, OR test).
    No-mask variant: an OPTIMISTIC pass first - the row constant is set once, 60 below tile 0's maximum, and nothing is
    checked per tile (no OR chain, no test: 22 fewer vector instructions per tile); its outputs are kept in registers and
    tested for inf / NaN, which is what a key more than 2^67 above tile 0's maximum - or a sum beyond the fp32 range -
    leaves behind.  If any wave of the block saw one, the whole block runs again in the checked form, whose arithmetic for
    rows that never raise their constant is ... the checked form's (a block's result is then the checked pass's for every
    row, so a row's bits do not depend on which pass produced them only if BOTH passes agree on untroubled rows: they do,
    P differs by the exact factor 2^66 between the passes and O / l is scale-free up to fp32 rounding of the same sums)."""
    global FAST, SFX
    g = Gen()
    e = g.e
    # ---------------- once per block ----------------
    for d, lo, hi, nrec in ((QD, "qlo", "qhi", "nrec"), (KD, "klo", "khi", "nrec"), (VD, "vlo", "vhi", "nrec"), (OD, "olo", "ohi", "nreco")):
        e("s_mov_b32 %s, %%[%s]" % (sr(d), lo))
        e("s_and_b32 %s, %%[%s], 0xffff" % (sr(d + 1), hi))
        e("s_mov_b32 %s, %%[%s]" % (sr(d + 2), nrec))
        e("s_mov_b32 %s, 0x00020000" % sr(d + 3))
    e("s_mov_b32 %s, 0xff800000" % sr(sNINF))
    e("s_mov_b32 %s, 0xc7000000" % sr(sN32K))
    e("s_mov_b32 %s, 0x47000000" % sr(sP32K))
    e("s_movk_i32 %s, 0x1000" % sr(sC4096))
    e("s_mov_b32 %s, %%[ntiles]" % sr(sNT))
    e("s_mov_b32 %s, %%[mb]" % sr(sMB))
    e("s_mov_b32 %s, %%[kb]" % sr(sKB))
    e("s_mov_b32 %s, %%[vb]" % sr(sVB))
    e("s_mov_b32 %s, %%[orowb]" % sr(sOB))
    e("s_lshl_b32 %s, %%[wave], 11" % sr(sTMP))                     # 2 w * 1024
    e("s_add_u32 %s, %s, %s" % (sr(sKB0), sr(sKB), sr(sTMP)))
    e("s_add_u32 %s, %s, %s" % (sr(sVB0), sr(sVB), sr(sTMP)))
    e("s_add_u32 %s, %s, 0x400" % (sr(sKB1), sr(sKB0)))
    e("s_add_u32 %s, %s, 0x400" % (sr(sVB1), sr(sVB0)))
    e("s_mov_b32 %s, %%[lastf]" % sr(sLAST))
    e("s_lshl_b32 %s, %%[wave], 2" % sr(sWV4))
    e("s_mov_b32 %s, %%[npadn]" % sr(sNPADN))          # -(keys beyond the end in a ragged last tile that is not tile 0), as a float
    e("s_mov_b32 %s, %%[mode0]" % sr(sMODE))              # 0: optimistic pass first; 1: the checked pass only (A/B switch)
    # Q fragments straight into AGPRs (rows beyond the video read as zeros: buffer bounds)
    for ks in range(4):
        e("buffer_load_dwordx4 %s, %%[qoff], %s, 0 offen offset:%d" % (ar(Qf(0, ks), 4), sr(QD, 4), 32 * ks))
    for ks in range(4):
        e("buffer_load_dwordx4 %s, %%[qoff], %s, %s offen offset:%d" % (ar(Qf(1, ks), 4), sr(QD, 4), sr(sC4096), 32 * ks))
    e("v_xor_b32 %s, 64, %%[dk0]" % vr(VDK1))
    e("v_add_u32 %s, 0x400, %s" % (vr(VDK1), vr(VDK1)))
    e("v_add_u32 %s, 0x400, %%[dv0]" % vr(VDV1))
    e("v_mov_b32 %s, 0x3f803f80" % vr(RT))
    for i in range(4):
        e("v_accvgpr_write_b32 %s, %s" % (ar(ONES + i), vr(RT)))
    e("v_mbcnt_lo_u32_b32 %s, -1, 0" % vr(VH16))
    e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (vr(VH16), vr(VH16)))
    e("v_lshrrev_b32 %s, 5, %s" % (vr(VH16), vr(VH16)))
    e("v_lshlrev_b32 %s, 4, %s" % (vr(VH16), vr(VH16)))
    # ---------------- a pass starts here ----------------
    g.label("RESTART")
    # prologue DMA: K(0), V(0), K(1) | V(1), K(2)
    def dma_tile(is_k, tile, slot):
        base, desc, v0, v1 = (sKB0, KD, "%[dk0]", vr(VDK1)) if is_k else (sVB0, VD, "%[dv0]", vr(VDV1))
        e("s_mov_b32 %s, 0x%x" % (sr(sTMP), tile * 8192))
        e("s_add_u32 m0, %s, 0x%x" % (sr(base), slot * 8192))
        e("s_nop 0")
        e("buffer_load_dwordx4 %s, %s, %s offen lds" % (v0, sr(desc, 4), sr(sTMP)))
        e("s_add_u32 m0, m0, 0x400")
        e("s_nop 0")
        e("buffer_load_dwordx4 %s, %s, %s offen lds" % (v1, sr(desc, 4), sr(sTMP)))
    dma_tile(True, 0, 0); dma_tile(False, 0, 0); dma_tile(True, 1, 1); dma_tile(False, 1, 1); dma_tile(True, 2, 2)
    for i in range(96):
        e("v_accvgpr_write_b32 %s, 0" % ar(i))
    for i in range(32):
        e("v_mov_b32 %s, 0" % vr(NEGC(0, i)))
    e("v_mov_b32 %s, %s" % (vr(VM[0]), sr(sNINF)))
    e("v_mov_b32 %s, %s" % (vr(VM[1]), sr(sNINF)))
    # ring for the pre-iteration t = -1: (t, t+1, t+2) % 3 = (2, 0, 1)
    e("s_mov_b32 %s, 0x4000" % sr(sR0))
    e("s_mov_b32 %s, 0" % sr(sR1))
    e("s_mov_b32 %s, 0x2000" % sr(sR2))
    e("s_mov_b32 %s, 0x4000" % sr(sKSO))             # (t + 3) * 8192 at t = -1; step 4 advances it to K(3)
    e("s_mov_b32 %s, 0x2000" % sr(sVSO))             # (t + 2) * 8192
    e("s_mov_b32 %s, 0" % sr(sT1))
    e("s_waitcnt vmcnt(4)")                          # Q, K(0), V(0), K(1) have landed (this wave's pieces)
    e("s_barrier")
    e("s_mov_b32 %s, %s" % (sr(sTMP), sr(sKB)))      # K(0) fragments from ring slot 0
    for op in k_addr_ops(None):
        e(op)
    for gi in range(8):
        e(k_read(gi))
    e("s_waitcnt lgkmcnt(0)")
    if not MASKED:
        e("s_cmp_eq_u32 %s, 1" % sr(sMODE))
        e("s_cbranch_scc1 CHECKED_%=")
        # ---------------- the optimistic pass ----------------
        FAST, SFX = True, "F"
        # read addresses of the three ring slots (the K ones are recomputed here on purpose: the checked pass's own
        # address registers are different ones)
        for slot in range(3):
            e("s_add_u32 %s, %s, 0x%x" % (sr(sTMP), sr(sKB), slot * 8192))
            e("v_add_u32 %s, %s, %%[koff]" % (vr(KAS + 4 * slot), sr(sTMP)))
            e("v_xor_b32 %s, 32, %s" % (vr(KAS + 4 * slot + 1), vr(KAS + 4 * slot)))
            e("v_xor_b32 %s, 64, %s" % (vr(KAS + 4 * slot + 2), vr(KAS + 4 * slot)))
            e("v_xor_b32 %s, 0x60, %s" % (vr(KAS + 4 * slot + 3), vr(KAS + 4 * slot)))
            e("s_add_u32 %s, %s, 0x%x" % (sr(sTMP), sr(sVB), slot * 8192))
            e("v_add_u32 %s, %s, %%[voff]" % (vr(VAS + 2 * slot), sr(sTMP)))
            e("v_xor_b32 %s, 64, %s" % (vr(VAS + 2 * slot + 1), vr(VAS + 2 * slot)))
        tile_loop_fast(g)
        epilogue(g, True)
        # any lane of any wave of the block with an inf / NaN?  (the K ring is free by now: one dword per wave at its base)
        e("v_cmp_u_f32 vcc, %s, %s" % (vr(VOR1), vr(VOR1)))
        e("s_cmp_lg_u64 vcc, 0")
        e("s_cselect_b32 %s, 1, 0" % sr(sTMP))
        e("v_mov_b32 %s, %s" % (vr(RT), sr(sTMP)))
        e("s_add_u32 %s, %s, %s" % (sr(sTMP2), sr(sKB), sr(sWV4)))
        e("v_mov_b32 %s, %s" % (vr(RT + 1), sr(sTMP2)))
        e("ds_write_b32 %s, %s" % (vr(RT + 1), vr(RT)))
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        e("v_mov_b32 %s, %s" % (vr(RT + 1), sr(sKB)))
        e("ds_read_b128 %s, %s" % (vr(RT + 4, 4), vr(RT + 1)))
        e("s_waitcnt lgkmcnt(0)")
        e("v_or3_b32 %s, %s, %s, %s" % (vr(RT), vr(RT + 4), vr(RT + 5), vr(RT + 6)))
        e("v_or_b32 %s, %s, %s" % (vr(RT), vr(RT), vr(RT + 7)))
        e("s_nop 0")
        e("v_readfirstlane_b32 %s, %s" % (sr(sTMP), vr(RT)))
        e("s_barrier")                                   # every wave has read the four words before the ring is refilled
        if ABL:
            e("s_mov_b32 %s, 0" % sr(sTMP))              # timing ablations: wrong values on purpose, never a second pass
        e("s_cmp_eq_u32 %s, 0" % sr(sTMP))
        e("s_cbranch_scc1 STOREF_%=")
        e("s_mov_b32 %s, 1" % sr(sMODE))
        e("s_branch RESTART_%=")
        g.label("STOREF")
        store_parked(g)
        e("s_waitcnt vmcnt(0)")
        e("s_branch END_%=")
        g.label("CHECKED")
    # ---------------- the checked pass ----------------
    FAST, SFX = False, ""
    tile_loop(g)
    epilogue(g, False)
    e("s_waitcnt vmcnt(0)")
    return g


def main():
    global ABL, MASKED
    if len(sys.argv) > 2 and sys.argv[1] == "--abl":       # tools/: timing ablations for the diagnostic library (not committed)
        for a in sys.argv[2].split(","):
            ABL = int(a)
            g = generate()
            body = g.lines + ["s_branch END_%="] + g.tail + ["END_%=:"]
            path = OUT.replace("_asm.inc", "_asm_abl%d.inc" % ABL)
            with open(path, "w") as f:
                f.write("// GENERATED by tools/gen_attn_w64.py --abl: timing ablation %d, WRONG results by construction.\n" % ABL)
                f.write('R"ASM(\n' + "\n".join(body) + '\n)ASM"\n')
            print("wrote", path)
        return
    ABL = 0
    for MASKED, path in ((False, OUT), (True, OUT.replace("_asm.inc", "_asm_mask.inc"))):
        g = generate()
        body = g.lines + ["s_branch END_%="] + g.tail + ["END_%=:"]
        with open(path, "w") as f:
            f.write("// GENERATED by tools/gen_attn_w64.py - do not edit.  %d instructions / labels.  Variant: %s.\n"
                    % (len(body), "key mask (per-tile flags)" if MASKED else "no key mask (only the last tile can be ragged)"))
            f.write('R"ASM(\n')
            for l in body:
                f.write(l + "\n")
            f.write(')ASM"\n')
        print("wrote", path, len(body), "lines")
    clob = ["memory", "vcc", "scc"] + ["v%d" % i for i in range(24, 256)] + ["a%d" % i for i in range(256)] + ["s%d" % i for i in range(36, 100)]
    with open(OUT.replace("_asm.inc", "_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_w64.py - do not edit.\n")
        f.write(", ".join('"%s"' % c for c in clob) + "\n")


if __name__ == "__main__":
    main()
