"""Scoring a corpus of ragged videos on 1..N GPUs (BASELINE configs[3]).

Videos are independent units (attention is within-video), so the corpus is dealt to ranks with no
exchange during compute; the only collective is the gather of the per-frame scores at the end
(`all_gather`, RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).

The batching mirrors what the reference's loader feeds the scorer (right-padding with 1000.0 and a
key mask, `data/dataset.py:157-161`, `train.py:118`) but buckets videos by length so the padding
waste stays small.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch

from .synth import PAD_VALUE


def video_cost(T: int, d_model: int = 256, in_features: int = 1024, num_layers: int = 4) -> float:
    """Algorithmic FLOPs of scoring one T-frame video: T·(2·Din·d + L·(24d² + 4Td))."""
    return float(T) * (2.0 * in_features * d_model + num_layers * (24.0 * d_model ** 2 + 4.0 * T * d_model))


def plan_shards(lengths: Sequence[int], world: int, **cost_kw) -> List[List[int]]:
    """Deterministic longest-processing-time assignment of video indices to `world` ranks.
    Every rank computes the same plan from the same lengths — no communication."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += video_cost(lengths[i], **cost_kw)
    return shards


def bucket_batches(indices: Sequence[int], lengths: Sequence[int], max_frames: int = 65536,
                   max_waste: float = 0.25, packed: bool = False) -> List[List[int]]:
    """Groups a shard's videos (sorted by length) into padded batches of at most `max_frames`
    padded frames whose padding waste stays under `max_waste`.  `packed`: the batches will be scored PACKED (frames
    concatenated, nothing padded), so only the frame budget counts - a 75-video shard of 30 k frames is ONE batch
    (one set of launches) instead of a dozen."""
    order = sorted(indices, key=lambda i: (lengths[i], i))
    batches: List[List[int]] = []
    cur: List[int] = []
    if packed:
        total = 0
        for i in order:
            if cur and total + lengths[i] > max_frames:
                batches.append(cur)
                cur, total = [], 0
            cur.append(i)
            total += lengths[i]
        if cur:
            batches.append(cur)
        return batches
    for i in order:
        if cur:
            tmax = lengths[i]                       # sorted ascending: newest is the longest
            total = sum(lengths[j] for j in cur) + lengths[i]
            padded = tmax * (len(cur) + 1)
            if padded > max_frames or 1.0 - total / padded > max_waste:
                batches.append(cur)
                cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


def pad_batch(videos: Sequence[torch.Tensor], device=None):
    """[T_i, D] tensors -> (x [B, Tmax, D] right-padded with 1000.0, mask [B, Tmax] bool or None).
    A handful of launches whatever the number of videos: one pad_sequence, one comparison for the mask."""
    lengths = [int(v.shape[0]) for v in videos]
    tmax = max(lengths)
    vs = [v.to(device=device, dtype=torch.float32) for v in videos]
    x = torch.nn.utils.rnn.pad_sequence(vs, batch_first=True, padding_value=PAD_VALUE)
    mask = None
    if any(t != tmax for t in lengths):
        lt = torch.tensor(lengths, device=x.device)
        mask = torch.arange(tmax, device=x.device)[None, :] >= lt[:, None]
    return x, mask


ScoreFn = Callable[[torch.Tensor, Optional[torch.Tensor]], torch.Tensor]   # (x, mask) -> scores [B, T]


PackedScoreFn = Callable[[torch.Tensor, List[int]], torch.Tensor]           # (x [sum T, D], lengths) -> scores [sum T]


def score_corpus(score_fn: ScoreFn, videos: Sequence[torch.Tensor], rank: int = 0, world: int = 1,
                 group=None, device=None, max_frames: int = 65536,
                 packed_fn: Optional[PackedScoreFn] = None, force_collective: bool = False) -> Dict[int, torch.Tensor]:
    """Scores every video once across `world` ranks and returns {video index: scores [T_i]} on EVERY
    rank (CPU tensors).  `score_fn` is `SimNet.score` on a GPU box.  With world == 1 no
    `torch.distributed` call is made (unless `force_collective`: the gather then runs over a one-rank group - how the
    RCCL branch is exercised on a one-GPU box).  With `packed_fn` (`SimNet.score_packed`) the batches are PACKED - the
    videos' frames concatenated, no sentinel padding, no mask - instead of padded; the scores are the same bits.

    The scores stay on the scorer's device until the end: the batches' outputs are concatenated, ONE indexed copy
    moves every valid frame into its row of the send buffer [n_max, t_max] (the index is built on the host from the
    lengths: two small uploads per call, none per video), ONE all_gather_into_tensor exchanges the shards and ONE
    device-to-host copy brings the result back.  No video ids travel: every rank derives the same plan."""
    lengths = [int(v.shape[0]) for v in videos]
    shards = plan_shards(lengths, world)
    mine = shards[rank]
    n_max = max((len(s) for s in shards), default=0)
    t_max = max(lengths) if lengths else 0
    pending, src, dst = [], [], []
    base = 0                                            # offset of the current batch in the concatenated outputs
    slot_of = {i: k for k, i in enumerate(mine)}
    for batch in bucket_batches(mine, lengths, max_frames, packed=packed_fn is not None):
        if packed_fn is not None:
            x = torch.cat([videos[i].to(device=device, dtype=torch.float32) for i in batch], dim=0)
            out = packed_fn(x, [lengths[i] for i in batch]).detach().float().reshape(-1)
            row = 0
            for i in batch:
                src.append(torch.arange(base + row, base + row + lengths[i]))
                row += lengths[i]
        else:
            x, mask = pad_batch([videos[i] for i in batch], device)
            out2 = score_fn(x, mask).detach().float()    # stays on the device: no sync per batch
            tb = out2.shape[1]
            out = out2.reshape(-1)
            for b, i in enumerate(batch):
                src.append(torch.arange(base + b * tb, base + b * tb + lengths[i]))
        for i in batch:
            dst.append(torch.arange(slot_of[i] * t_max, slot_of[i] * t_max + lengths[i]))
        pending.append(out)
        base += out.numel()
    collective = world > 1 or force_collective
    if collective:
        import torch.distributed as dist
        on_dev = device is not None and dist.get_backend(group) == "nccl"
    out_dev = pending[0].device if pending else (torch.device(device) if device is not None else torch.device("cpu"))
    send = torch.zeros((n_max, t_max), dtype=torch.float32, device=out_dev)
    if pending:
        flat = pending[0] if len(pending) == 1 else torch.cat(pending)
        src_i, dst_i = torch.cat(src).to(out_dev), torch.cat(dst).to(out_dev)
        send.view(-1).index_copy_(0, dst_i, flat.index_select(0, src_i))
    if not collective:
        host = send.cpu()                               # the one device-to-host copy
        return {i: host[slot_of[i], : lengths[i]].clone() for i in mine}
    if not on_dev:
        send = send.cpu()
    recv = torch.empty((world * n_max, t_max), dtype=torch.float32, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)      # concatenated along dim 0 (gloo and nccl)
    recv = recv.view(world, n_max, t_max).cpu()
    out: Dict[int, torch.Tensor] = {}
    for r in range(world):
        for slot, i in enumerate(shards[r]):
            out[i] = recv[r, slot, : lengths[i]].clone()
    return out


def score_host_batches(score_fn: ScoreFn, batches, device) -> List[torch.Tensor]:
    """Scores a sequence of HOST batches - padded (x [B,T,D] pinned or pageable, mask or None) or packed
    (x [sum T, D], list of lengths; `score_fn` is then `SimNet.score_packed`) - with the host-to-device
    copy of batch i+1 overlapped with the kernels of batch i: copies run on a side HIP stream into two device
    buffers that alternate, scores come back with an asynchronous D2H copy.  Returns the per-batch score tensors
    (host, valid after the final synchronize this function performs).  This is the PCIe-inclusive use of the
    scorer (features arrive from the loader's pinned host buffers, `train.py:140`): its steady-state rate is
    max(copy time, compute time) per batch instead of their sum."""
    compute = torch.cuda.current_stream(device)
    copy = torch.cuda.Stream(device=device)
    staged: List[Optional[tuple]] = [None, None]          # device (x, mask, ready-event) per slot
    freed = [None, None]                                  # event: the slot's previous batch was consumed
    outs: List[torch.Tensor] = []
    batches = list(batches)

    def upload(i: int) -> None:
        slot = i & 1
        x, mask = batches[i]
        with torch.cuda.stream(copy):
            if freed[slot] is not None:
                copy.wait_event(freed[slot])              # do not overwrite a buffer the kernels still read
            dx = x.to(device, non_blocking=True)
            # (x [B,T,D], mask or None) for padded batches, or (x [sum T, D], lengths list) for packed ones
            dm = mask.to(device, non_blocking=True) if isinstance(mask, torch.Tensor) else mask
            ev = torch.cuda.Event()
            ev.record(copy)
        staged[slot] = (dx, dm, ev)

    if batches:
        upload(0)
    for i in range(len(batches)):
        if i + 1 < len(batches):
            upload(i + 1)
        dx, dm, ev = staged[i & 1]
        compute.wait_event(ev)
        dx.record_stream(compute)
        if isinstance(dm, torch.Tensor):
            dm.record_stream(compute)
        sc = score_fn(dx, dm)
        done = torch.cuda.Event()
        done.record(compute)
        freed[i & 1] = done
        host = torch.empty(sc.shape, dtype=sc.dtype, pin_memory=True)
        host.copy_(sc, non_blocking=True)
        outs.append(host)
    torch.cuda.synchronize(device)
    return outs
