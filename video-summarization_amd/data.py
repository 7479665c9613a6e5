"""Input pipeline — the file half (reference ``src/data/dataset.py:64-168``, ``src/data/path.py``).

Two layers:

1. **Drop-in datasets** with the reference's surface: ``TSDataset(root, ex_dataset, datasets, key, split)``,
   ``PreTrainDataset(root)``, ``UserSummaries``, ``collate_fn_train / _test / _pretrain`` and ``PATH`` — same
   constructor arguments, same ``__getitem__`` tuples, same sentinel-1000 padding, so ``train.py:49-70`` builds its
   loaders unchanged.  The on-disk layout is the reference's: one container per dataset (``PATH[name]``) whose
   groups ``video_N`` hold ``features [T,1024]``, ``gtscore [T]``, ``user_summary``, ``user_scores``,
   ``change_points``, ``n_frames``, ``picks``.  ``open_container`` reads it from HDF5 when ``h5py`` is importable
   (guarded import: it is not in this image) and from an ``.npz`` archive with the same ``video_N/field`` keys
   otherwise (``tools/h5_to_npz.py`` converts; ``write_npz_container`` writes) — every line above the opener is the same
   code either way.

2. **The MI355X-side feed**: ``RaggedFeeder`` turns a dataset into length-bucketed PACKED host batches (the
   videos' frames concatenated — no sentinel rows, no mask; ``corpus.bucket_batches``) written by a producer
   thread into a ring of pinned host buffers, and ``score_dataset`` streams them to the device with the copy of
   batch i+1 overlapped with the kernels of batch i (side HIP stream, two device buffers) into
   ``SimNet.score_packed``.  ``val_step_from_dataset`` is ``val_step`` (``train.py:134-152``) from files to
   ``(loss, f_score, kendall, spearman)`` with that feed; a video's scores are the same bits as scoring it alone.
"""
from __future__ import annotations

import glob
import os
import queue
import threading
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset

from .corpus import bucket_batches
from .synth import IN_FEATURES, PAD_VALUE

# reference src/data/path.py:1-6
PATH = {
    'ovp': 'eccv16_dataset_ovp_google_pool5.h5',
    'summe': 'summarizer_dataset_summe_google_pool5.h5',
    'tvsum': 'summarizer_dataset_tvsum_google_pool5.h5',
    'youtube': 'eccv16_dataset_youtube_google_pool5.h5',
}
FIELDS = ("features", "gtscore", "user_summary", "user_scores", "change_points", "n_frames", "picks")


# --------------------------------------------------------------------------------------------
# containers: HDF5 (h5py, if present) or .npz with the same keys
# --------------------------------------------------------------------------------------------
class _NpzGroup:
    def __init__(self, z, prefix):
        self._z, self._p = z, prefix

    def __getitem__(self, field):
        return self._z[self._p + "/" + field]

    def __contains__(self, field):
        return (self._p + "/" + field) in self._z.files


class NpzContainer:
    """``f.keys()``, ``f[key][field][...]`` and the context-manager protocol of ``h5py.File`` over an ``.npz``
    archive whose member names are ``video_N/field``."""

    def __init__(self, path):
        self._z = np.load(path, allow_pickle=False)
        names: Dict[str, None] = {}
        for member in self._z.files:
            names.setdefault(member.split("/", 1)[0], None)
        self._keys = list(names)

    def keys(self):
        return list(self._keys)

    def __getitem__(self, key):
        if key not in self._keys:
            raise KeyError(key)
        return _NpzGroup(self._z, key)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._z.close()
        return False


def open_container(path: str):
    """Opens one dataset container for reading.  ``<path>`` as given if it exists and h5py is importable; otherwise
    ``<path minus .h5>.npz``.  Raises with both reasons when neither works — never a silent empty dataset."""
    npz = path[:-3] + ".npz" if path.endswith(".h5") else path + ".npz"
    if path.endswith(".npz"):
        return NpzContainer(path)
    h5_error = None
    if os.path.exists(path):
        try:
            import h5py                                   # guarded: not installed in every image
            return h5py.File(path, "r")
        except ImportError as e:
            h5_error = e
    if os.path.exists(npz):
        return NpzContainer(npz)
    if h5_error is not None:
        raise ImportError("%s exists but h5py is not installed (%s) and there is no %s beside it; convert it with "
                          "tools/h5_to_npz.py where h5py is available" % (path, h5_error, npz))
    raise FileNotFoundError("neither %s nor %s exists" % (path, npz))


def write_npz_container(path: str, videos: Dict[str, Dict[str, np.ndarray]]) -> str:
    """Writes ``{video_N: {field: array}}`` in the ``video_N/field`` layout ``NpzContainer`` reads."""
    flat = {"%s/%s" % (k, f): np.asarray(v) for k, rec in videos.items() for f, v in rec.items()}
    np.savez(path, **flat)
    return path if path.endswith(".npz") else path + ".npz"


# --------------------------------------------------------------------------------------------
# the reference's dataset surface
# --------------------------------------------------------------------------------------------
class UserSummaries:                                         # dataset.py:146-154
    def __init__(self, user_summary, user_scores, name, changes_point, n_frames, picks):
        self.user_summary = user_summary
        self.user_scores = user_scores
        self.change_points = changes_point
        self.n_frames = n_frames
        self.picks = picks
        self.name = name


class TSDataset(Dataset):
    """dataset.py:64-136.  ``split="val"``: the videos of ``ex_dataset`` (the split's ``key`` list, or all) with
    their ``UserSummaries``; otherwise every dataset of ``datasets`` ("a+b"), ``key`` restricting ``ex_dataset``
    only, videos of <= 50 frames dropped (:117)."""

    def __init__(self, root, ex_dataset, datasets, key=None, split: str = "train"):
        self.root, self.key, self.split, self.ex_dataset = root, key, split, ex_dataset
        self.datasets = datasets.split("+")
        self.data, self.target, self.user_summaries = [], [], []
        if split == "val":
            with open_container(os.path.join(root, PATH[ex_dataset])) as f:
                names = self.get_datasets(self.key) if key else f.keys()
                for name in names:
                    g = f[name]
                    self.data.append(g['features'][...].astype(np.float32))
                    self.target.append(g['gtscore'][...].astype(np.float32))
                    self.user_summaries.append(UserSummaries(
                        np.array(g['user_summary']), np.array(g['user_scores']), name, np.array(g['change_points']),
                        np.array(g['n_frames']), np.array(g['picks'])))
        else:
            for dataset in self.datasets:
                with open_container(os.path.join(root, PATH[dataset])) as f:
                    names = self.get_datasets(self.key) if (key and dataset == ex_dataset) else f.keys()
                    for name in names:
                        features = f[name]['features'][...].astype(np.float32)
                        target = f[name]['gtscore'][...].astype(np.float32)
                        if features.shape[0] > 50:
                            self.data.append(features)
                            self.target.append(target)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        features, targets = torch.tensor(self.data[idx]), torch.tensor(self.target[idx])
        if self.split == "train":
            return features, targets
        return features, targets, self.user_summaries[idx]

    def get_datasets(self, keys: List[str]):                 # dataset.py:138-141: the basename of each split key
        return [str(Path(k).name) for k in keys]


class PreTrainDataset(Dataset):
    """dataset.py:39-60: ``root/frames/<video>.npy`` ([T,1024] features) with ``root/video/<video>.npy`` (the
    512-d video representation)."""

    def __init__(self, root):
        self.root, self.data = root, []
        for frame_path in glob.glob(os.path.join(root, "frames") + "/*"):
            name = os.path.basename(frame_path).split(".")[0]
            self.data.append((np.load(frame_path), np.load("%s/%s.npy" % (os.path.join(root, "video"), name))))

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        feature, vid_rep = self.data[idx]
        return torch.tensor(feature), torch.tensor(vid_rep)


def collate_fn_train(batch):                                 # dataset.py:157-161
    features, targets = zip(*batch)
    return (pad_sequence(features, batch_first=True, padding_value=PAD_VALUE),
            pad_sequence(targets, batch_first=True, padding_value=PAD_VALUE))


def collate_fn_test(batch):                                  # dataset.py:164-168
    features, targets, user_summaries = batch[0]
    return features.unsqueeze(0), targets.unsqueeze(0), user_summaries


def collate_fn_pretrain(batch):                              # dataset.py:139-143
    features, vid_reps = zip(*batch)
    return pad_sequence(features, batch_first=True, padding_value=PAD_VALUE), torch.stack(vid_reps, dim=0)


# --------------------------------------------------------------------------------------------
# the MI355X-side feed: packed ragged batches through a pinned ring
# --------------------------------------------------------------------------------------------
class _PinnedRing:
    """`slots` host buffers of [max_frames, D] floats, page-locked when a HIP device is present.  A slot handed back
    with the event of its host-to-device copy is only reused after that copy has finished."""

    def __init__(self, slots: int, max_frames: int, in_features: int):
        pin = torch.cuda.is_available()
        self.buffers = [torch.empty((max_frames, in_features), dtype=torch.float32, pin_memory=pin) for _ in range(slots)]
        self._free: "queue.Queue[Tuple[int, Optional[torch.cuda.Event]]]" = queue.Queue()
        for i in range(slots):
            self._free.put((i, None))

    def acquire(self, stop: Optional[threading.Event] = None) -> Optional[int]:
        """A free slot, or None once ``stop`` is set (checked every 50 ms while waiting)."""
        while True:
            try:
                slot, ev = self._free.get(timeout=0.05)
            except queue.Empty:
                if stop is not None and stop.is_set():
                    return None
                continue
            if ev is not None:
                ev.synchronize()
            return slot

    def release(self, slot: int, copied: Optional["torch.cuda.Event"]) -> None:
        self._free.put((slot, copied))


class RaggedFeeder:
    """Iterates ``videos`` (a sequence of [T_i, D] arrays / tensors, e.g. ``TSDataset.data``) as PACKED host batches
    ``(slot, x [sum T, D] view of a ring buffer, lengths, indices)`` in length-bucketed order, filled by a producer
    thread while the consumer computes.  ``done(slot, event)`` returns a buffer to the ring.  ``close()`` (also on
    ``with`` exit, and when iteration ends) stops the producer and joins it, whether or not every batch was consumed:
    a consumer that raises or breaks out of its loop must not leave a thread blocked on the ring with the pinned
    buffers alive."""

    def __init__(self, videos: Sequence, max_frames: int = 65536, max_waste: float = 0.25, slots: int = 3,
                 indices: Optional[Sequence[int]] = None):
        self.videos = videos
        self.lengths = [int(v.shape[0]) for v in videos]
        idx = list(range(len(videos))) if indices is None else list(indices)
        self.batches = bucket_batches(idx, self.lengths, max_frames, max_waste)
        d = int(videos[0].shape[1]) if len(videos) else IN_FEATURES
        cap = max([sum(self.lengths[i] for i in b) for b in self.batches] + [1])
        self.ring = _PinnedRing(slots, cap, d)
        self._q: "queue.Queue" = queue.Queue(maxsize=slots)
        self._err: List[BaseException] = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()

    def _put(self, item) -> bool:
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def _produce(self) -> None:
        try:
            for batch in self.batches:
                slot = self.ring.acquire(self._stop)
                if slot is None:
                    return
                buf, row = self.ring.buffers[slot], 0
                for i in batch:
                    v = self.videos[i]
                    t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
                    buf[row: row + self.lengths[i]].copy_(t)
                    row += self.lengths[i]
                if not self._put((slot, buf[:row], [self.lengths[i] for i in batch], list(batch))):
                    return
        except BaseException as e:              # surfaced to the consumer: a feeder must not die silently
            self._err.append(e)
        finally:
            self._put(None)

    def __iter__(self) -> Iterator[Tuple[int, torch.Tensor, List[int], List[int]]]:
        try:
            while True:
                # never block for good: after close() the producer's final sentinel may have been dropped (a closed feeder
                # iterated again, or close() from another thread while this one waits)
                try:
                    item = self._q.get(timeout=0.1)
                except queue.Empty:
                    if self._stop.is_set() or not self._thread.is_alive():
                        if self._err:
                            raise self._err[0]
                        return
                    continue
                if item is None:
                    if self._err:
                        raise self._err[0]
                    return
                yield item
        finally:                                # exhausted, the consumer raised, or the generator was dropped
            self.close()

    def close(self, timeout: float = 5.0) -> None:
        """Stops the producer thread and waits for it; idempotent."""
        self._stop.set()
        if self._thread.is_alive() and threading.current_thread() is not self._thread:
            self._thread.join(timeout)

    @property
    def alive(self) -> bool:
        return self._thread.is_alive()

    def __enter__(self) -> "RaggedFeeder":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def done(self, slot: int, copied=None) -> None:
        self.ring.release(slot, copied)


@torch.no_grad()
def score_dataset(model, videos: Sequence, device, max_frames: int = 65536, indices: Optional[Sequence[int]] = None
                  ) -> Dict[int, torch.Tensor]:
    """{video index: sigmoid scores [T_i] (CPU)} for ``videos`` — files' arrays to scores with no padding: the
    feeder's packed batches go host -> device on a side stream (batch i+1 under the kernels of batch i, two device
    buffers) into ``model.score_packed``.  HIP device only (the scorer has no CPU path)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("score_dataset runs on the MI355X HIP kernels only")
    feeder = RaggedFeeder(videos, max_frames=max_frames, indices=indices)
    compute = torch.cuda.current_stream(device)
    copy = torch.cuda.Stream(device=device)
    pending, out = [], {}
    staged = None                                          # (dx, lengths, indices, copied-event)

    def upload(item):
        slot, hx, lengths, idx = item
        with torch.cuda.stream(copy):
            dx = hx.to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy)
        feeder.done(slot, ev)
        return dx, lengths, idx, ev

    try:
        it = iter(feeder)
        first = next(it, None)
        if first is not None:
            staged = upload(first)
        while staged is not None:
            nxt = next(it, None)
            upcoming = upload(nxt) if nxt is not None else None
            dx, lengths, idx, ev = staged
            compute.wait_event(ev)
            dx.record_stream(compute)
            sc = model.score_packed(dx, lengths)
            host = torch.empty(sc.shape, dtype=sc.dtype, pin_memory=True)
            host.copy_(sc, non_blocking=True)
            pending.append((host, lengths, idx))
            staged = upcoming
    finally:
        feeder.close()                           # also when score_packed / the upload raised: no producer left behind
    torch.cuda.synchronize(device)
    for host, lengths, idx in pending:
        row = 0
        for i, t in zip(idx, lengths):
            out[i] = host[row: row + t].clone()
            row += t
    return out


@torch.no_grad()
def val_step_from_dataset(model, dataset: TSDataset, device, max_frames: int = 65536):
    """``val_step`` (train.py:134-152) over a ``split="val"`` ``TSDataset``: files -> pinned ring -> packed scoring ->
    keyshot evaluation.  Returns ``(mean MSE loss, f_score, kendall_tau, spearman_r)`` like the reference."""
    import torch.nn.functional as F
    from .evaluation import eval_metrics
    model.eval()
    if getattr(model, "_lib_dh", model.d_model // model.num_heads) not in (32, 64, 128):
        raise RuntimeError("packed scoring needs head dim 32 or 64; use harness.val_step for this model")
    scores = score_dataset(model, dataset.data, device, max_frames)
    score_dict, user_dict, loss = {}, {}, 0.0
    for i, user in enumerate(dataset.user_summaries):
        loss += F.mse_loss(scores[i].view(1, -1), torch.from_numpy(dataset.target[i]).view(1, -1)).item()   # train.py:145
        score_dict[user.name] = scores[i].numpy()
        user_dict[user.name] = user
    f_score, ktau, spr = eval_metrics(score_dict, user_dict)
    return loss / max(len(dataset), 1), f_score, ktau, spr
