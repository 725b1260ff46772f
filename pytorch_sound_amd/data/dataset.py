"""Drop-in for pytorch_sound/data/dataset.py - the feeding side of ``Trainer.train``: ``SpeechDataset``,
``BucketRandomBatchSampler``, ``SpeechDataLoader`` (same names, arguments, RNG consumption and collate results), plus
the MI355X side of it:

* ``BucketRandomBatchSampler(rank=, world_size=)``: one process per GPU - every rank walks the SAME bucket sequence
  (step times match: a batch's clips have similar lengths) and takes a disjoint strided slice of each global batch;
* ``ragged_collate_fn`` + ``DevicePrefetcher``: variable-length audio is shipped back to back in ONE pinned buffer
  (the zero padding never crosses PCIe), copied on a side stream while the previous step computes, and laid out as the
  padded (N, Tmax) batch + validity mask by ``psnd_pad_collate`` in HBM.
"""
import math
from typing import Any, Callable, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Sampler
from torch.utils.data.dataloader import default_collate

from pytorch_sound_amd.data.meta import MetaFrame, MetaType


def _read_wav(path: str) -> Tuple[np.ndarray, int]:
    """float32 mono in [-1, 1) at the file's own rate - what ``librosa.load(path, sr=None)`` returns (dataset.py:105)."""
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    if x.dtype == np.uint8:
        x = (x.astype(np.float32) - 128.0) / 128.0
    elif x.dtype.kind == 'i':
        x = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1))
    else:
        x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    return x, int(sr)


class SpeechDataset(Dataset):
    """MetaFrame-driven data set (dataset.py:14-127).

    :param meta_frame: a ``MetaFrame``
    :param fix_len: crop every audio column to this many samples (0 = whole clips)
    :param fix_shuffle: draw a new crop start for every audio column instead of sharing the first one
    :param skip_audio: drop the audio columns from ``cols``
    :param audio_mask: append a ones-mask shaped like the (cropped) audio
    :param extra_features: [(column name, function)] - function(item) is appended after the columns
    """

    def __init__(self, meta_frame: MetaFrame, fix_len: int = 0, fix_shuffle: bool = False, skip_audio: bool = False,
                 audio_mask: bool = False, extra_features: List[Tuple[str, Callable]] = None):
        self.meta_frame = meta_frame
        self.fix_len = fix_len
        self.fix_shuffle = fix_shuffle
        self.cols = self.meta_frame.process_columns
        self.audio_mask = audio_mask
        self.extra_features = extra_features
        if self.extra_features:
            known = [name for _, name in self.meta_frame.columns]
            assert all([name in known for name, _ in extra_features]), \
                'Unmatched extra_feature name! {} {}'.format(str(known), str(extra_features))
            self.target_idx_map = {name: i for i, (_, name) in enumerate(self.meta_frame.process_columns)}
        if skip_audio:
            self.cols = [c for c in self.cols if c[0] != MetaType.AUDIO]

    def __getitem__(self, idx: int) -> List:
        return self.handle_fields(self.meta_frame.iloc[idx])

    def __len__(self) -> int:
        return len(self.meta_frame)

    def handle_fields(self, meta_item) -> List:
        """one row of the frame -> [column items ..., extra features ..., mask]"""
        out, mask, start = [], None, -1
        for kind, name in self.meta_frame.process_columns:
            if kind == MetaType.AUDIO:
                item = self.load_audio(meta_item[name])
                if self.fix_len:                                             # random crop (one start per row unless fix_shuffle)
                    if start == -1 or self.fix_shuffle:
                        start = np.random.randint(0, max(1, len(item) - self.fix_len + 1))
                    item = item[start:start + self.fix_len]
                if self.audio_mask and mask is None:
                    mask = np.ones_like(item)
            elif kind == MetaType.SCALAR:
                item = int(meta_item[name])
            elif kind == MetaType.MIDI:
                item = self.load_midi(meta_item[name])
            elif kind == MetaType.TEXT:
                item = self.load_txt(meta_item[name])
            else:
                raise NotImplementedError('{} is not implemented !'.format(name))
            out.append(item)
        for name, func in (self.extra_features or []):
            out.append(func(out[self.target_idx_map[name]]))
        if mask is not None:
            out.append(mask)
        return out

    def load_audio(self, file_path: str) -> np.ndarray:
        if file_path.endswith('.wav'):
            wav, sr = _read_wav(file_path)
            assert sr == self.meta_frame.sr, 'sample rate miss match.\n {}\t {} in {}'.format(self.meta_frame.sr, sr, file_path)
        elif file_path.endswith('.npy'):
            wav = np.load(file_path)
        else:
            raise NotImplementedError('{} : File Type is not implemented to load audio data !'.format(file_path))
        return wav

    @staticmethod
    def load_midi(file_path: str):
        raise NotImplementedError('MIDI columns need the reference\'s pretty_midi front end (utils/sound.py:parse_midi), which is '
                                  'outside the accelerated path (SURVEY.md section 8); override load_midi')

    @staticmethod
    def load_txt(txt: str):
        raise NotImplementedError('TEXT columns need the reference\'s text front end (utils/text.py:eng_t2i), which is outside the '
                                  'accelerated path (SURVEY.md section 8); override load_txt')


class BucketRandomBatchSampler(Sampler):
    """Batches drawn from contiguous index buckets (dataset.py:128-167): the frame is sorted by length, so a bucket holds
    clips of similar length.  Bucket order and in-bucket order come from the global ``np.random`` state exactly as in the
    reference (one shuffle per bucket, one ``choice`` per batch), so a seeded run yields the reference's batches.

    ``rank`` / ``world_size`` (not in the reference): ``batch_size`` is then the per-rank size; buckets are cut into
    global batches of ``batch_size * world_size`` and rank r takes ``ids[r::world_size]``.  Every rank must hold the same
    ``np.random`` state when iteration starts (seed it identically per epoch)."""

    def __init__(self, data_source: Dataset, n_buckets: int, batch_size: int, skip_last_bucket: bool = False,
                 rank: Optional[int] = None, world_size: Optional[int] = None):
        self.world_size = int(world_size or 1)
        self.rank = int(rank or 0)
        if not 0 <= self.rank < self.world_size:
            raise ValueError('rank {} outside world of {}'.format(self.rank, self.world_size))
        self.local_batch = batch_size
        batch_size = batch_size * self.world_size
        assert len(data_source) > n_buckets * batch_size, 'Data size is too small to use bucket sampler !'
        self.n_buckets = n_buckets
        self.data_size = len(data_source)
        self.batch_size = batch_size
        self.bucket_size = int(math.ceil(self.data_size / self.n_buckets))
        self.bucket_size -= self.bucket_size % batch_size
        if self.n_buckets <= 0:
            raise ValueError("the num of buckets has to be a positive value.")
        self.skip_last_bucket = skip_last_bucket

    @property
    def buckets(self) -> List[List[int]]:
        n = self.n_buckets - int(self.skip_last_bucket)
        return [list(range(b * self.bucket_size, (b + 1) * self.bucket_size)) for b in range(n)]

    def __iter__(self):
        pools = self.buckets
        for pool in pools:
            np.random.shuffle(pool)
        while pools:
            b = np.random.choice(range(len(pools)))
            ids, pools[b] = pools[b][-self.batch_size:], pools[b][:-self.batch_size]     # the tail of the shuffled bucket
            if not pools[b]:
                pools.pop(b)
            yield ids[self.rank::self.world_size] if self.world_size > 1 else ids

    def __len__(self) -> int:
        return self.bucket_size * self.n_buckets // self.batch_size


def _pad_zero(arrays: List[np.ndarray]) -> np.ndarray:
    """zero-pad 1-3 dimensional arrays to their common maximum shape (dataset.py:228-248)"""
    nd = arrays[0].ndim
    if not 1 <= nd <= 3:
        raise ValueError
    top = list(arrays[0].shape)
    for a in arrays[1:]:
        for d, size in enumerate(a.shape):
            top[d] = max(top[d], size)
    out = np.zeros((len(arrays), *top), dtype=arrays[0].dtype)
    for i, a in enumerate(arrays):
        out[(i,) + tuple(slice(0, s) for s in a.shape)] = a
    return out


def _collate_field(field: List[Any]):
    first = field[0]
    if not isinstance(first, np.ndarray):
        assert all([type(x) == type(first) for x in field[1:]])
        return torch.LongTensor(field) if isinstance(first, int) else field       # floats / lists stay python lists (dataset.py:214)
    if any(a.shape != first.shape for a in field[1:]):
        return torch.from_numpy(_pad_zero(field))
    return torch.from_numpy(np.stack(field))


class SpeechDataLoader(DataLoader):
    """``DataLoader`` with the zero-padding collate and the optional bucket sampler (dataset.py:170-250).  ``rank`` /
    ``world_size`` are handed to the bucket sampler (see there)."""

    def __init__(self, dataset: SpeechDataset, batch_size: int, num_workers: int, n_buckets: int = 10, is_bucket: bool = False,
                 is_shuffle: bool = False, skip_last_bucket: bool = False, pin_memory: bool = True, drop_last: bool = False,
                 rank: Optional[int] = None, world_size: Optional[int] = None, collate_fn: Optional[Callable] = None):
        batch_sampler = None
        if is_bucket:
            batch_sampler = BucketRandomBatchSampler(dataset, n_buckets=n_buckets, batch_size=batch_size,
                                                     skip_last_bucket=skip_last_bucket, rank=rank, world_size=world_size)
        super().__init__(dataset, num_workers=num_workers, collate_fn=collate_fn or self.pad_collate_fn, pin_memory=pin_memory,
                         batch_size=(1 if is_bucket else batch_size), shuffle=(not is_bucket and is_shuffle),
                         batch_sampler=batch_sampler, drop_last=drop_last)

    @staticmethod
    def pad_collate_fn(batch: List[Any]):
        """items of a mini batch -> one entry per field: ints -> LongTensor, arrays of one shape -> stacked, arrays of
        different shapes -> zero-padded to the largest; a batch of one goes through ``default_collate``."""
        if len(batch) > 1:
            return [_collate_field([item[i] for item in batch]) for i in range(len(batch[0]))]
        if None in batch:
            return None
        return default_collate(batch)


# ---------------------------------------------------------------------------------------------------------------
# MI355X side: ragged host batches, padded in HBM
# ---------------------------------------------------------------------------------------------------------------
class RaggedBatch:
    """N variable-length fp32 clips back to back: ``flat`` (sum of lengths, pinned when possible), ``lens`` (N int64)."""

    def __init__(self, clips: List[np.ndarray], pin: Optional[bool] = None):
        lens = np.asarray([len(c) for c in clips], dtype=np.int64)
        if pin is None:                 # loader workers must not touch the GPU runtime: there the DataLoader's pin thread
            pin = torch.cuda.is_available() and torch.utils.data.get_worker_info() is None      # calls pin_memory() below
        # straight into page-locked memory (the caching host allocator hands the same blocks back batch after batch)
        flat = torch.empty(int(lens.sum()), dtype=torch.float32, pin_memory=bool(pin))
        np.concatenate([np.asarray(c, dtype=np.float32) for c in clips], out=flat.numpy())
        self.flat = flat
        self.lens = torch.from_numpy(lens)
        self.offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64))

    def __len__(self):
        return len(self.lens)

    def pin_memory(self):
        """DataLoader(pin_memory=True) protocol for custom batch types"""
        if not self.flat.is_pinned():
            self.flat = self.flat.pin_memory()
        return self

    def padded(self) -> torch.Tensor:
        """host restatement of the padded layout (tests; no device)"""
        out = torch.zeros(len(self), int(self.lens.max()))
        for n, (o, l) in enumerate(zip(self.offs.tolist(), self.lens.tolist())):
            out[n, :l] = self.flat[o:o + l]
        return out

    def to_device(self, device, want_mask: bool = False, multiple: int = 1):
        """(N, Tmax) zero-padded batch (+ mask) in HBM via psnd_pad_collate, enqueued on the current stream of `device`.
        ``multiple`` rounds Tmax up (e.g. to the hop size) so downstream shapes repeat and hipGraphs are reused."""
        from pytorch_sound_amd._lib import lib, check, ptr, stream_ptr
        N = len(self)
        Tmax = int(self.lens.max())
        Tmax = (Tmax + multiple - 1) // multiple * multiple
        flat = self.flat.to(device, non_blocking=True)
        meta = torch.stack([self.offs, self.lens]).to(device, non_blocking=True)
        out = torch.empty((N, Tmax), dtype=torch.float32, device=device)
        mask = torch.empty((N, Tmax), dtype=torch.float32, device=device) if want_mask else None
        with torch.cuda.device(device):
            check(lib().psnd_pad_collate(ptr(flat), ptr(meta[0]), ptr(meta[1]), N, Tmax, ptr(out), ptr(mask), stream_ptr(device)),
                  'psnd_pad_collate')
        return (out, mask) if want_mask else out


def ragged_collate_fn(batch: List[Any]):
    """``pad_collate_fn`` with every field of variable-length 1-D float arrays kept ragged (``RaggedBatch``) instead of
    padded on the host; all other fields exactly as ``pad_collate_fn`` (batches of one included)."""
    if len(batch) <= 1:
        return SpeechDataLoader.pad_collate_fn(batch)
    out = []
    for i in range(len(batch[0])):
        field = [item[i] for item in batch]
        if isinstance(field[0], np.ndarray) and field[0].ndim == 1 and field[0].dtype.kind == 'f':
            out.append(RaggedBatch(field))
        else:
            out.append(_collate_field(field))
    return out


class DevicePrefetcher:
    """Iterate a loader one batch ahead: while step k computes, batch k+1 is copied on a side stream (ragged audio as one
    pinned buffer) and padded in HBM (``psnd_pad_collate``); the consumer's stream waits on an event, never on the host.
    Yields lists with tensors on ``device`` (RaggedBatch fields become the padded (N, Tmax) tensor; with ``want_mask``
    the mask is appended at the end, like ``audio_mask`` does on the host)."""

    def __init__(self, loader, device, want_mask: bool = False, multiple: int = 1):
        self.loader, self.device = loader, torch.device(device)
        self.want_mask, self.multiple = want_mask, multiple
        self.stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        if batch is None:
            return None, None
        with torch.cuda.stream(self.stream):
            out, mask = [], None
            for x in batch:
                if isinstance(x, RaggedBatch):
                    if self.want_mask and mask is None:
                        x, mask = x.to_device(self.device, True, self.multiple)
                    else:
                        x = x.to_device(self.device, False, self.multiple)
                elif isinstance(x, torch.Tensor):
                    x = x.to(self.device, non_blocking=True)
                out.append(x)
            if mask is not None:
                out.append(mask)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._stage(next(it, None))
        while nxt[0] is not None:
            cur, ev = nxt
            nxt = self._stage(next(it, None))
            cur_stream = torch.cuda.current_stream(self.device)
            cur_stream.wait_event(ev)
            for x in cur:
                if isinstance(x, torch.Tensor):
                    x.record_stream(cur_stream)
            yield cur
