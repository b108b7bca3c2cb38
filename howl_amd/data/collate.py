"""Device-side collate for device-resident clips: the reference's training collate chain
``compose(truncate_length, TimeshiftTransform().train(), NoiseTransform().train(), batchify)``
(``training/run/pretrain_gsc.py:78-80``; ``howl/data/transform/transform.py:120-196``, ``operator.py:73-86``) as one
``howl_collate_augment`` launch per batch -- optionally with ``DatasetMixer`` (``transform.py:199-231``) in front, as
``training/run/train.py:218`` composes it when a noise dataset is configured.

The per-batch gates and per-sample magnitudes are drawn on the host from the same ``random`` generator in the same order
as the reference modules draw them (one gate draw per augmentation parameter per batch; per example: shift amount then
head/tail for Timeshift, strength for each active Noise parameter); only the noise *samples* differ (counter-based
device generator instead of torch's CPU generator), so augmented batches match the reference in distribution.

Host cost (the reference spreads this work over ``cpu_count`` DataLoader worker processes, ``howl/data/dataloader.py:9-27``;
here ONE process feeds a ~1 ms device step, so the host side of a batch has to cost well under that):
* ``__call__`` makes the draws of a whole batch as array operations on numpy's MT19937 ``RandomState`` loaded with the state of
  the ``random`` generator the reference modules would use, and hands the advanced state back: the same stream, the same
  values, the same number of draws consumed (``tests/test_host.py`` holds the two forms to each other draw for draw);
* the per-batch decisions ride to the device in ONE copy out of a ring of pinned staging buffers;
* ``prefetch(id_batches)`` prepares batch k + 1 (draws, sort, packed buffer) in a worker thread while the caller launches
  batch k and its training step.
"""
import queue
import random
import threading

import numpy as np
import torch

from howl_amd import ops
from howl_amd.data.common.batch import ClassificationBatch

__all__ = ["DeviceCollate"]

TIMESHIFT_DOMAIN, TIMESHIFT_IDX, TIMESHIFT_PROB = [0.25, 0.5, 0.75, 1], 0, 0.75
WHITE_DOMAIN, WHITE_IDX = [0.0001, 0.00025, 0.0005, 0.001, 0.002], 3
SP_DOMAIN, SP_IDX = [1 / 20000, 1 / 15000, 1 / 10000, 1 / 5000, 1 / 2500], 2
NOISE_PROB = 0.75
MIXER_DOMAIN, MIXER_IDX, MIXER_PROB = [0.1, 0.2, 0.3, 0.4, 0.5], 1, 0.75


class DeviceCollate:
    def __init__(self, bank_audio, bank_lengths, bank_labels, max_len: int, sr: int = 16000, seed: int = None,
                 training: bool = True, background=None, do_replace: bool = False, replica: int = 0, row_offsets=None):
        """``background`` = (bg_audio (N, Lbg) on the device, bg_lengths): the noise dataset of ``DatasetMixer``.
        ``replica``: this process's rank in a data-parallel job -- it keys the device noise generator (and, with a ``seed``, the
        host draws), so that row r of every rank's shard does not receive the same shift / noise pattern.
        ``row_offsets``: ragged bank -- ``bank_audio`` is the (total, 1) view of a flat sample buffer and clip i starts at
        sample ``row_offsets[i]`` (``WakeWordClipBank``); default: a dense (N, Lmax) matrix, clip i = row i."""
        self.audio, self.lengths, self.labels = bank_audio, [int(v) for v in bank_lengths.tolist()], bank_labels
        self.last_max_len = 0      # longest row of the last batch (host knowledge: no read-back for the frame count)
        self.labels_host = None if bank_labels is None else [int(v) for v in bank_labels.tolist()]   # gathered on the host
        self.bg_audio = None if background is None else background[0]
        self.bg_lengths = None if background is None else [int(v) for v in background[1]]
        self.do_replace = do_replace
        self.row_offsets = None if row_offsets is None else [int(v) for v in row_offsets]
        self.max_len, self.sr, self.training = max_len, sr, training
        self._rand = random if seed is None else random.Random(seed + 1000003 * replica)    # replica 0: the stream of Random(seed)
        self._np_live = False      # True while numpy's copy of a PRIVATE stream is ahead of self._rand (see draw_arrays)
        self._calls = 0
        self._seed = 0 if seed is None else seed
        self._replica_key = (replica * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF      # 0 for a single process: same stream as before
        self._lengths_np = np.asarray(self.lengths, dtype=np.int64)
        self._labels_np = None if self.labels_host is None else np.asarray(self.labels_host, dtype=np.int64)
        self._rows_np = None if self.row_offsets is None else np.asarray(self.row_offsets, dtype=np.int64)
        self._np_rs = np.random.RandomState(0)
        self._ring, self._ring_next = [], 0

    @property
    def rand(self):
        """The ``random``-like generator the reference modules would draw from (the global module, or this collate's own
        ``random.Random``).  Reading it brings it up to date with the array draws made since (``draw_arrays``)."""
        if self._np_live:
            self._stream_from_numpy(self._np_rs)
            self._np_live = False
        return self._rand

    @rand.setter
    def rand(self, value):
        self._rand, self._np_live = value, False

    def draw(self, clip_ids):
        """Host-side parameter draws in the reference's order; returns per-sample lists (before the length sort)."""
        n = len(clip_ids)
        lens = [min(self.lengths[i], self.max_len) for i in clip_ids]            # truncate_length
        self.last_mix = self.draw_mixer(lens)                                     # DatasetMixer comes first (train.py:218)
        shift, head = [0] * n, [0] * n
        if self.rand.random() < TIMESHIFT_PROB and self.training:                 # AugmentModule.forward gate
            for k in range(n):
                w = min(int(self.rand.random() * TIMESHIFT_DOMAIN[TIMESHIFT_IDX] * self.sr), int(0.5 * lens[k]))
                shift[k] = w
                head[k] = 1 if self.rand.random() < 0.5 else 0
        sigma, sp = [0.0] * n, [0.0] * n
        if self.rand.random() < NOISE_PROB and self.training:                     # "white" parameter
            for k in range(n):
                sigma[k] = WHITE_DOMAIN[WHITE_IDX] * self.rand.random()
        if self.rand.random() < NOISE_PROB and self.training:                     # "salt_pepper" parameter
            for k in range(n):
                sp[k] = SP_DOMAIN[SP_IDX] * self.rand.random()
        return lens, shift, head, sigma, sp

    # ---- the same draws as array operations ------------------------------------------------------------------------
    def _stream_to_numpy(self):
        """numpy's RandomState IS the Mersenne Twister of ``random``: ``random_sample`` and ``random.random()`` build a double
        from the same two 32-bit outputs, so with the state copied over the two produce the same sequence."""
        if not self._np_live:
            st = self._rand.getstate()
            self._np_rs.set_state(("MT19937", np.asarray(st[1][:-1], dtype=np.uint32), st[1][-1]))
            self._gauss = st[2]
        return self._np_rs

    def _stream_from_numpy(self, rs):
        _, key, pos = rs.get_state()[:3]
        self._rand.setstate((3, tuple(key.tolist()) + (int(pos),), self._gauss))

    def draw_arrays(self, clip_ids):
        """``draw`` for a whole batch at once (no background mixer): numpy arrays lens, shift, head, sigma, sp -- element for
        element the values ``draw`` returns, with the generator left in the state ``draw`` leaves it in."""
        ids = np.asarray(clip_ids, dtype=np.int64)
        n = ids.size
        lens = np.minimum(self._lengths_np[ids], self.max_len)                    # truncate_length
        rs = self._stream_to_numpy()
        shift, head = np.zeros(n, np.int64), np.zeros(n, np.int64)
        if rs.random_sample() < TIMESHIFT_PROB and self.training:                 # AugmentModule.forward gate
            r = rs.random_sample(2 * n).reshape(n, 2)                             # per clip: shift draw, then head/tail draw
            shift = np.minimum((r[:, 0] * TIMESHIFT_DOMAIN[TIMESHIFT_IDX] * self.sr).astype(np.int64), (0.5 * lens).astype(np.int64))
            head = (r[:, 1] < 0.5).astype(np.int64)
        sigma, sp = np.zeros(n), np.zeros(n)
        if rs.random_sample() < NOISE_PROB and self.training:                     # "white" parameter
            sigma = WHITE_DOMAIN[WHITE_IDX] * rs.random_sample(n)
        if rs.random_sample() < NOISE_PROB and self.training:                     # "salt_pepper" parameter
            sp = SP_DOMAIN[SP_IDX] * rs.random_sample(n)
        # a private stream stays in numpy between consecutive array draws (the two state copies are 0.14 ms, most of a batch's
        # host time) and is handed back when anything reads ``self.rand``; the global ``random`` module has other users
        # (VTLP's alpha, transform.py:441), so it is brought up to date at once
        if self._rand is random:
            self._stream_from_numpy(rs)
        else:
            self._np_live = True
        self.last_mix = None
        return lens, shift, head, sigma, sp

    def draw_mixer(self, lens):
        """``DatasetMixer.forward`` draws (AugmentModule gate per parameter, then per example choice / randint / alpha)."""
        if self.bg_audio is None:
            return None
        n = len(lens)
        bg_id, bg_off, alpha = [0] * n, [0] * n, [0.0] * n
        for name, prob in (("strength", MIXER_PROB), ("replace", 0.1 if self.do_replace else 0.0)):
            if self.rand.random() < prob and self.training:
                for k in range(n):
                    j = self.rand.choice(range(len(self.bg_lengths)))
                    while self.bg_lengths[j] < lens[k]:
                        j = self.rand.choice(range(len(self.bg_lengths)))
                    b = self.rand.randint(lens[k], self.bg_lengths[j])
                    a = 1.0 if name == "replace" else self.rand.random() * MIXER_DOMAIN[MIXER_IDX]
                    bg_id[k], bg_off[k], alpha[k] = j, b - lens[k], a   # "replace" (alpha 1) overrides an earlier mix
        return bg_id, bg_off, alpha

    def _upload(self, sections):
        """Per-batch host decisions -> device tensors through ONE host->device copy: ``sections`` is a list of
        (kind, values) with kind "i" (int32), "f" (float32) or "l" (int64); returns one device tensor per section, views of
        a single packed buffer.  (One ``torch.tensor(...).to(device)`` per array was a dozen small copies per batch --
        4 % of a MobileNet step.)"""
        words, spans = [], []
        off = 0
        for kind, vals in sections:
            a = np.ascontiguousarray(vals, dtype={"i": np.int32, "f": np.float32, "l": np.int64}[kind])
            if kind == "l" and off % 2:
                words.append(np.zeros(1, np.int32))      # int64 views need an 8-byte aligned offset
                off += 1
            raw = a.view(np.int32).reshape(-1)
            spans.append((kind, off, raw.size, a.shape))
            words.append(raw)
            off += raw.size
        return self._to_device(self._stage(words, off), spans)

    RING = 8         # pinned staging buffers: a slot is reused only after the copy that read it has executed

    def _stage(self, words, nwords):
        """The packed words of a batch -> a pinned ring slot (device banks) or a plain host array; safe to call from the
        prefetch thread."""
        if not self.audio.is_cuda:
            return torch.from_numpy(np.concatenate(words) if words else np.zeros(0, np.int32)), None
        if len(self._ring) < self.RING:
            self._ring.append([torch.empty(max(nwords, 4096), dtype=torch.int32).pin_memory(), None])
        slot = self._ring[self._ring_next % self.RING]
        self._ring_next += 1
        if slot[1] is not None:
            slot[1].synchronize()             # the copy issued from this slot RING batches ago
            slot[1] = None
        if slot[0].numel() < nwords:
            slot[0] = torch.empty(nwords, dtype=torch.int32).pin_memory()
        host = slot[0][:nwords]
        np.concatenate(words, out=host.numpy())
        return host, slot

    def _to_device(self, staged, spans):
        host, slot = staged
        dev = self.audio.device
        buf = host.to(dev, non_blocking=True)
        if slot is not None:
            slot[1] = torch.cuda.Event()
            slot[1].record()
        out = []
        for kind, o, n, shape in spans:
            v = buf[o:o + n]
            out.append(v if kind == "i" else v.view(torch.float32 if kind == "f" else torch.int64).reshape(shape))
        return out

    def _launch(self, clip_ids, rows, shift, src_end, sigma, sp, lout, dst_off=None, extra=()):
        """One ``howl_collate_augment_window`` launch: batch row r takes samples [shift[r], src_end[r]) of clip
        ``clip_ids[rows[r]]`` (mixed with its background first, noise added), at column ``dst_off[r]``.  ``extra``: more
        (kind, values) sections to ride in the same upload (labels, lengths); returns (audio, [extra tensors])."""
        pick = lambda a: [a[k] for k in rows]
        self._calls += 1
        bank_rows = clip_ids if self.row_offsets is None else [self.row_offsets[i] for i in clip_ids]
        sections = [("i", pick(bank_rows)), ("i", src_end), ("i", shift), ("i", [1] * len(rows)), ("f", pick(sigma)),
                    ("f", pick(sp))]
        has_mix = self.last_mix is not None and any(a != 0.0 for a in self.last_mix[2])
        if has_mix:
            sections += [("i", pick(self.last_mix[0])), ("i", pick(self.last_mix[1])), ("f", pick(self.last_mix[2]))]
        if dst_off is not None:
            sections.append(("i", dst_off))
        n_own = len(sections)
        dev_t = self._upload(sections + list(extra))
        idx, end, sh, ones, sg, spp = dev_t[:6]
        mix = (self.bg_audio, dev_t[6], dev_t[7], dev_t[8]) if has_mix else None
        audio = ops.collate_augment(self.audio, idx, end, sh, ones, sg, spp, ((self._seed << 32) ^ self._calls ^ self._replica_key) & 0xFFFFFFFFFFFFFFFF, lout, mix=mix,
                                    dst_off=dev_t[n_own - 1] if dst_off is not None else None)
        return audio, dev_t[n_own:]

    def prepare(self, clip_ids):
        """Host half of ``__call__``: the draws of the batch (reference order), the length sort and the packed staging buffer.
        No device work, so a worker thread may run it for batch k + 1 while batch k trains (``prefetch``); batches must be
        prepared in the order they are consumed (the draws are one stream)."""
        # DatasetMixer's rejection loops stay scalar draws; so does a generator that is not a Mersenne Twister with an
        # exportable state (a test double assigned to ``self.rand``)
        if self.bg_audio is not None or not (self._rand is random or isinstance(self._rand, random.Random)):
            return ("scalar", list(clip_ids))
        ids = np.asarray(clip_ids, dtype=np.int64)
        lens, shift, head, sigma, sp = self.draw_arrays(ids)
        out_len = lens - shift
        order = np.argsort(-out_len, kind="stable")                              # batchify: longest first (stable)
        first = np.where(head[order] != 0, shift[order], 0)                      # head crop drops the first w samples,
        out_sorted = out_len[order]                                              # tail crop the last w
        self._calls += 1
        rows = ids[order] if self._rows_np is None else self._rows_np[ids[order]]
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.int32)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        n = ids.size
        words = [i32(rows), i32(first + out_sorted), i32(first), np.ones(n, np.int32), f32(sigma[order]), f32(sp[order])]
        spans = [("i", 0, n, (n,)), ("i", n, n, (n,)), ("i", 2 * n, n, (n,)), ("i", 3 * n, n, (n,)), ("f", 4 * n, n, (n,)),
                 ("f", 5 * n, n, (n,)), ("l", 6 * n, 2 * n, (n,))]
        words.append(np.ascontiguousarray(out_sorted, dtype=np.int64).view(np.int32))
        if self._labels_np is not None:
            words.append(self._labels_np[ids[order]].view(np.int32))
            spans.append(("l", 8 * n, 2 * n, (n,)))
        lout = int(out_sorted[0]) if n else 0
        seed = ((self._seed << 32) ^ self._calls ^ self._replica_key) & 0xFFFFFFFFFFFFFFFF
        return ("packed", self._stage(words, sum(w.size for w in words)), spans, lout, seed)

    def launch(self, prepared) -> ClassificationBatch:
        """Device half of ``__call__``: one host->device copy and one ``howl_collate_augment_window`` launch on the current stream."""
        if prepared[0] == "scalar":
            return self._call_scalar(prepared[1])
        _, staged, spans, lout, seed = prepared
        self.last_max_len = lout
        dev_t = self._to_device(staged, spans)
        idx, end, sh, ones, sg, spp, lengths = dev_t[:7]
        audio = ops.collate_augment(self.audio, idx, end, sh, ones, sg, spp, seed, lout)
        return ClassificationBatch(audio, dev_t[7] if self._labels_np is not None else None, lengths)

    def __call__(self, clip_ids) -> ClassificationBatch:
        """compose(truncate_length, Timeshift, Noise, batchify) (pretrain_gsc.py:78-80)."""
        return self.launch(self.prepare(clip_ids))

    def prefetch(self, id_batches, depth: int = 2):
        """Generator over ``launch(prepare(ids))`` for every id list of ``id_batches`` with the host half running ``depth``
        batches ahead in a worker thread -- the reference's DataLoader workers, as one thread: same draws in the same order
        (the worker prepares the batches sequentially), the consumer only copies and launches.  The collate must own its
        ``random`` stream (``seed=...``): a worker drawing from the global ``random`` module would race the main thread's
        own draws (VTLP's alpha)."""
        if self._rand is random:
            raise ValueError("DeviceCollate.prefetch needs a private stream: construct with seed=...")
        if depth > self.RING - 2:
            # prepare() stages into a ring of RING pinned buffers; a slot is only protected from re-use once launch() has issued
            # its copy, so the worker may run at most RING - 2 batches ahead (queue depth + the one it holds + the one launching)
            raise ValueError(f"DeviceCollate.prefetch: depth {depth} exceeds the staging ring ({self.RING} slots: depth <= {self.RING - 2})")
        q = queue.Queue(maxsize=max(depth, 1))
        stop = threading.Event()

        def work():
            try:
                for ids in id_batches:
                    item = self.prepare(ids)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                    if stop.is_set():
                        return
                q.put(None)
            except BaseException as e:      # surfaces in the consumer
                q.put(e)

        t = threading.Thread(target=work, name="howl-collate-prefetch", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield self.launch(item)
        finally:
            stop.set()
            t.join(timeout=5.0)

    def _call_scalar(self, clip_ids) -> ClassificationBatch:
        """``__call__`` with per-example Python draws (the form that also carries DatasetMixer)."""
        clip_ids = list(clip_ids)
        lens, shift, head, sigma, sp = self.draw(clip_ids)
        out_len = [l - w for l, w in zip(lens, shift)]
        self.last_max_len = max(out_len)
        order = sorted(range(len(clip_ids)), key=lambda k: -out_len[k])          # batchify: longest first (stable)
        first = [shift[k] if head[k] else 0 for k in order]                      # head crop drops the first w samples,
        extra = [("l", [out_len[k] for k in order])]
        if self.labels_host is not None:
            extra.append(("l", [self.labels_host[clip_ids[k]] for k in order]))
        audio, ex = self._launch(clip_ids, order, first, [f + out_len[k] for f, k in zip(first, order)], sigma, sp,
                                 max(out_len), extra=extra)                      # tail crop the last w
        return ClassificationBatch(audio, ex[1] if self.labels_host is not None else None, ex[0])

    def frame_batch(self, examples, batchifier) -> ClassificationBatch:
        """compose([DatasetMixer,] Timeshift, Noise, WakeWordFrameBatchifier) (train.py:211-229) on device clips: the
        augmentation draws come first (per batch, reference order), the batchifier then picks one window per augmented
        clip -- word end times are NOT corrected for a head crop, as in the reference -- and one launch produces the batch."""
        from howl_amd.data.transform.batchifier import DeviceClip
        clip_ids = [ex.clip_id for ex in examples]
        lens, shift, head, sigma, sp = self.draw(clip_ids)
        cropped = [DeviceClip(ex.clip_id, l - w, ex.timestamp_label_map, ex.transcription)
                   for ex, l, w in zip(examples, lens, shift)]
        plan = batchifier.plan(cropped)
        first = [(shift[k] if head[k] else 0) + a for k, a in zip(plan.source, plan.start)]
        audio, ex = self._launch(clip_ids, plan.source, first, [f + n for f, n in zip(first, plan.length)], sigma, sp,
                                 plan.width, dst_off=plan.dst_off, extra=[("l", list(plan.labels))])
        return ClassificationBatch(audio, ex[0], torch.tensor(plan.length))

    def sequence_batch(self, examples, batchifier):
        """compose([DatasetMixer,] Timeshift, Noise, AudioSequenceBatchifier) (train.py:206-229): whole augmented clips,
        ``np.argsort(-lengths)`` order, zero padded on the right; labels from the batchifier's tokenizer."""
        from howl_amd.data.common.batch import SequenceBatch
        clip_ids = [ex.clip_id for ex in examples]
        lens, shift, head, sigma, sp = self.draw(clip_ids)
        out_len = [l - w for l, w in zip(lens, shift)]
        order, labels, label_lengths = batchifier.order_and_labels(examples, out_len)
        first = [shift[k] if head[k] else 0 for k in order]
        audio, _ = self._launch(clip_ids, order, first, [f + out_len[k] for f, k in zip(first, order)], sigma, sp, max(out_len))
        return SequenceBatch(audio, labels, torch.tensor([out_len[k] for k in order]), label_lengths)
