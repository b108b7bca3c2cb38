"""Streaming decision logic over model outputs: ``InferenceEngine`` / ``FrameInferenceEngine`` with the reference's
API and finite-state machine (``howl/model/inference.py:19-267``), running on the MI355X hot path.

``FrameInferenceEngine.infer`` evaluates ALL strided windows of a clip in one batched launch of the fused frontend +
model (instead of one launch and one device->host sync per 63 ms stride, ``inference.py:247-261``) and then replays
the reference's per-window loop on the host, so ``label_history`` / ``pred_history`` and the early exit are identical.
Models that carry streaming state between windows keep the sequential path.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from howl_amd.context import InferenceContext
from howl_amd.data.transform.operator import ZmuvTransform
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.settings import SETTINGS
from howl_amd.utils import audio_utils

from .base import RegisteredModel
from .decision import ProbabilitySmoother, SequenceMatcher

__all__ = ["FrameInferenceEngine", "InferenceEngine"]


class InferenceEngine:
    """Sequential-model engine: one forward over the whole clip, then the frame-by-frame decision logic on the host."""

    def __init__(self, model: RegisteredModel, zmuv_transform: ZmuvTransform, context: InferenceContext,
                 time_provider=time.time):
        cfg = SETTINGS.inference_engine
        self.model, self.zmuv, self.context, self.settings = model, zmuv_transform, context, cfg
        self.std = StandardAudioTransform().eval()
        self.time_provider = time_provider
        self.sample_rate = SETTINGS.audio.sample_rate
        self.blank_idx = context.blank_label
        # per-class reweighting of the probabilities (missing entries count as 1)
        self.inference_weights = 1
        if cfg.inference_weights:
            w = np.ones(context.num_labels)
            w[:len(cfg.inference_weights)] = cfg.inference_weights
            self.inference_weights = w
        self.coloring = context.coloring
        negative = context.negative_label
        if self.coloring:
            negative = self.coloring.color_map[negative]
        self.negative_label = negative
        self.threshold = cfg.inference_threshold
        self.inference_window_ms = cfg.inference_window_ms
        self.smoothing_window_ms = cfg.smoothing_window_ms
        self.tolerance_window_ms = cfg.tolerance_window_ms
        self.sequence = cfg.inference_sequence
        self._smoother = ProbabilitySmoother(self.smoothing_window_ms, self.threshold, negative,
                                             self.coloring.color_map if self.coloring else None)
        self._matcher = SequenceMatcher(self.sequence, self.inference_window_ms, self.tolerance_window_ms)
        self.curr_time = 0
        self.label_history = []
        self.reset()

    # the reference exposes both histories as plain attributes; pred_history lives in the smoother
    @property
    def pred_history(self):
        return self._smoother.frames

    @pred_history.setter
    def pred_history(self, frames):
        self._smoother.frames = list(frames)

    def to(self, device: torch.device):
        self.model, self.zmuv = self.model.to(device), self.zmuv.to(device)
        return self

    def reset(self):
        self.model.streaming_state = None
        self.curr_time = 0
        self._smoother.clear()
        self.label_history = []

    def _now_ms(self, curr_time):
        return self.time_provider() * 1000 if curr_time is None else curr_time

    def append_label(self, label: int, curr_time: float = None):
        self.label_history.append((self._now_ms(curr_time), label))

    def sequence_present(self, curr_time: float = None) -> bool:
        # the matcher reads the settings through the engine so that tests / callers may retune them after construction
        self._matcher.sequence, self._matcher.window_ms = self.sequence, self.inference_window_ms
        self._matcher.tolerance_ms = self.tolerance_window_ms
        return self._matcher.present(self.label_history, self._now_ms(curr_time))

    def _append_probability_frame(self, prediction: np.ndarray, curr_time: float = None) -> int:
        now = self._now_ms(curr_time)
        self._smoother.window_ms, self._smoother.threshold = self.smoothing_window_ms, self.threshold
        label = self._smoother.push(now, prediction)
        self.label_history.append((now, label))
        return label

    def _weighted(self, prediction: np.ndarray) -> np.ndarray:
        # the reference multiplies in place (inference.py:200,263): the product is rounded back to the probabilities' own
        # dtype (fp32) before the renormalisation, which matters for comparisons right at the threshold
        prediction = (prediction * self.inference_weights).astype(prediction.dtype, copy=False)
        return prediction / prediction.sum()

    @torch.no_grad()
    def infer(self, audio_data: torch.Tensor) -> bool:
        """Whole clip as one batch through a sequential model (``inference.py:179-211``)."""
        delta_ms = int(audio_data.size(-1) / self.sample_rate * 1000)
        self.std = self.std.to(audio_data.device)
        transformed = self.std.log_mel_for_model(audio_data.unsqueeze(0), self.zmuv)
        predictions = self.model(transformed, lengths=None)
        predictions = F.softmax(predictions, -1).squeeze(1).cpu().numpy()   # one device->host copy for all frames
        sequence_present = False
        delta_ms /= len(predictions)
        for prediction in predictions:
            prediction = self._weighted(prediction)
            self.curr_time += delta_ms
            if np.argmax(prediction) == self.blank_idx:
                continue
            self._append_probability_frame(prediction, curr_time=self.curr_time)
            if self.sequence_present(self.curr_time):
                sequence_present = True
                break
        return sequence_present


class FrameInferenceEngine(InferenceEngine):
    def __init__(self, max_window_size_ms: int, eval_stride_size_ms: int, *args):
        super().__init__(*args)
        self.max_window_size_ms, self.eval_stride_size_ms = max_window_size_ms, eval_stride_size_ms

    def _stateless(self) -> bool:
        return not self.model.is_streaming or type(self.model).streaming_state is RegisteredModel.streaming_state

    @torch.no_grad()
    def window_probabilities(self, audio_data: torch.Tensor) -> np.ndarray:
        """softmax(model(zmuv(std(window)))) for every complete strided window, one batched launch -> (W, C)."""
        starts, chunk = audio_utils.stride_starts(audio_data.size(-1), self.max_window_size_ms, self.eval_stride_size_ms,
                                                  self.sample_rate)
        if not starts or chunk < 1000:
            return np.zeros((0, self.context.num_labels), np.float32)
        self.std = self.std.to(audio_data.device)
        stride_sz = starts[1] - starts[0] if len(starts) > 1 else chunk
        flat = audio_data.reshape(-1).contiguous()
        windows = flat.as_strided((len(starts), chunk), (stride_sz, 1))   # overlapping views, no copy
        feats = self.std.log_mel_for_model(windows, self.zmuv)
        lengths = self.std.compute_lengths(torch.full((len(starts),), chunk, device=audio_data.device))
        return self.model(feats, lengths).softmax(-1).cpu().numpy()

    MAX_WINDOWS_PER_LAUNCH = 8192

    @torch.no_grad()
    def window_probabilities_many(self, clips) -> list:
        """``window_probabilities`` of several clips with ONE frontend launch, one model forward and one device->host copy for all
        of their windows (clips whose windows have the same length share a batch: every clip at least one window long does)."""
        out = [None] * len(clips)
        groups = {}
        for i, clip in enumerate(clips):
            starts, chunk = audio_utils.stride_starts(clip.size(-1), self.max_window_size_ms, self.eval_stride_size_ms, self.sample_rate)
            if not starts or chunk < 1000:
                out[i] = np.zeros((0, self.context.num_labels), np.float32)
                continue
            stride_sz = starts[1] - starts[0] if len(starts) > 1 else chunk
            groups.setdefault((chunk, clip.device), []).append((i, len(starts), stride_sz))
        for (chunk, device), members in groups.items():
            self.std = self.std.to(device)
            views = [clips[i].reshape(-1).contiguous().as_strided((n, chunk), (stride_sz, 1)) for i, n, stride_sz in members]
            # at most MAX_WINDOWS_PER_LAUNCH windows per forward: 64 clips of 30 s at a 63-ms stride are ~30 k windows, i.e. 1 GB of
            # window copies and as much again per activation tensor -- the pass stays O(cap), not O(dataset)
            pieces, held, parts = [], 0, []
            def flush():
                nonlocal pieces, held
                if not pieces:
                    return
                windows = pieces[0] if len(pieces) == 1 else torch.cat(pieces)      # (windows of this launch, chunk): the only copy
                feats = self.std.log_mel_for_model(windows, self.zmuv)
                lengths = self.std.compute_lengths(torch.full((windows.size(0),), chunk, device=device))
                parts.append(self.model(feats, lengths).softmax(-1).cpu().numpy())
                pieces, held = [], 0
            for v in views:
                lo = 0
                while lo < v.size(0):
                    take = min(v.size(0) - lo, self.MAX_WINDOWS_PER_LAUNCH - held)
                    pieces.append(v[lo:lo + take])
                    held += take
                    lo += take
                    if held == self.MAX_WINDOWS_PER_LAUNCH:
                        flush()
            flush()
            probs = parts[0] if len(parts) == 1 else np.concatenate(parts)
            lo = 0
            for i, n, _ in members:
                out[i] = probs[lo:lo + n]
                lo += n
        return out

    def _run_fsm(self, probs) -> bool:
        sequence_present = False
        for prediction in probs:
            self._append_probability_frame(self._weighted(prediction), curr_time=self.curr_time)
            self.curr_time += self.eval_stride_size_ms
            if self.sequence_present(self.curr_time):
                sequence_present = True
                break
        return sequence_present

    @torch.no_grad()
    def infer_many(self, clips) -> list:
        """``[reset(); infer(clip) for clip in clips]`` with the windows of ALL clips scored in one batch (an evaluation pass over a
        dataset, train.py:42-94, is host-bound clip by clip: one launch chain and one host copy per clip); the label
        histories, smoothing and sequence search run per clip exactly as ``infer`` runs them.  Leaves the engine reset."""
        if not self._stateless():
            res = []
            for clip in clips:
                self.reset()
                res.append(bool(self._infer_sequential(clip)))
            self.reset()
            return res
        res = []
        for probs in self.window_probabilities_many(clips):
            self.reset()
            res.append(self._run_fsm(probs))
        self.reset()
        return res

    @torch.no_grad()
    def infer(self, audio_data: torch.Tensor) -> bool:
        if not self._stateless():
            return self._infer_sequential(audio_data)
        probs = self.window_probabilities(audio_data)
        sequence_present = False
        for prediction in probs:
            self._append_probability_frame(self._weighted(prediction), curr_time=self.curr_time)
            self.curr_time += self.eval_stride_size_ms
            if self.sequence_present(self.curr_time):
                sequence_present = True
                break
        return sequence_present

    def _infer_sequential(self, audio_data: torch.Tensor) -> bool:
        sequence_present = False
        for window in audio_utils.stride(audio_data, self.max_window_size_ms, self.eval_stride_size_ms, self.sample_rate):
            if window.size(-1) < 1000:
                break
            self.ingest_frame(window.squeeze(0), self.curr_time)
            self.curr_time += self.eval_stride_size_ms
            if self.sequence_present(self.curr_time):
                sequence_present = True
                break
        return sequence_present

    @torch.no_grad()
    def ingest_frame(self, frame: torch.Tensor, curr_time: float = None) -> int:
        """One window, as the live client feeds it (``inference.py:247-267``)."""
        self.std = self.std.to(frame.device)
        lengths = torch.tensor([frame.size(-1)]).to(frame.device)
        transformed_lengths = self.std.compute_lengths(lengths)
        transformed_frame = self.std.log_mel_for_model(frame.unsqueeze(0), self.zmuv)
        prediction = self.model(transformed_frame, transformed_lengths).softmax(-1)[0].cpu().numpy()
        return self._append_probability_frame(self._weighted(prediction), curr_time=curr_time)
