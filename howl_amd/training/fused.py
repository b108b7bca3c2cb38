"""Fused training step: frontend -> forward -> loss -> backward -> (RCCL all-reduce) -> AdamW.  Frame objective
(res8, mobilenet: cross-entropy) and sequence objective (seq-lstm: log_softmax + CTC).

This is the loop body of ``training/run/pretrain_gsc.py:124-133`` / ``training/run/train.py:286-302`` with every
stage a C-ABI call on the current HIP stream, gradients written straight into one flat fp32 buffer (the unit of the
single data-parallel all-reduce per step) and no host synchronisation (the loss stays on the device).
"""
import os

import torch

from howl_amd import ops, parallel


class FlatParams:
    """Re-homes a list of parameters into one contiguous buffer (and a matching flat gradient buffer)."""

    def __init__(self, params):
        self.params = list(params)
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.numel = sum(sizes)
        self.flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grad_views = []
        off = 0
        with torch.no_grad():
            for p, n in zip(self.params, sizes):
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                self.grad_views.append(self.grad[off:off + n].view(p.shape))
                off += n

    def attach_grads(self):
        """Expose the flat gradient through ``p.grad`` (for torch optimisers / inspection)."""
        for p, g in zip(self.params, self.grad_views):
            p.grad = g


class FusedTrainer:
    """Works with any model that exposes ``hot_parameters()`` (flat-buffer order), ``_launch_forward(feat)`` and
    ``_launch_backward(feat, dlogits, out_grads=views)``: ``Res8``, ``MobileNetClassifier`` and ``SimpleLstm`` (``step``; the
    latter with frame lengths), and with ``SequentialLstm`` (``step_sequence``: ``_launch_forward(feat, lengths)`` /
    ``_launch_backward(dscores, out_grads)``).  With more than one rank the flat gradient is summed over RCCL before the
    fused AdamW (which applies 1/world); res8 overlaps that all-reduce with the tail of its backward pass."""

    def __init__(self, model, std_transform, zmuv_transform, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8,
                 process_group=None, late_grads=None):
        """``late_grads`` (data-parallel res8 only; default from ``HOWL_DP_LATE``, else "merged" -- the schedule with fewer
        unknowns until a SCALE record on the 8-GPU node says which wins):
        "overlap" -- two-part backward, TWO collectives per step: everything but conv0's 405 gradients is reduced under
                     conv0's weight-gradient kernels (async), the 405 floats after them (on the critical path);
        "merged"  -- one-part backward, ONE collective per step over the whole flat buffer (nothing overlapped: pays when
                     the second small collective's latency exceeds what the overlap hides, i.e. part 2 < ~60 us)."""
        self.model, self.std, self.zmuv = model, std_transform, zmuv_transform
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.fp = FlatParams(model.hot_parameters())
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.step_count = 0
        self.group = process_group
        self.rank, self.world = parallel.world_info(process_group)
        self.late_grads = late_grads or os.environ.get("HOWL_DP_LATE", "merged")
        if self.late_grads not in ("overlap", "merged"):
            raise ValueError(f"late_grads / HOWL_DP_LATE must be 'overlap' or 'merged', got {self.late_grads!r}")
        self.collectives_last_step = 0     # gradient collectives the last step issued (bench.py reports it)
        self.collectives_last_step_reduced = 0   # ... the last step that did reduce (skip_allreduce steps excluded)
        self.skip_allreduce = False        # measurement only (bench.py: step time with vs without the collectives)
        self._ahead = None                 # (next_audio, its features, keep-alive) left by step_sequence(next_audio=...)

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank ``src``'s weights and BatchNorm buffers."""
        parallel.broadcast_([self.fp.flat] + list(self.model.buffers()), src, self.group)

    def features(self, audio):
        return self.std.log_mel_for_model(audio, self.zmuv)

    def step(self, audio, labels, lengths=None, max_frames=None):
        """One optimisation step on a (B, L) PCM batch; returns the (local) mean loss as a device tensor."""
        feat = self.features(audio)
        return self.step_on_features(feat, labels, lengths, max_frames)

    def step_on_features(self, feat, labels, lengths=None, max_frames=None):
        """``lengths`` / ``max_frames``: frame counts for models that pack their input (``SimpleLstm``); ignored otherwise."""
        bwd_kw = {}
        fused_xent = hasattr(self.model, "_launch_forward_xent") and self.model.num_labels <= self.model.XENT_MAX_LABELS
        if fused_xent:     # res8: the loss rides in the forward's last launch and the backward's second (two launches fewer)
            logits, nll, dlogits = self.model._launch_forward_xent(feat, labels)
            loss = torch.empty(1, dtype=torch.float32, device=logits.device)
            bwd_kw = dict(xent=(nll, loss))
        else:
            if getattr(self.model, "NEEDS_LENGTHS", False):
                logits = self.model._launch_forward(feat, lengths, max_frames)
            else:
                logits = self.model._launch_forward(feat)
            loss, dlogits = ops.xent(logits, labels)
        late = getattr(self.model, "LATE_GRAD_PARAMS", 0)
        if fused_xent and self.world == 1:
            # single replica: nothing sits between the backward's last fold and the optimiser, so the step rides in that launch
            # (howl_res8_bwd_xent's HowlAdamW: one launch and one pass over the gradients fewer)
            bwd_kw["adamw"] = (self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas, self.eps, self.weight_decay,
                               self.step_count + 1, 1.0)
        self.model.optimizer_step_done = False
        if self.skip_allreduce:
            self.model._launch_backward(feat, dlogits, out_grads=self.fp.grad_views, **bwd_kw)
            scale, self.collectives_last_step = 1.0 / self.world, 0
        elif self.world > 1 and late and self.late_grads == "overlap":
            # two-part backward: the all-reduce of everything but the first `late` parameters (res8: conv0.weight, whose
            # gradient needs the last data gradient) runs on the process group's stream while their kernels still execute
            n0 = sum(p.numel() for p in self.fp.params[:late])
            # split on a 256-byte boundary of the flat buffer (RCCL's vectorised paths want aligned base addresses): the few
            # gradients of the next parameter that move to the late collective are final after part 1 as well
            n0 = min(-(-n0 // 64) * 64, self.fp.numel)
            # what that rounding relies on: an aligned base, and that the parameters the spill-over touches are complete
            # after part 1 (the model declares how many leading parameters part 2 still writes: only the first `late`)
            assert not self.fp.grad.is_cuda or self.fp.grad.data_ptr() % 256 == 0, \
                "flat gradient buffer must be 256-byte aligned for the split"
            assert sum(p.numel() for p in self.fp.params[:late + 1]) >= n0 or late + 1 >= len(self.fp.params), \
                "the rounded split must end inside the first parameter that is final after part 1"
            self.model._launch_backward(feat, dlogits, out_grads=self.fp.grad_views, part=1, **bwd_kw)
            pending = parallel.allreduce_start_(self.fp.grad[n0:], self.group)
            self.model._launch_backward(feat, dlogits, out_grads=self.fp.grad_views, part=2, **bwd_kw)
            scale = parallel.allreduce_sum_(self.fp.grad[:n0], self.group)
            pending.wait()
            self.collectives_last_step = self.collectives_last_step_reduced = 2
        else:
            self.model._launch_backward(feat, dlogits, out_grads=self.fp.grad_views, **bwd_kw)
            scale = parallel.allreduce_sum_(self.fp.grad, self.group)
            self.collectives_last_step = self.collectives_last_step_reduced = 1 if self.world > 1 else 0
        self.step_count += 1
        if not getattr(self.model, "optimizer_step_done", False):
            ops.adamw_step(self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas, self.eps, self.weight_decay,
                           self.step_count, scale)
        self.last_logits = logits
        return loss

    def step_sequence(self, audio, frame_lengths, targets, target_lengths, blank, max_target=None, max_frames=None, next_audio=None):
        """One optimisation step of the sequence objective (train.py:286-302 with ``objective=ctc``) on a (B, L) PCM batch
        sorted by decreasing length: ``frame_lengths`` = ``StandardAudioTransform.compute_lengths`` of the sample counts,
        ``targets`` the padded (B, Lmax) label matrix.  The length vectors may live on the host (as the reference's batches
        do: copied over each step) or on the device; for device-resident ones pass their maxima (``max_frames``,
        ``max_target``) so that nothing is read back.  Returns the mean CTC loss as a device tensor.
        ``next_audio``: the PCM batch of the NEXT step (a one-batch look-ahead, as a prefetching DataLoader gives): its frontend
        runs inside this step's forward call -- on the CUs the recurrence leaves idle -- and the next call, given the same
        tensor as ``audio``, finds its features ready.  Same launches' worth of work per step, one of them off the critical path;
        only for frontends without a pending VTLP draw (see ``log_mel_for_model_args``)."""
        feat = None
        if self._ahead is not None and self._ahead[0] is audio:
            feat = self._ahead[1]
        self._ahead = None
        if feat is None:
            feat = self.features(audio)
        nxt = None
        if next_audio is not None and getattr(self.model, "TAKES_NEXT_LOGMEL", False):
            nxt = self.std.log_mel_for_model_args(next_audio, self.zmuv)
        loss = self.step_sequence_on_features(feat, frame_lengths, targets, target_lengths, blank, max_target, max_frames,
                                              next_logmel=None if nxt is None else nxt[0])
        if nxt is not None:
            self._ahead = (next_audio, nxt[1], nxt[2])      # (the record's tensors stay alive with it)
        return loss

    def step_sequence_on_features(self, feat, frame_lengths, targets, target_lengths, blank, max_target=None, max_frames=None,
                                  next_logmel=None):
        if max_target is None:
            max_target = int(target_lengths.max()) if target_lengths.numel() else 0
        kw = {}
        if next_logmel is not None:
            kw["next_logmel"] = next_logmel
        if getattr(self.model, "TAKES_CTC", False):
            # short windows, many rows: head + log_softmax / CTC + the head's backward over the rows ride in ONE launch behind the
            # forward recurrence (howl_seq_head_ctc); the model says through `ctc_nll` whether that launch covered the batch
            dev = feat.device
            tg = targets.to(dev, torch.int64)
            kw["ctc"] = (tg if tg.stride(-1) == 1 else tg.contiguous(), target_lengths.to(dev, torch.int64).contiguous(), int(blank), max_target)
            if not ops.on_device(frame_lengths):
                frame_lengths = frame_lengths.to(torch.int64)
        scores = self.model._launch_forward(feat, frame_lengths, max_frames, **kw)   # (T_len, B, C) view of a (B, T_len, C) buffer
        if getattr(self.model, "ctc_nll", None) is not None:
            loss = torch.empty((), dtype=torch.float32, device=scores.device)
            adamw = None
            if self.world == 1:
                adamw = (self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas, self.eps, self.weight_decay, self.step_count + 1, 1.0)
            self.model._launch_backward(None, out_grads=self.fp.grad_views, ctc_mean=(self.model.ctc_nll, kw["ctc"][1], loss), adamw=adamw)
            return self._finish_sequence_step(scores, loss)
        # log-softmax + CTC + their backward as one launch at any clip length (128-frame windows beyond 128 frames); the batch
        # mean of the loss rides in the head's backward launch (HowlCtcMean).  Outside the kernel's range (C > 64, a target
        # of more than 31 labels, T > 8192): HowlHipError from the library -- no vendor kernels on the training path
        if "ctc" in kw:      # (already on the device for the fused launch's benefit: no second upload)
            targets, target_lengths = kw["ctc"][0], kw["ctc"][1]
        loss, dscores, nll, tl_dev = ops.ctc_loss_fwd_bwd(scores, targets, frame_lengths, target_lengths, blank, max_target,
                                                          defer_mean=True)
        ctc_mean = (nll, tl_dev, loss)
        # single replica: nothing sits between the backward's gradient fold and the optimiser, so the step rides in the fold
        # (howl_seq_lstm_bwd's HowlAdamW: one launch and one pass over the gradients fewer)
        adamw = None
        if self.world == 1:
            adamw = (self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas, self.eps, self.weight_decay, self.step_count + 1, 1.0)
        self.model._launch_backward(dscores, out_grads=self.fp.grad_views, ctc_mean=ctc_mean, adamw=adamw)
        return self._finish_sequence_step(scores, loss)

    def _finish_sequence_step(self, scores, loss):
        if self.skip_allreduce:
            scale, self.collectives_last_step = 1.0 / self.world, 0
        else:
            scale = parallel.allreduce_sum_(self.fp.grad, self.group)
            self.collectives_last_step = self.collectives_last_step_reduced = 1 if self.world > 1 else 0
        self.step_count += 1
        if not getattr(self.model, "optimizer_step_done", False):
            ops.adamw_step(self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas, self.eps, self.weight_decay,
                           self.step_count, scale)
        self.last_logits = scores
        return loss

    def decay_lr(self, factor):
        self.lr *= factor


FusedRes8Trainer = FusedTrainer
