"""-m gpu: the library from several host threads and HIP streams at once.

The reference's serving path runs the engine on the PortAudio callback thread (howl/client/howl_client.py:68-94,129-137 ->
FrameInferenceEngine.infer, howl/model/inference.py:223-267) while the main thread does whatever the application does.  The
library is written for that -- every entry point takes the caller's stream, the error text and the per-kernel LDS-limit tables
are thread_local, the profile table and the side lanes are mutex-guarded (csrc/capi.hip) -- and this file checks it: results
bit-identical to the single-threaded ones (golden G8 / a solo run), whatever runs next to them."""
import ctypes
import threading

import numpy as np
import pytest
import torch

from gpu_util import DEV, make_res8, t

pytestmark = pytest.mark.gpu


def _g8_engine(golden):
    from howl_amd.context import InferenceContext
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.model.inference import FrameInferenceEngine
    g4 = golden("g4_zmuv")
    ctx = InferenceContext(["hey", "fire", "fox"], token_type="word")
    model = make_res8(ctx.num_labels, train=False).streaming()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.mean.copy_(t(g4["mean"]))
    zmuv.mean2.copy_(t(g4["mean2"]))
    zmuv.total.copy_(t(g4["total"]))
    return FrameInferenceEngine(500, 63, model, zmuv, ctx)


def _infer_loop(engine, clip, g, rounds, out, key, start):
    """What the client's callback thread does, on a stream of its own: reset + infer, history compared with G8 every round."""
    try:
        stream = torch.cuda.Stream(device=DEV)
        with torch.cuda.stream(stream):
            start.wait()
            ok = 0
            for _ in range(rounds):
                engine.reset()
                present = engine.infer(clip)
                hist = np.array(engine.label_history, dtype=np.float64)
                if bool(present) == bool(g["present"]) and hist.shape == g["label_history"].shape and \
                        np.array_equal(hist, g["label_history"]):
                    ok += 1
            stream.synchronize()
        out[key] = ok
    except BaseException as e:      # surfaces in the test thread
        out[key] = e


def _train_steps(steps, stream=None):
    """A res8 training run of `steps` fused steps from closed-form weights; returns the flat parameter buffer."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedRes8Trainer
    from howl_amd.utils.synth import synthetic_pcm
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        B, C = 64, 12
        pcm = synthetic_pcm(B, 16000).to(DEV)
        labels = (torch.arange(B) % C).to(DEV)
        std = StandardAudioTransform().to(DEV).eval()
        zmuv = ZmuvTransform().to(DEV)
        zmuv.update(std(pcm[:4]))
        model = make_res8(C)
        tr = FusedRes8Trainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
        losses = [tr.step(pcm, labels) for _ in range(steps)]
        torch.cuda.current_stream().synchronize()
        return tr.fp.flat.clone(), torch.stack([l.reshape(()) for l in losses]).clone()


def test_engine_on_a_worker_thread_while_the_main_thread_trains(golden):
    from howl_amd.settings import SETTINGS
    SETTINGS.inference_engine.inference_sequence = [0, 1, 2]
    g = golden("g8_frame_engine")
    solo_flat, solo_losses = _train_steps(12)
    engine = _g8_engine(golden)
    clip = t(g["clip"]).to(DEV)
    out, start = {}, threading.Event()
    th = threading.Thread(target=_infer_loop, args=(engine, clip, g, 25, out, "engine", start))
    th.start()
    start.set()
    flat, losses = _train_steps(12, torch.cuda.Stream(device=DEV))      # a different model, another stream, this thread
    th.join(timeout=300)
    assert not th.is_alive()
    assert out["engine"] == 25, out["engine"]                            # every round's label history == G8, bit for bit
    assert torch.equal(flat, solo_flat) and torch.equal(losses, solo_losses)


def test_two_engines_on_two_streams_and_threads(golden):
    from howl_amd.settings import SETTINGS
    SETTINGS.inference_engine.inference_sequence = [0, 1, 2]
    g = golden("g8_frame_engine")
    clip = t(g["clip"]).to(DEV)
    engines = [_g8_engine(golden), _g8_engine(golden)]
    out, start = {}, threading.Event()
    threads = [threading.Thread(target=_infer_loop, args=(e, clip, g, 20, out, k, start)) for k, e in enumerate(engines)]
    for th in threads:
        th.start()
    start.set()
    for th in threads:
        th.join(timeout=300)
        assert not th.is_alive()
    assert out == {0: 20, 1: 20}, out


def test_last_error_is_per_thread():
    """howl_last_error() is the calling thread's: a failing call on thread B leaves thread A's text alone (and vice versa)."""
    from howl_amd import lib
    L = lib.get()
    with pytest.raises(lib.HowlHipError, match="null pointer"):
        L.call("howl_logmel_fwd", None, 1, 16000, 16000, None, 40, 1e-7, None, None, 0, None)
    mine = L.cdll.howl_last_error().decode()
    assert "howl_logmel_fwd" in mine
    seen = {}

    def other():
        try:
            seen["before"] = L.cdll.howl_last_error().decode()          # a fresh thread has no error text
            try:
                L.call("howl_ctc_loss", None, 0, 0, 10, 1, 5, None, 0, 0, None, None, 4, None, None, None, 0, 0, None, 0, None)
            except lib.HowlHipError as e:
                seen["raised"] = str(e)
            seen["after"] = L.cdll.howl_last_error().decode()
        except BaseException as e:
            seen["exc"] = e

    th = threading.Thread(target=other)
    th.start()
    th.join(timeout=60)
    assert "exc" not in seen, seen
    assert seen["before"] == "" and "howl_ctc_loss" in seen["raised"] and "howl_ctc_loss" in seen["after"]
    assert L.cdll.howl_last_error().decode() == mine


def test_shutdown_releases_side_lanes_and_later_calls_make_new_ones(monkeypatch):
    """howl_shutdown destroys the side queues / events of every host thread; the next call that wants a lane makes a
    fresh one.  The only user of a lane is howl_seq_lstm_bwd under HOWL_LSTM_RIDE=lane (the measured-and-not-kept two-queue form
    of round 5): a seq-lstm step that way, from this thread and from a worker thread, before and after a shutdown, gives the
    bits of the default (rider) form."""
    from howl_amd import lib
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    from oracle import models as om
    B, C = 512, 5
    pcm = synthetic_pcm(B, 8000).to(DEV)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4]))
    lengths, targets, tl = torch.full((B,), 38), torch.tensor([[0, 1, 2]] * B), torch.tensor([3] * B)

    def step(out, key):
        try:
            model = RegisteredModel.find_registered_class("seq-lstm")(C)
            model.load_state_dict({k: v.clone() for k, v in om.lstm_init(C).items()})
            tr = FusedTrainer(model.to(DEV).train(), std, zmuv, lr=1e-3, weight_decay=1e-5)
            tr.step_sequence(pcm, lengths, targets, tl, 4)
            torch.cuda.synchronize()
            out[key] = tr.fp.flat.clone()
        except BaseException as e:
            out[key] = e

    out = {}
    step(out, "rider")
    monkeypatch.setenv("HOWL_LSTM_RIDE", "lane")
    step(out, "lane")
    th = threading.Thread(target=step, args=(out, "lane-thread"))
    th.start()
    th.join(timeout=120)
    lib.get().call("howl_shutdown")
    step(out, "lane-after-shutdown")
    th = threading.Thread(target=step, args=(out, "lane-thread-after-shutdown"))
    th.start()
    th.join(timeout=120)
    for k, v in out.items():
        assert torch.is_tensor(v), (k, v)
        assert torch.equal(v, out["rider"]), k
