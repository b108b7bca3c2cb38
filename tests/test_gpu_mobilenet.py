"""-m gpu parity of MobileNetClassifier ("mobilenet", BASELINE configs[4]) against the oracle restatement
(oracle/mobilenet.py -- parity unpinned against torchvision, see its header), through the Python drop-in boundary."""
import pytest
import torch

from gpu_util import DEV, maxerr
from mb_util import check_grads, oracle_step
from oracle import frontend as ofe
from oracle import mobilenet as omb
from oracle import models as om

pytestmark = pytest.mark.gpu


def make_mobilenet(C, train=True):
    from howl_amd.model import RegisteredModel
    model = RegisteredModel.find_registered_class("mobilenet")(C)
    sd = omb.mobilenet_init(C)
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    model = model.to(DEV)
    return (model.train() if train else model.eval()), sd


def test_registry_and_state_dict():
    from howl_amd.model import RegisteredModel
    assert "mobilenet" in RegisteredModel.registered_names()
    model, sd = make_mobilenet(12)
    x = torch.randn(4, 3, 40, 81, device=DEV)
    model(x, None)                                    # re-homes the parameters into the flat buffer
    out = model.state_dict()
    assert set(out) == set(sd)
    for k in sd:
        if "running" not in k and "num_batches" not in k:
            assert torch.equal(out[k].cpu(), sd[k]), k
    assert out["downsample.1.num_batches_tracked"].item() == 1


# (96, 101, 12): the first block has 203 k pixels -- multi-tile blocks, two-level arrival (more than 64 partial rows), 64 x 64 and
# 32-row tiles side by side; (7, 57, 3): nothing divides anything (odd batch, odd widths at every stride, 3 labels)
@pytest.mark.parametrize("B,T,C,dropout", [(8, 81, 12, True), (16, 41, 4, False), (7, 57, 3, True), (96, 101, 12, False),
                                            (9, 61, 35, False)])
def test_autograd_forward_backward_vs_oracle(B, T, C, dropout):
    torch.manual_seed(B + T)
    x = torch.randn(B, 3, 40, T) * 1.5
    labels = torch.arange(B) % C
    keep = (torch.rand(B, omb.LAST_CHANNEL) >= 0.2).float() if dropout else None
    model, sd = make_mobilenet(C)
    model.forced_keep_mask = keep
    model.dropout_p = 0.2 if dropout else 0.0
    logits = model(x.to(DEV), None)
    loss = torch.nn.functional.cross_entropy(logits, labels.to(DEV))
    loss.backward()

    ref, ref_grads, osd = oracle_step(sd, x, labels, keep)
    assert maxerr(logits, ref) < 1e-3
    assert torch.equal(logits.argmax(1).cpu(), ref.argmax(1))
    check_grads([p.grad for p in model.hot_parameters()], sd, x, labels, keep, ref_grads)
    got = model.state_dict()
    for l in omb.layer_table():
        assert maxerr(got[l["bn"] + ".running_var"], osd[l["bn"] + ".running_var"].detach()) < 1e-4

    # eval mode on the updated running statistics
    model.eval()
    with torch.no_grad():
        elog = model(x.to(DEV), None)
    esd = {k: v.detach().cpu().clone() for k, v in got.items()}
    eref = omb.mobilenet_forward(esd, x, False)
    assert maxerr(elog, eref) < 2e-4
    assert torch.equal(elog.argmax(1).cpu(), eref.argmax(1))


def test_fused_step_vs_oracle_at_baseline_size():
    """BASELINE configs[4] geometry per GPU at a reduced batch: PCM in -> updated weights out, two AdamW steps."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    B, L, C = 32, 16000, 12
    pcm = synthetic_pcm(B, L)
    labels = torch.arange(B) % C
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    model, sd = make_mobilenet(C)
    model.dropout_p = 0.0                              # the dropout draw is the only RNG in the step
    trainer = FusedTrainer(model, std, zmuv, lr=0.001, weight_decay=0.0)   # envs/mobilenet.env
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x = z(ofe.standard_audio_transform(pcm, fb))
    names = omb.mobilenet_param_names()
    opt = om.AdamWState([sd[n] for n in names], 0.001, 0.0)
    before = {k: v.clone() for k, v in sd.items()}
    loss = trainer.step(pcm.to(DEV), labels.to(DEV))
    ref_loss, ref_logits, ref_grads = om.train_step(lambda s, xx: omb.mobilenet_forward(s, xx, True), sd, names, opt, x, labels)
    assert maxerr(trainer.last_logits, ref_logits) < 1e-3
    assert torch.equal(trainer.last_logits.argmax(1).cpu(), ref_logits.argmax(1))
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    check_grads(trainer.fp.grad_views, before, x, labels, None, [ref_grads[n] for n in names], eps=2e-5)
    # the first AdamW step moves every weight by ~lr * sign(g): compare where both gradients agree on a solid value (a
    # gradient that is rounding noise, or that flipped with a ReLU6 mask, moves the weight the other way on one side)
    moved = agree = 0
    for n, p, g in zip(names, model.hot_parameters(), trainer.fp.grad_views):
        ref_g = ref_grads[n]
        solid = (ref_g.abs() > 1e-3 * ref_g.abs().max()) & ((g.detach().cpu() - ref_g).abs() < 0.1 * ref_g.abs())
        if solid.any():
            assert maxerr(p.detach().cpu()[solid], sd[n][solid]) < 1e-4, n   # lr = 1e-3
        moved += solid.numel()
        agree += int(solid.sum())
    assert agree > 0.5 * moved
    trainer.step(pcm.to(DEV), labels.to(DEV))          # second step: exercised for the determinism check below
    # determinism
    model2, _ = make_mobilenet(C)
    model2.dropout_p = 0.0
    trainer2 = FusedTrainer(model2, std, zmuv, lr=0.001, weight_decay=0.0)
    for step in range(2):
        trainer2.step(pcm.to(DEV), labels.to(DEV))
    assert torch.equal(trainer2.fp.flat, trainer.fp.flat)


def test_pretrain_gsc_entry_point_mobilenet(tmp_path, monkeypatch):
    """`python -m training.run.pretrain_gsc --model mobilenet` (envs/mobilenet.env: lr 0.001, weight decay 0) on generated
    clips: variable batch lengths from the device collate, fused step, eval-mode accuracy pass, checkpoint keys."""
    for k, v in dict(NUM_EPOCHS="2", BATCH_SIZE="32", MAX_WINDOW_SIZE_SECONDS="1", LEARNING_RATE="0.001", WEIGHT_DECAY="0",
                     NUM_MELS="40", DEVICE="cuda:0").items():
        monkeypatch.setenv(k, v)
    from howl_amd.settings import SETTINGS
    SETTINGS.reset()
    from howl_amd.training.run import pretrain_gsc
    ws = tmp_path / "ws"
    pretrain_gsc.main(["--model", "mobilenet", "--workspace", str(ws), "--synthetic", "256"])
    sd = torch.load(ws / "model-best.pt.bin")
    assert set(sd) == set(omb.mobilenet_init(30))
    assert sd["model.classifier.1.weight"].shape == (30, 1280)
    assert sd["downsample.1.num_batches_tracked"].item() > 0
    import json
    lines = [json.loads(l) for l in (ws / "logs" / "scalars.jsonl").read_text().splitlines()]
    losses = [l["value"] for l in lines if l["tag"] == "Training/Loss"]
    assert len(losses) > 4 and all(v == v for v in losses)
    SETTINGS.reset()


def test_fused_steps_are_bit_repeatable():
    """Every per-channel reduction of the MobileNet kernels ends in "the last block to arrive folds the partial rows in index
    order" (device-scope relaxed atomics, no fence: csrc/mobilenet.hip `Arrive`), and the data- and weight-gradient blocks of a
    layer share a launch: four fused steps, run twice from the same seeds, must leave exactly the same weights -- a stale
    partial row, a race on the shared partial buffers or an arrival counter left non-zero would show up as a difference."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    B, C = 192, 12
    pcm = synthetic_pcm(B, 16000).to(DEV)
    labels = (torch.arange(B) % C).to(DEV)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4]))

    def run():
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)                     # dropout masks
        model, _ = make_mobilenet(C)
        trainer = FusedTrainer(model, std, zmuv, lr=0.001, weight_decay=0.0)
        for _ in range(4):
            trainer.step(pcm, labels)
        torch.cuda.synchronize()
        return trainer.fp.flat.clone()

    first = run()
    assert torch.isfinite(first).all()
    for _ in range(3):
        assert torch.equal(run(), first)


def test_arrival_protocol_stress_across_xcds():
    """Stress of the fence-free "last block folds" hand-off (csrc/mobilenet.hip `arrive`): 40 training-mode forward + backward
    passes over the same 512-utterance batch (every per-channel reduction spans hundreds of blocks on all eight XCDs, two-level
    arrivals included) must reproduce logits, every gradient and the BatchNorm buffers bit for bit -- one stale partial row in
    any of the ~200 folds of a pass would change them."""
    from howl_amd.utils.synth import synthetic_pcm
    from howl_amd.data.transform.transform import StandardAudioTransform
    B, C = 512, 12
    std = StandardAudioTransform().to(DEV).eval()
    x = std(synthetic_pcm(B, 16000).to(DEV))
    labels = (torch.arange(B) % C).to(DEV)
    model, _ = make_mobilenet(C)
    model.train()
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run():
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(3)
        torch.cuda.manual_seed(3)                      # dropout mask
        logits = model(x, None)
        torch.nn.functional.cross_entropy(logits, labels).backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        bufs = torch.cat([b.reshape(-1).float() for b in model.buffers()])
        return logits.detach().clone(), grads.clone(), bufs.clone()

    ref = run()
    assert all(torch.isfinite(t).all() for t in ref)
    for it in range(40):
        out = run()
        for a, b in zip(out, ref):
            assert torch.equal(a, b), f"pass {it}: a reduction folded a stale or missing partial row"


def test_config5_at_full_size_with_device_collate():
    """BASELINE configs[4] at its per-GPU size: 512 x 1 s, 12 labels, the timeshift + white / salt-pepper collate on the
    device (all gates OPEN) feeding FusedTrainer.step.  (a) the first step's training-mode logits at B = 512 agree with the
    oracle's forward on the very batch the device collate produced (<= 1e-3, argmax exact) -- BatchNorm couples the whole
    batch, so this is the full-size comparison, not a slice; (b) the loss is the oracle's; (c) two steps are finite and
    (d) bit-repeatable from the same seeds."""
    from howl_amd.data.collate import DeviceCollate
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedTrainer
    from howl_amd.utils.synth import synthetic_pcm
    from test_gpu_collate import _seed_with
    B, L, C = 512, 16000, 12
    pcm = synthetic_pcm(B, L)
    lens = [L - 31 * (i % 40) for i in range(B)]
    labels = torch.arange(B) % C
    seed = _seed_with(lens, True, True, True)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    keep = (torch.rand(B, omb.LAST_CHANNEL, generator=torch.Generator().manual_seed(3)) >= 0.2).float()

    def run():
        model, sd = make_mobilenet(C)
        model.forced_keep_mask = keep                    # the Dropout(0.2) draw of training mode, fixed for the comparison
        trainer = FusedTrainer(model, std, zmuv, lr=0.001, weight_decay=0.0)      # envs/mobilenet.env
        collate = DeviceCollate(pcm.to(DEV), torch.tensor(lens), labels.to(DEV), max_len=L, seed=seed)
        first = None
        for _ in range(2):
            batch = collate(list(range(B)))
            audio = torch.nn.functional.pad(batch.audio_data, (0, L - batch.audio_data.shape[1]))
            loss = trainer.step(audio, batch.labels)
            if first is None:
                first = (audio.cpu(), batch.labels.cpu(), trainer.last_logits.cpu().clone(), loss.item())
        torch.cuda.synchronize()
        return sd, first, trainer.fp.flat.clone(), loss.item()

    sd, (audio, lab, logits, loss0), flat, loss1 = run()
    assert audio.abs().max().item() <= 1.0 and (audio[:, -1] == 0).any()          # cropped + padded rows are present
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x = z(ofe.standard_audio_transform(audio, fb))
    osd = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ref = omb.mobilenet_forward(osd, x, True, keep)
    assert maxerr(logits, ref) < 1e-3, maxerr(logits, ref)
    assert torch.equal(logits.argmax(1), ref.argmax(1))
    assert abs(loss0 - torch.nn.functional.cross_entropy(ref, lab).item()) < 1e-4
    assert torch.isfinite(flat).all() and loss1 == loss1
    _, (audio2, _, logits2, _), flat2, _ = run()
    assert torch.equal(audio2, audio) and torch.equal(logits2, logits) and torch.equal(flat2, flat)
