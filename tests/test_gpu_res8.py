"""-m gpu parity: res8 forward / backward / loss / AdamW on the HIP path vs golden vectors (captured from the
reference) and the oracle.  Tolerances: logits <= 1e-3 abs (north_star), argmax exact; in practice ~1e-5."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, golden_features, make_res8, maxerr, t
from oracle import frontend as ofe
from oracle import models as om

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3


@pytest.mark.parametrize("C", [4, 12, 30])
def test_golden_eval_and_train(golden, C):
    g = golden(f"g5_res8_c{C}")
    x, _ = golden_features(golden)
    xd = x.to(DEV)
    model = make_res8(C, train=False)
    with torch.no_grad():
        logits = model(xd, None)
    assert maxerr(logits, g["eval_logits"]) < LOGIT_TOL
    assert torch.equal(logits.argmax(1).cpu(), t(g["eval_logits"]).argmax(1))

    # three optimisation steps through the autograd seam + torch.optim.AdamW, exactly as pretrain_gsc.py:126-133
    model.train()
    labels = t(g["labels"]).to(DEV)
    opt = torch.optim.AdamW(model.parameters(), 0.01, weight_decay=1e-5)
    crit = torch.nn.CrossEntropyLoss()
    names = om.res8_param_names()
    params = dict(model.named_parameters())
    noise = {n: torch.zeros_like(params[n], dtype=torch.float64, device="cpu") for n in names}
    for step in range(3):
        # the state this step starts from, for the oracle's replay of THIS step below
        before = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        moments = [(opt.state[params[n]]["exp_avg"].cpu().clone(), opt.state[params[n]]["exp_avg_sq"].cpu().clone())
                   if step > 0 else (torch.zeros_like(before[n]), torch.zeros_like(before[n])) for n in names]
        scores = model(xd, None)
        opt.zero_grad()
        model.zero_grad()
        loss = crit(scores, labels)
        loss.backward()
        if step == 0:
            assert maxerr(scores, g["train_logits"]) < LOGIT_TOL
            for n, p in model.named_parameters():
                ref = g["grad0." + n]
                assert maxerr(p.grad, ref) < 2e-5 * max(1.0, float(np.abs(ref).max())), n
            for i in (1, 6):
                bn = getattr(model, f"bn{i}")
                assert maxerr(bn.running_mean, g[f"bn{i}.running_mean.1"]) < 1e-5
                assert maxerr(bn.running_var, g[f"bn{i}.running_var.1"]) < 1e-5
        assert abs(loss.item() - float(g[f"loss{step}"])) < 1e-4        # the reference's loss at every step of its trajectory
        opt.step()
        # ONE step of the oracle from the identical state (weights, BatchNorm buffers, Adam moments): no trajectory between the two,
        # so the only slack is what AdamW does to a gradient's rounding noise IN THIS STEP -- a weight moves by ~lr times the
        # relative error of its (bias-corrected) moment, and fp32 gradients of two correct implementations agree to ~4e-6 of the
        # tensor's largest entry (measured 1e-6 .. 6e-6 against the reference's own gradients): tolerance 1e-5 + 3 lr x noise,
        # noise = min(1, 4e-6 max|g| / |g|)
        opt_t = om.AdamWState([before[n] for n in names], 0.01, 1e-5)
        opt_t.m, opt_t.v, opt_t.t = [m for m, _ in moments], [v for _, v in moments], step
        loss_t, _, og = om.train_step(lambda s_, xx: om.res8_forward(s_, xx, True), before, names, opt_t, x, t(g["labels"]))
        assert abs(loss.item() - loss_t.item()) < 2e-5
        for n in names:
            a = og[n].abs().double()
            nz = (4e-6 * a.max() / a.clamp_min(1e-30)).clamp(max=1.0)
            noise[n] = torch.maximum(noise[n], nz)
            dev_o = (params[n].detach().cpu().double() - before[n].double()).abs()
            assert (dev_o - (1e-5 * max(1.0, float(before[n].abs().max())) + 3 * 0.01 * nz)).max().item() < 0, (step, n)
    # Against the REFERENCE's three-step trajectory the weights can only be held to what AdamW's sign-like first steps leave of
    # fp32 rounding: a weight whose gradient is noise moves by +-lr per step, and those few perturb every later gradient.  The
    # oracle's own replay shows how much that is on the testing host: on the host that wrote the goldens it ends 1e-7 from them,
    # with another thread count or CPU up to 2e-3 on ~1 % of the weights (logits of magnitude 40 up to 9e-2 apart).  So each
    # weight gets 1e-4 + 3 lr x noise plus the oracle's own distance -- and that distance is BOUNDED (VERDICT / ADVICE r4): an
    # oracle that drifts beyond what was measured fails the test instead of widening it.
    sd_o = om.res8_init(C)
    opt_o = om.AdamWState([sd_o[n] for n in names], 0.01, 1e-5)
    for step in range(3):
        om.train_step(lambda s_, xx: om.res8_forward(s_, xx, True), sd_o, names, opt_o, x, t(g["labels"]))
    sd = model.state_dict()
    tight = drifted = 0
    for k, v in sd.items():
        ref = t(g["sd3." + k]).double()
        d = (v.detach().cpu().double() - ref).abs()
        tol = 1e-4 * max(1.0, float(ref.abs().max()))
        if "running_" in k:
            # BatchNorm statistics sit downstream of the few O(lr) weight differences: relative agreement
            assert d.max().item() < 1e-3 * max(1.0, float(ref.abs().max())), k
        elif k in noise:
            band = tol + 3 * 0.01 * noise[k]
            odist = (sd_o[k].double() - ref).abs()
            assert odist.max().item() < 5e-3, (k, "oracle trajectory drifted from the golden", odist.max().item())
            drifted += int((odist > band).sum())
            assert (d - (band + odist)).max().item() < 0, k
            tight += int((noise[k] < 1e-2).sum())
    nw = sum(v.numel() for v in noise.values())
    assert tight > 0.95 * nw                                          # tolerances above 4e-4 for a small minority only
    assert drifted < 0.02 * nw, drifted                               # (measured <= 1 % of the weights, on hosts unlike the goldens')
    assert int(sd["bn3.num_batches_tracked"]) == 3
    model.eval()
    with torch.no_grad():
        after = model(xd, None)
    # forward parity at IDENTICAL weights: the oracle evaluated on the weights / BatchNorm buffers the device arrived at
    own = {k: v.detach().cpu().clone() for k, v in sd.items()}
    assert maxerr(after, om.res8_forward(own, x, False)) < LOGIT_TOL
    # against the reference's own trajectory the few noise-gradient weights that moved the other way (see above) show up in
    # logits of magnitude ~40: relative 1e-4 of the largest logit (measured 6e-5 at C = 12), argmax exact
    # (plus what the oracle's own trajectory on this host's CPU ends away from it, for the same reason)
    ref_after = t(g["eval_logits_after3"])
    slack = maxerr(om.res8_forward(sd_o, x, False), ref_after)
    assert slack < 5e-3 * ref_after.abs().max().item(), slack      # bounded: a drifting oracle fails here (measured <= 2.2e-3)
    assert maxerr(after, ref_after) < max(LOGIT_TOL, 1e-4 * ref_after.abs().max().item()) + slack
    assert torch.equal(after.argmax(1).cpu(), ref_after.argmax(1))


def test_golden_half_second_window(golden):
    g, g4 = golden("g5_res8_c4_half"), golden("g4_zmuv")
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    std = StandardAudioTransform().to(DEV).eval()
    z = ZmuvTransform().to(DEV)
    z.mean.copy_(t(g4["mean"]))
    z.mean2.copy_(t(g4["mean2"]))
    x = std.log_mel_for_model(t(g["audio"]).to(DEV), z)      # end to end: PCM -> fused frontend -> res8 (T = 41)
    model = make_res8(4, train=False)
    with torch.no_grad():
        ev = model(x, None)
    assert maxerr(ev, g["eval_logits"]) < LOGIT_TOL and torch.equal(ev.argmax(1).cpu(), t(g["eval_logits"]).argmax(1))
    model.train()
    sc = model(x, None)
    loss = torch.nn.functional.cross_entropy(sc, (torch.arange(6) % 4).to(DEV))
    loss.backward()
    assert maxerr(sc, g["train_logits"]) < LOGIT_TOL and abs(loss.item() - float(g["loss0"])) < 1e-4
    for n in ("conv0.weight", "conv1.weight", "conv6.weight", "output.weight"):
        p = dict(model.named_parameters())[n]
        assert maxerr(p.grad, g["grad0." + n]) < 5e-5, n


@pytest.mark.parametrize("B,L,C", [(64, 16000, 30), (256, 8000, 4), (512, 16000, 12)])
def test_fused_step_vs_oracle_at_baseline_sizes(B, L, C):
    """BASELINE configs 1-3 geometry: one fused training step (PCM in, updated weights out) against the oracle."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedRes8Trainer
    from howl_amd.utils.synth import synthetic_pcm
    pcm = synthetic_pcm(B, L)
    labels = torch.arange(B) % C
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    model = make_res8(C)
    trainer = FusedRes8Trainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
    loss = trainer.step(pcm.to(DEV), labels.to(DEV))
    grads = [g.clone() for g in trainer.fp.grad_views]

    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x = z(ofe.standard_audio_transform(pcm, fb))
    sd = om.res8_init(C)
    names = om.res8_param_names()
    opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)
    ref_loss, ref_logits, ref_grads = om.train_step(lambda s, xx: om.res8_forward(s, xx, True), sd, names, opt, x, labels)
    assert maxerr(trainer.last_logits, ref_logits) < LOGIT_TOL
    assert torch.equal(trainer.last_logits.argmax(1).cpu(), ref_logits.argmax(1))
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    for n, g in zip(names, grads):
        ref = ref_grads[n]
        assert maxerr(g, ref) < 5e-5 * max(1.0, ref.abs().max().item()), n
    for n, p in zip(names, model.hot_parameters()):
        # first AdamW step moves every weight by ~lr * sign(g): compare where the gradient is not rounding noise
        solid = ref_grads[n].abs() > 1e-5
        assert maxerr(p.detach().cpu()[solid], sd[n][solid]) < 2e-4, n
    for i in range(1, 7):
        assert maxerr(getattr(model, f"bn{i}").running_var, sd[f"bn{i}.running_var"]) < 1e-4

    # determinism: the same step from the same state gives bit-identical gradients
    model2 = make_res8(C)
    trainer2 = FusedRes8Trainer(model2, std, zmuv, lr=0.01, weight_decay=1e-5)
    trainer2.step(pcm.to(DEV), labels.to(DEV))
    assert torch.equal(trainer2.fp.grad, trainer.fp.grad) and torch.equal(trainer2.fp.flat, trainer.fp.flat)


def test_loss_decreases_and_errors_are_loud():
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedRes8Trainer
    from howl_amd.utils.synth import synthetic_pcm
    B, C = 64, 12
    pcm = synthetic_pcm(B, 16000).to(DEV)
    labels = (torch.arange(B) % C).to(DEV)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4]))
    model = make_res8(C)
    trainer = FusedRes8Trainer(model, std, zmuv, lr=0.01)
    feat = trainer.features(pcm)
    x_cpu = feat.detach().float().cpu().contiguous()          # (B, 1, 40, T): the oracle trainer steps on the SAME features
    losses = [trainer.step_on_features(feat, labels).item() for _ in range(40)]
    # trajectory pin against the oracle trainer (same features, same closed-form start, same AdamW): the first steps agree to
    # rounding; AdamW's sign-like early updates amplify 1e-7 gradient differences, so the bound widens with the step index
    sd_o, names = om.res8_init(C), om.res8_param_names()
    opt_o = om.AdamWState([sd_o[n] for n in names], 0.01, 0.0)
    ref = [om.train_step(lambda s_, xx: om.res8_forward(s_, xx, True), sd_o, names, opt_o, x_cpu, labels.cpu())[0].item()
           for _ in range(40)]
    for k in range(12):
        assert abs(losses[k] - ref[k]) < 1e-3 * (1 + k) * max(1.0, ref[k]), (k, losses[k], ref[k])
    assert losses[-1] < 0.5 * losses[0], losses                # the oracle ends at ~0.25 x the first loss
    assert abs(losses[-1] - ref[-1]) < 0.15 * losses[0], (losses[-1], ref[-1])
    cpu_model = make_res8(C).cpu()
    with pytest.raises(Exception):
        cpu_model(torch.zeros(2, 1, 40, 81), None)            # no CPU fallback
    with pytest.raises(Exception):
        model(torch.zeros(2, 1, 48, 81, device=DEV), None)    # a mel count the kernels are not built for: error, not garbage


@pytest.mark.parametrize("B,T,C", [(1, 83, 4), (3, 82, 12), (5, 64, 12), (2, 44, 30), (7, 10, 4), (4, 3, 12), (16, 81, 12),
                                   (1, 81, 30), (40, 41, 4)])
def test_odd_geometries_vs_oracle(B, T, C):
    """Frame counts other than the reference's two window lengths (H = T // 3 from 1 to the 27 maximum, frames beyond 3H
    dropped by the pooling, single-utterance batch, position-tile counts that leave some waves without work) and the small
    batches of the reference's presets (16: envs/res8.env; 1: the engines) where several workgroups share an utterance: training
    forward + every gradient and the eval forward against the oracle, through the strided (B,1,M,T) feature view."""
    torch.manual_seed(B * 100 + T)
    feats = torch.randn(B, T, 40) * 1.2                       # (B,T,M) memory, viewed as (B,1,M,T) like the frontend's output
    x = feats.permute(0, 2, 1).unsqueeze(1)
    labels = torch.arange(B) % C
    model = make_res8(C)
    logits = model(x.to(DEV), None)
    torch.nn.functional.cross_entropy(logits, labels.to(DEV)).backward()
    sd = om.res8_init(C)
    names = om.res8_param_names()
    params = [sd[n].requires_grad_(True) for n in names]
    ref = om.res8_forward(sd, x.contiguous(), True)
    ref_grads = torch.autograd.grad(torch.nn.functional.cross_entropy(ref, labels), params)
    assert maxerr(logits, ref) < LOGIT_TOL
    assert torch.equal(logits.argmax(1).cpu(), ref.argmax(1))
    for n, p, g in zip(names, model.hot_parameters(), ref_grads):
        assert maxerr(p.grad, g) < 1e-4 * max(1.0, g.abs().max().item()), n
    model.eval()
    with torch.no_grad():
        ev = model(x.to(DEV), None)
    sd_eval = {k: v.detach() for k, v in sd.items()}
    assert maxerr(ev, om.res8_forward(sd_eval, x.contiguous(), False)) < LOGIT_TOL
    with torch.no_grad():                                     # T > 83 in eval mode: the windowed path (tested below)
        assert model(torch.randn(2, 1, 40, 84, device=DEV), None).shape == (2, C)


@pytest.mark.parametrize("B", [1, 16, 64, 96, 512])
def test_merged_dgrad_wgrad_launch_is_bit_identical_to_separate_launches(monkeypatch, B):
    """howl_res8_bwd runs dgrad and wgrad of a layer side by side in ONE launch (half the CUs each).  Repeating the same step
    must give bit-identical gradients, and the two halves launched one after the other with the same grids
    (HOWL_RES8_BWD_PAIR=0) must give exactly the same bits: the merged launch changes scheduling, not arithmetic.  Batches of
    1, 16 and 64 run the sliced geometries (4 + 2, 4 + 2 and 2 + 2 workgroups per utterance)."""
    T, C = 81, 12
    torch.manual_seed(7)
    x = (torch.randn(B, T, 40) * 1.2).permute(0, 2, 1).unsqueeze(1).to(DEV)
    labels = (torch.arange(B) % C).to(DEV)

    def grads():
        model = make_res8(C)
        torch.nn.functional.cross_entropy(model(x, None), labels).backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in model.hot_parameters()]

    first = grads()
    for _ in range(4):
        again = grads()
        for a, b in zip(first, again):
            assert torch.equal(a, b)
    monkeypatch.setenv("HOWL_RES8_BWD_PAIR", "0")
    separate = grads()
    for a, b in zip(first, separate):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,C", [(1, 4), (64, 30), (512, 12)])
def test_loss_inside_the_forward_launch_is_bit_identical(B, C):
    """howl_res8_fwd_xent / howl_res8_bwd_xent (cross-entropy in the forward's last launch, batch mean in the backward's head
    launch) against howl_res8_fwd + howl_xent_fwd_bwd + howl_res8_bwd on the device: same logits, loss, dlogits and gradients,
    bit for bit, in one call and in the two-part form of the data-parallel step."""
    from howl_amd import ops
    T = 81
    torch.manual_seed(11)
    x = (torch.randn(B, T, 40) * 1.1).permute(0, 2, 1).unsqueeze(1).to(DEV)
    labels = (torch.arange(B) * 7 % C).to(DEV)

    def run(fused, parts):
        model = make_res8(C).train()
        if fused:
            logits, nll, dlogits = model._launch_forward_xent(x, labels)
            loss = torch.empty(1, device=DEV)
            kw = dict(xent=(nll, loss))
        else:
            logits = model._launch_forward(x)
            loss, dlogits = ops.xent(logits, labels)
            kw = {}
        grads = None
        for part in parts:
            grads = model._launch_backward(x, dlogits, out_grads=grads, part=part, **kw)
        torch.cuda.synchronize()
        return logits, loss, dlogits, grads, model.bn3.running_var.clone()

    ref = run(False, (0,))
    for parts in ((0,), (1, 2)):
        got = run(True, parts)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
        for a, b in zip(got[3], ref[3]):
            assert torch.equal(a, b)
        assert torch.equal(got[4], ref[4])


@pytest.mark.parametrize("B,T", [(3, 84), (2, 120), (4, 201), (1, 500)])
def test_long_inputs_in_eval_mode_vs_oracle(B, T):
    """Res8 accepts any T in the reference (cnn.py:127-145).  Beyond the 83 frames that fit the on-chip map the HIP path runs row
    strips with exchanged halo rows (the module, eval and training) or overlapping 27-row windows (howl_res8_fwd_long, eval only):
    logits of both vs the oracle over the whole clip."""
    C = 12
    sd = om.res8_init(C)
    gen = torch.Generator().manual_seed(T)
    for i in range(1, 7):
        sd[f"bn{i}.running_mean"] = torch.rand(45, generator=gen) * 0.5 + 0.1
        sd[f"bn{i}.running_var"] = torch.rand(45, generator=gen) * 1.5 + 0.5
    x = torch.randn(B, 3, 40, T, generator=gen)
    ref = om.res8_forward({k: v.clone() for k, v in sd.items()}, x, False)
    model = make_res8(C, state=sd, train=False)
    with torch.no_grad():
        got = model(x.to(DEV), None)
    assert maxerr(got, ref) < LOGIT_TOL
    assert torch.equal(got.argmax(1).cpu(), ref.argmax(1))
    # (the module runs long clips as row strips since round 5; the windowed forward is still an entry point of the library)
    x0, sb, st, sm = model._feat_view(x.to(DEV))
    assert maxerr(model._launch_forward_long(x0, sb, st, sm), ref) < LOGIT_TOL
    # training mode takes the same clip as row strips with exchanged halo rows (test_res8_trains_beyond_83_frames_vs_oracle)
    model.train()
    tr = model(x.to(DEV), None)
    ref_tr = om.res8_forward({k: v.clone() for k, v in sd.items()}, x, True)
    assert maxerr(tr, ref_tr) < LOGIT_TOL


@pytest.mark.parametrize("B,T", [(24, 41), (16, 81), (64, 81), (1, 81)])
def test_sliced_small_batch_kernels_match_one_workgroup_per_utterance(monkeypatch, B, T):
    """Small batches run several workgroups per utterance (csrc/res8.hip conv_slices / pair_slices: position tiles of the
    forward and the data gradient, N tiles of the weight gradient).  Against the same step with slicing switched off
    (HOWL_RES8_SLICES=0) only fp32 summation orders differ (BatchNorm partial rows, nothing else): logits, BatchNorm buffers
    and every gradient agree to rounding, far inside the oracle tolerance."""
    C = 12
    torch.manual_seed(B + T)
    x = (torch.randn(B, T, 40) * 1.2).permute(0, 2, 1).unsqueeze(1).to(DEV)
    labels = (torch.arange(B) % C).to(DEV)

    def run():
        model = make_res8(C)
        logits = model(x, None)
        torch.nn.functional.cross_entropy(logits, labels).backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), [p.grad.clone() for p in model.hot_parameters()], model.bn3.running_var.clone()

    sliced = run()
    monkeypatch.setenv("HOWL_RES8_SLICES", "0")
    plain = run()
    assert maxerr(sliced[0], plain[0]) < 2e-5
    assert maxerr(sliced[2], plain[2]) < 1e-6
    # (BatchNorm statistics that differ in their last bit flip the ReLU mask of the handful of activations that sit within
    # 1e-6 of zero -- ~2 per million: isolated gradient entries then move by up to a few 1e-3 of the tensor's largest entry, as
    # observed at 50 x 41 frames of Gaussian features; the geometries here agree to 2e-6.  A single utterance has gradients of
    # ~1e-8 behind its own BatchNorm: absolute floor)
    for a, b in zip(sliced[1], plain[1]):
        assert maxerr(a, b) < 5e-3 * b.abs().max().item() + 1e-6


def test_reference_written_workspace_on_the_gpu(golden, tmp_path):
    """f4: ``howl_amd.workspace.Workspace.load_model`` on the workspace the REFERENCE wrote (tests/golden/ref_workspace, see
    make_golden.py --only-checkpoint) -> frontend + ZMUV + res8 eval forward on the HIP path reproduce the logits the
    reference computed from the same files (``hubconf.py:53-84`` is the consumer this stands in for)."""
    import shutil
    from pathlib import Path
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.model import RegisteredModel
    from howl_amd.workspace import Workspace
    g = golden("g12_ref_workspace")
    src = Path(__file__).resolve().parent / "golden" / "ref_workspace"
    ws = Workspace(tmp_path / "ws", delete_existing=False)
    for name in ("model-best.pt.bin", "zmuv.pt.bin"):
        shutil.copyfile(src / name, ws.path / name)
    model = RegisteredModel.find_registered_class("res8")(int(g["num_labels"]))
    ws.load_model(model, best=True)
    model = model.to(DEV).eval()
    zmuv = ZmuvTransform()
    zmuv.load_state_dict(torch.load(str(ws.path / "zmuv.pt.bin"), map_location="cpu"))
    zmuv = zmuv.to(DEV)
    std = StandardAudioTransform().to(DEV).eval()
    audio = t(g["audio"]).to(DEV)
    with torch.no_grad():
        by_protocol = model(zmuv(std(audio)), None)                      # the reference's call chain (3-channel features)
        fused = model(std.log_mel_for_model(audio, zmuv), None)          # the fused fast path of the training step
    for logits in (by_protocol, fused):
        assert maxerr(logits, g["eval_logits"]) < LOGIT_TOL
        assert torch.equal(logits.argmax(1).cpu(), t(g["eval_logits"]).argmax(1))


@pytest.mark.parametrize("B,T", [(512, 81), (96, 41), (5, 62)])
def test_fused_backward_matches_the_unfused_reference(monkeypatch, B, T):
    """Round 4: BatchNorm backward + skip gradient + ReLU mask are applied inside the tile staging of both roles of the backward
    pair (three tensors per element, staged under the K loop in half-tile phases), the weight-gradient partials are folded by
    the next pair, ds_2 joins dx_0 in layer 1's data gradient.  HOWL_RES8_BWD_FUSED=0 runs the elementwise pass as its own
    launch (bn_relu_bwd_kernel writes dz): both must give the same gradients to rounding, repeatably."""
    C = 12
    torch.manual_seed(11)
    x = (torch.randn(B, T, 40) * 1.1).permute(0, 2, 1).unsqueeze(1).to(DEV)
    labels = (torch.arange(B) % C).to(DEV)

    def grads():
        model = make_res8(C)
        torch.nn.functional.cross_entropy(model(x, None), labels).backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in model.hot_parameters()]

    fused = grads()
    for a, b in zip(fused, grads()):
        assert torch.equal(a, b)                              # the staging phases race with nothing
    monkeypatch.setenv("HOWL_RES8_BWD_FUSED", "0")
    unfused = grads()
    for a, b in zip(fused, unfused):
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() < 5e-6 * scale


@pytest.mark.parametrize("B,L,C", [(64, 16000, 30), (512, 16000, 12)])
def test_optimiser_step_in_the_fold_is_bit_identical_on_the_device(B, L, C, monkeypatch):
    """Round 5: on a single replica the AdamW step rides in the backward's last fold launch (HowlAdamW).  Three steps that way
    and three with the optimiser's own launch (HOWL_NO_FOLD_ADAMW=1) from the same state: parameters, both moments and the
    gradients of the last step are bit-identical -- a block that read a gradient or a moment before its writer was done would
    show here (B = 64: all seven weight-gradient rows fold in that launch; B = 512: two, the rest rides as the extra row)."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedRes8Trainer
    from howl_amd.utils.synth import synthetic_pcm
    pcm = synthetic_pcm(B, L).to(DEV)
    labels = (torch.arange(B) % C).to(DEV)
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4]))
    out = []
    for apart in (False, True):
        if apart:
            monkeypatch.setenv("HOWL_NO_FOLD_ADAMW", "1")
        model = make_res8(C)
        tr = FusedRes8Trainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
        for _ in range(3):
            tr.step(pcm, labels)
        torch.cuda.synchronize()
        assert model.optimizer_step_done          # (with the switch the library still takes the step: as its own launch)
        out.append([t.clone() for t in (tr.fp.flat, tr.m, tr.v, tr.fp.grad)])
    for a, b in zip(*out):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,T,C", [(512, 81, 12), (64, 81, 30), (3, 82, 12), (1, 83, 4), (16, 41, 4), (7, 10, 4), (129, 40, 12)])
def test_res8_at_80_mel_bins_vs_oracle(B, T, C):
    """NUM_MELS = 80, the reference's stock default (settings.py:32; cnn.py:113-145 pools (3,4) over whatever width it is given):
    20 pooled columns run as two strips of 10 per utterance that fetch each other's edge column in the forward, the data
    gradient and the weight gradient (csrc/res8.hip, HaloSlot).  Training forward, every gradient, the BatchNorm buffers and the
    eval forward against the oracle, from a full 512-utterance batch (two strips per workgroup) down to one utterance; odd
    batches; bit-repeatable."""
    from gpu_util import res8_oracle_with_kernel_relus
    names = om.res8_param_names()
    labels = torch.arange(B) % C
    torch.manual_seed(B * 100 + T)
    feats = torch.randn(B, T, 80) * 1.2
    x = feats.permute(0, 2, 1).unsqueeze(1)
    model = make_res8(C)
    logits = model(x.to(DEV), None)
    torch.nn.functional.cross_entropy(logits, labels.to(DEV)).backward()
    # One ReLU whose input lies within fp32 rounding of zero comes out on the other side in another summation order and moves a
    # gradient by O(1 / positions) -- percent level at three utterances, for the oracle against its own fp64 run as much as for
    # the kernels (measured: 3 of 6 random batches at 3 x 82 frames, 40 or 80 bins alike).  So the comparison shares the
    # decisions: the kernels' saved ReLU patterns may differ from the oracle's only where the oracle's pre-activation is within
    # rounding of zero (asserted), and with those patterns the oracle's gradients must match tightly at every batch size.
    ref, ref_grads, flips, sd = res8_oracle_with_kernel_relus(model, x, labels, B, T, 80, C)
    assert maxerr(logits, ref) < LOGIT_TOL
    for n, p in zip(names, model.hot_parameters()):
        g = ref_grads[n]
        assert maxerr(p.grad, g) < 2e-5 * max(1.0, g.abs().max().item()), (n, flips)
    for i in (1, 4, 6):
        assert maxerr(getattr(model, f"bn{i}").running_mean, sd[f"bn{i}.running_mean"]) < 1e-5
        assert maxerr(getattr(model, f"bn{i}").running_var, sd[f"bn{i}.running_var"]) < 1e-4
    if B >= 512:     # ... and against the oracle's own decisions, no masks involved (a flip moves a 512-utterance gradient by ~1e-6;
                     # at 64 utterances one flip still moved conv4's gradient by 7e-5 of 7e-3 on the device)
        from gpu_util import res8_plain_oracle_grads
        plain, plain_grads = res8_plain_oracle_grads(x, labels, C)
        assert maxerr(logits, plain) < LOGIT_TOL
        for n, p in zip(names, model.hot_parameters()):
            assert maxerr(p.grad, plain_grads[n]) < 5e-5 * max(1.0, plain_grads[n].abs().max().item()), n
    # the same step from the same state: the same bits
    model2 = make_res8(C)
    logits2 = model2(x.to(DEV), None)
    torch.nn.functional.cross_entropy(logits2, labels.to(DEV)).backward()
    assert torch.equal(logits2, logits)
    for p, q in zip(model.hot_parameters(), model2.hot_parameters()):
        assert torch.equal(p.grad, q.grad)
    model.eval()
    with torch.no_grad():
        ev = model(x.to(DEV), None)
    sd_eval = {k: v.detach() for k, v in sd.items()}
    assert maxerr(ev, om.res8_forward(sd_eval, x.contiguous(), False)) < LOGIT_TOL


def test_res8_at_80_mel_bins_long_input_and_fused_step(monkeypatch):
    """80 mel bins end to end: the frontend (two filterbank banks of 40), the fused trainer step (PCM in, updated weights out:
    loss inside the forward, AdamW inside the last fold) against the oracle; and clips beyond 83 frames in eval mode (windows x
    strips)."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.settings import SETTINGS
    from howl_amd.training.fused import FusedRes8Trainer
    from howl_amd.utils.synth import synthetic_pcm
    monkeypatch.setattr(SETTINGS.audio_transform, "num_mels", 80)
    B, L, C = 96, 16000, 12
    pcm = synthetic_pcm(B, L)
    labels = torch.arange(B) % C
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    model = make_res8(C)
    trainer = FusedRes8Trainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
    loss = trainer.step(pcm.to(DEV), labels.to(DEV))
    grads = [g.clone() for g in trainer.fp.grad_views]
    fb = ofe.mel_fb(80)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x = z(ofe.standard_audio_transform(pcm, fb))
    assert x.shape[2] == 80
    sd = om.res8_init(C)
    names = om.res8_param_names()
    opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)
    ref_loss, ref_logits, ref_grads = om.train_step(lambda s_, xx: om.res8_forward(s_, xx, True), sd, names, opt, x, labels)
    assert maxerr(trainer.last_logits, ref_logits) < LOGIT_TOL
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    for n, g in zip(names, grads):
        assert maxerr(g, ref_grads[n]) < 5e-5 * max(1.0, ref_grads[n].abs().max().item()), n
    for n, p in zip(names, model.hot_parameters()):
        solid = ref_grads[n].abs() > 1e-5
        assert maxerr(p.detach().cpu()[solid], sd[n][solid]) < 2e-4, n
    # long clips, eval mode
    model.eval()
    xl = torch.randn(2, 1, 80, 201)
    with torch.no_grad():
        ev = model(xl.to(DEV), None)
    sd_eval = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = om.res8_forward(sd_eval, xl, False)
    assert maxerr(ev, ref) < 1e-4 * max(1.0, ref.abs().max().item())


def test_golden_stock_80_mel_bins(golden, monkeypatch):
    """G13: outputs of the REFERENCE's classes at its stock NUM_MELS = 80 (settings.py:32) for the six GSC clips -- the frontend
    module (eval and the recorded VTLP draw), then res8 on the reference's own features: eval logits, one training step's logits /
    loss / every gradient / BatchNorm buffers, and the logits after the AdamW step."""
    import random
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.settings import SETTINGS
    from gpu_util import t
    g = golden("g13_res8_80mel")
    monkeypatch.setattr(SETTINGS.audio_transform, "num_mels", 80)
    std = StandardAudioTransform().to(DEV).eval()
    audio = t(g["audio"]).to(DEV)
    feats = std(audio)
    ref = t(g["feats"])
    d = (feats[:, 0].cpu() - ref[:, 0]).abs()
    assert d.max().item() < 2e-3 and d[ref[:, 0] > -8].max().item() < 2e-4
    std.train()
    random.seed(11)
    out = std(audio, mels_only=True)
    assert abs(std.last_vtlp_alpha - float(g["vtlp_alpha"])) == 0.0
    d = (out.cpu() - t(g["mels_vtlp"])).abs()
    assert d.max().item() < 2e-3 and d[t(g["mels_vtlp"]) > -8].max().item() < 2e-4
    z = ofe.Zmuv()
    z.mean, z.mean2 = t(g["zmuv_mean"]), t(g["zmuv_mean2"])
    x = z(ref).to(DEV)
    C = 12
    model = make_res8(C, train=False)
    with torch.no_grad():
        assert maxerr(model(x, None), g["eval_logits"]) < LOGIT_TOL
    model.train()
    opt = torch.optim.AdamW(model.parameters(), 0.01, weight_decay=1e-5)
    sc = model(x, None)
    loss = torch.nn.functional.cross_entropy(sc, t(g["labels"]).to(DEV))
    loss.backward()
    assert maxerr(sc, g["train_logits"]) < LOGIT_TOL and abs(loss.item() - float(g["loss0"])) < 1e-4
    for n, p in model.named_parameters():
        assert maxerr(p.grad, g["grad0." + n]) < 5e-5 * max(1.0, float(np.abs(g["grad0." + n]).max())), n
    for i in (1, 6):
        assert maxerr(getattr(model, f"bn{i}").running_mean, g[f"bn{i}.running_mean.1"]) < 1e-5
        assert maxerr(getattr(model, f"bn{i}").running_var, g[f"bn{i}.running_var.1"]) < 1e-4
    opt.step()
    model.eval()
    with torch.no_grad():
        assert maxerr(model(x, None), g["eval_logits_after1"]) < 2e-3


@pytest.mark.parametrize("B,T,C,M", [(64, 101, 12, 40), (512, 121, 12, 40), (8, 161, 4, 80), (3, 250, 12, 40), (33, 84, 30, 80)])
def test_res8_trains_beyond_83_frames_vs_oracle(B, T, C, M):
    """cnn.py:127-145 takes any T (MAX_WINDOW_SIZE_SECONDS beyond 1.03 s).  More than 27 pooled rows do not fit the kernels' tile:
    the utterance runs as row strips of equal height that fetch real halo rows (and corners, at 80 mel bins) from their
    neighbours; a last strip that owns fewer rows than its block keeps the rest at zero in every tile and out of every sum
    (csrc/res8.hip StripGeom).  1.25 s, 1.5 s at the full batch, 2 s at 80 bins (53 rows: 27 + 26), 3.1 s in four strips, 84 frames
    (28 rows: 14 + 14): training forward, every gradient with the kernels' own ReLU decisions, BatchNorm buffers, bit-repeatable."""
    from gpu_util import res8_oracle_with_kernel_relus
    names = om.res8_param_names()
    labels = torch.arange(B) % C
    torch.manual_seed(B * 100 + T)
    x = (torch.randn(B, T, M) * 1.2).permute(0, 2, 1).unsqueeze(1)
    model = make_res8(C)
    logits = model(x.to(DEV), None)
    torch.nn.functional.cross_entropy(logits, labels.to(DEV)).backward()
    ref, ref_grads, flips, sd = res8_oracle_with_kernel_relus(model, x, labels, B, T, M, C)
    assert maxerr(logits, ref) < LOGIT_TOL
    for n, p in zip(names, model.hot_parameters()):
        g = ref_grads[n]
        assert maxerr(p.grad, g) < 2e-5 * max(1.0, g.abs().max().item()), (n, flips)
    for i in (1, 4, 6):
        assert maxerr(getattr(model, f"bn{i}").running_mean, sd[f"bn{i}.running_mean"]) < 1e-5
        assert maxerr(getattr(model, f"bn{i}").running_var, sd[f"bn{i}.running_var"]) < 1e-4
    if B >= 512:     # ... and against the oracle's own decisions, no masks involved
        from gpu_util import res8_plain_oracle_grads
        plain, plain_grads = res8_plain_oracle_grads(x, labels, C)
        assert maxerr(logits, plain) < LOGIT_TOL
        for n, p in zip(names, model.hot_parameters()):
            assert maxerr(p.grad, plain_grads[n]) < 5e-5 * max(1.0, plain_grads[n].abs().max().item()), n
    model2 = make_res8(C)
    logits2 = model2(x.to(DEV), None)
    torch.nn.functional.cross_entropy(logits2, labels.to(DEV)).backward()
    assert torch.equal(logits2, logits)
    for p, q in zip(model.hot_parameters(), model2.hot_parameters()):
        assert torch.equal(p.grad, q.grad)
    # eval mode on the same clip goes through the windowed path: same function
    model.eval()
    with torch.no_grad():
        ev = model(x.to(DEV), None)
    ref_ev = om.res8_forward({k: v.detach() for k, v in sd.items()}, x.contiguous(), False)
    assert maxerr(ev, ref_ev) < 1e-4 * max(1.0, ref_ev.abs().max().item())


def test_fused_trainer_on_two_second_windows():
    """The fused step (loss inside the forward's last launch, AdamW inside the last fold) at MAX_WINDOW_SIZE_SECONDS=2: 161 frames,
    two row strips (27 + 26 rows) per utterance, against the oracle's step."""
    from howl_amd.data.transform.operator import ZmuvTransform
    from howl_amd.data.transform.transform import StandardAudioTransform
    from howl_amd.training.fused import FusedRes8Trainer
    from howl_amd.utils.synth import synthetic_pcm
    B, L, C = 96, 32000, 12
    pcm = synthetic_pcm(B, L)
    labels = torch.arange(B) % C
    std = StandardAudioTransform().to(DEV).eval()
    zmuv = ZmuvTransform().to(DEV)
    zmuv.update(std(pcm[:4].to(DEV)))
    model = make_res8(C)
    trainer = FusedRes8Trainer(model, std, zmuv, lr=0.01, weight_decay=1e-5)
    loss = trainer.step(pcm.to(DEV), labels.to(DEV))
    grads = [g.clone() for g in trainer.fp.grad_views]
    fb = ofe.mel_fb(40)
    z = ofe.Zmuv()
    z.update(ofe.standard_audio_transform(pcm[:4], fb))
    x = z(ofe.standard_audio_transform(pcm, fb))
    assert x.shape[-1] == 161
    sd = om.res8_init(C)
    names = om.res8_param_names()
    opt = om.AdamWState([sd[n] for n in names], 0.01, 1e-5)
    ref_loss, ref_logits, ref_grads = om.train_step(lambda s_, xx: om.res8_forward(s_, xx, True), sd, names, opt, x, labels)
    assert maxerr(trainer.last_logits, ref_logits) < LOGIT_TOL
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    for n, g in zip(names, grads):
        assert maxerr(g, ref_grads[n]) < 5e-5 * max(1.0, ref_grads[n].abs().max().item()), n
    for n, p in zip(names, model.hot_parameters()):
        solid = ref_grads[n].abs() > 1e-5
        assert maxerr(p.detach().cpu()[solid], sd[n][solid]) < 2e-4, n
    losses = [trainer.step(pcm.to(DEV), labels.to(DEV)).item() for _ in range(12)]
    assert losses[-1] < loss.item()


@pytest.mark.parametrize("mels", [40, 80])
def test_golden_two_second_windows(golden, mels):
    """G14: outputs of the REFERENCE's res8 on 161-frame inputs (MAX_WINDOW_SIZE_SECONDS=2; pairs of the GSC clips) at 40 and 80
    mel bins: eval logits (module: row strips; and the windowed howl_res8_fwd_long), one training step's logits, loss, every
    gradient and BatchNorm buffers.  Three utterances: a flipped ReLU decision would show at the percent level, so the gradient
    bound is checked with the reference's own values only where the batch is well conditioned -- it is (fixed inputs)."""
    g = golden("g14_res8_two_second_windows")
    pre = f"m{mels}."
    x = t(g[pre + "x"]).to(DEV)
    model = make_res8(12, train=False)
    with torch.no_grad():
        assert maxerr(model(x, None), g[pre + "eval_logits"]) < LOGIT_TOL
        x0, sb, st, sm = model._feat_view(x)
        assert maxerr(model._launch_forward_long(x0, sb, st, sm), g[pre + "eval_logits"]) < LOGIT_TOL
    model.train()
    sc = model(x, None)
    loss = torch.nn.functional.cross_entropy(sc, (torch.arange(3) % 12).to(DEV))
    loss.backward()
    assert maxerr(sc, g[pre + "train_logits"]) < LOGIT_TOL and abs(loss.item() - float(g[pre + "loss0"])) < 1e-4
    for n, p in model.named_parameters():
        ref = g[pre + "grad0." + n]
        assert maxerr(p.grad, ref) < 5e-5 * max(1.0, float(np.abs(ref).max())), n
    for i in (1, 6):
        assert maxerr(getattr(model, f"bn{i}").running_mean, g[pre + f"bn{i}.running_mean.1"]) < 1e-5
        assert maxerr(getattr(model, f"bn{i}").running_var, g[pre + f"bn{i}.running_var.1"]) < 1e-4


def test_eval_mode_takes_any_length_and_small_buffers():
    """cnn.py:127-145 takes any T.  Eval mode runs on three rotating activation buffers and the forward part of the workspace
    (a long clip must not allocate a training step's 14 tensors) and the row-strip count is no longer capped at 64
    (5,184 frames = 65 s: ADVICE r5; now 1024 strips): 64, 67 and 149 strips and a short multi-strip batch against the oracle."""
    from howl_amd import lib
    model = make_res8(12, train=False)
    L = lib.get().cdll
    assert L.howl_res8_eval_workspace_bytes_mels(4, 2000, 40) < L.howl_res8_workspace_bytes_mels(4, 2000, 40) // 4
    sd = {k: v.detach() for k, v in om.res8_init(12).items()}
    torch.manual_seed(3)
    for B, T in ((2, 5184), (1, 5400), (1, 12001), (3, 700)):
        x = (torch.randn(B, T, 40) * 1.2).permute(0, 2, 1).unsqueeze(1)
        with torch.no_grad():
            ev = model(x.to(DEV), None)
        ref = om.res8_forward(sd, x.contiguous(), False)
        assert maxerr(ev, ref) < 1e-4 * max(1.0, ref.abs().max().item()), (B, T)
    assert L.howl_res8_row_strips(5184) == 64 and L.howl_res8_row_strips(5400) == 67
    # nothing large stays cached between calls
    held = sum(b.ws.numel() + 12 * b.rot[0].numel() for b in model._eval_cache.values())
    assert held <= model.EVAL_CACHE_BYTES and not model._buffers_cache
