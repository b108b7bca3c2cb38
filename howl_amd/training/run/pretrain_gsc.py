"""``python -m training.run.pretrain_gsc --model res8 --workspace W [--load-weights] [--eval]`` on MI355X.

Same flow, settings (env vars) and workspace artefacts as ``training/run/pretrain_gsc.py:22-144`` of the reference:
ZMUV pass (<= 2001 single-clip updates), per-epoch training with LR decay, dev/test accuracy, model(-best) checkpoints.
Differences that do not change the arithmetic of the hot path: clips live on the device (``ClipBank``) instead of
DataLoader workers, the res8 step is the fused C-ABI pipeline, and the loss is logged without a per-step host sync.
``--synthetic N`` trains on N generated clips when no dataset is mounted (as on the GPU box).
"""
import argparse
import logging
from pathlib import Path

import torch

from howl_amd.data.collate import DeviceCollate
from howl_amd.data.transform.operator import ZmuvTransform
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.model import RegisteredModel
from howl_amd.settings import SETTINGS
from howl_amd.training.data import ClipBank, load_gsc_splits, read_wav16k, synthetic_bank
from howl_amd.training.fused import FusedRes8Trainer
from howl_amd.utils.random_utils import set_random_seed
from howl_amd.workspace import Workspace


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", type=str, choices=RegisteredModel.registered_names(), default="res8")
    ap.add_argument("--workspace", type=str, default=str(Path("workspaces") / "default"))
    ap.add_argument("--load-weights", action="store_true")
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--synthetic", type=int, default=0, help="train on this many generated clips (no dataset needed)")
    args = ap.parse_args(argv)

    workspace = Workspace(Path(args.workspace), delete_existing=not args.eval)
    writer = workspace.summary_writer
    set_random_seed(SETTINGS.training.seed)
    device = torch.device(SETTINGS.training.device)
    sample_rate = SETTINGS.audio.sample_rate
    max_len = int(SETTINGS.training.max_window_size_seconds * sample_rate)
    num_labels = 30   # pretrain_gsc.py:91 hard-codes the head size

    if args.synthetic:
        train = synthetic_bank(args.synthetic, max_len, num_labels, device, seed=1)
        dev = synthetic_bank(max(args.synthetic // 8, 64), max_len, num_labels, device, seed=2)
        test = synthetic_bank(max(args.synthetic // 8, 64), max_len, num_labels, device, seed=3)
    else:
        splits = load_gsc_splits(Path(SETTINGS.dataset.dataset_path), SETTINGS.training.vocab)
        train, dev, test = (ClipBank([read_wav16k(p) for p, _ in s], [l for _, l in s], max_len, device) for s in splits)

    std_transform = StandardAudioTransform().to(device).eval()
    zmuv_transform = ZmuvTransform().to(device)
    model = RegisteredModel.find_registered_class(args.model)(num_labels).to(device)
    params = [p for p in model.parameters() if p.requires_grad]
    logging.info(f"{sum(p.numel() for p in params)} parameters")

    zmuv_path = workspace.path / "zmuv.pt.bin"
    if zmuv_path.exists():
        zmuv_transform.load_state_dict(torch.load(str(zmuv_path), map_location="cpu"))
    else:
        perm = torch.randperm(len(train))[:2001]                       # prep_dl: batch size 1, shuffled, <= 2001 clips
        for i in perm.tolist():
            n = int(train.lengths[i])
            zmuv_transform.update(std_transform(train.audio[i:i + 1, :n]))
    torch.save({k: v.cpu() for k, v in zmuv_transform.state_dict().items()}, str(zmuv_path))

    def evaluate_accuracy(bank, prefix, epoch_idx=None, save=False):
        std_transform.eval()
        model.eval()
        num_corr = torch.zeros((), device=device)
        num_tot = 0
        with torch.no_grad():
            for batch in bank.batches(SETTINGS.training.batch_size, shuffle=False, drop_last=False):
                scores = model(std_transform.log_mel_for_model(batch.audio_data, zmuv_transform),
                               std_transform.compute_lengths(batch.lengths))
                num_tot += scores.size(0)
                num_corr += (scores.max(1)[1] == batch.labels).float().sum()
        acc = num_corr.item() / max(num_tot, 1)
        if save and not args.eval:
            writer.add_scalar(f"{prefix}/Metric/acc", acc, epoch_idx)
            workspace.increment_model(model, acc / 10)
        return acc

    if args.load_weights or args.eval:
        workspace.load_model(model, best=True)
        model.to(device)
    if args.eval:
        print("dev_acc: ", evaluate_accuracy(dev, "Dev"))
        print("test_acc: ", evaluate_accuracy(test, "Test"))
        return

    workspace.write_args(args)
    workspace.save_settings(SETTINGS)
    writer.add_scalar("Meta/Parameters", sum(p.numel() for p in params))
    fused = args.model in ("res8", "mobilenet")
    if fused:
        trainer = FusedRes8Trainer(model, std_transform, zmuv_transform, SETTINGS.training.learning_rate,
                                   weight_decay=SETTINGS.training.weight_decay)
    else:
        optimizer = torch.optim.AdamW(params, SETTINGS.training.learning_rate, weight_decay=SETTINGS.training.weight_decay)
        criterion = torch.nn.CrossEntropyLoss()
    # train_comp = compose(truncate, Timeshift.train(), Noise.train(), batchify) (pretrain_gsc.py:78-80), on the device
    train_collate = DeviceCollate(train.audio, train.lengths, train.labels, max_len, sr=sample_rate)
    dev_acc = 0
    for epoch_idx in range(SETTINGS.training.num_epochs):
        model.train()
        std_transform.train()
        for ids in train.index_batches(SETTINGS.training.batch_size, shuffle=True, drop_last=True):
            batch = train_collate(ids)
            if fused:
                loss = trainer.step(batch.audio_data, batch.labels)
            else:
                scores = model(std_transform.log_mel_for_model(batch.audio_data, zmuv_transform),
                               std_transform.compute_lengths(batch.lengths))
                optimizer.zero_grad()
                loss = criterion(scores, batch.labels)
                loss.backward()
                optimizer.step()
            writer.add_scalar("Training/Loss", loss.detach(), epoch_idx)     # stays on the device until flush
        if fused:
            trainer.decay_lr(SETTINGS.training.lr_decay)
        else:
            for group in optimizer.param_groups:
                group["lr"] *= SETTINGS.training.lr_decay
        dev_acc = evaluate_accuracy(dev, "Dev", epoch_idx, save=True)
    test_acc = evaluate_accuracy(test, "Test")
    writer.close()
    print("model: ", args.model)
    print("dev_acc: ", dev_acc)
    print("test_acc: ", test_acc)


if __name__ == "__main__":
    main()
