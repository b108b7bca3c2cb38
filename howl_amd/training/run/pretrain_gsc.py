"""``python -m training.run.pretrain_gsc --model res8 --workspace W [--load-weights] [--eval]`` on MI355X.

Same flow, settings (env vars) and workspace artefacts as ``training/run/pretrain_gsc.py:22-144`` of the reference:
ZMUV pass (<= 2001 single-clip updates), per-epoch training with LR decay, dev/test accuracy, model(-best) checkpoints.
Differences that do not change the arithmetic of the hot path: clips live on the device (``ClipBank``) instead of
DataLoader workers, the training step is the fused C-ABI pipeline, and the loss is logged without a per-step host sync.
``--synthetic N`` trains on N generated clips when no dataset is mounted (as on the GPU box).

Data parallel (BASELINE configs[2]: batch 4096 over 8 GPUs): start one process per GPU with
``python -m torch.distributed.run --nproc-per-node N -m training.run.pretrain_gsc ...``.  ``BATCH_SIZE`` is then the GLOBAL
batch: every rank draws the same shuffled batches and takes its contiguous share (``howl_amd.parallel.shard``), rank 0's
initial weights / BatchNorm buffers / ZMUV statistics are broadcast, the flat gradient is summed over RCCL inside the fused
step, evaluation batches are dealt round-robin and their counts summed, and only rank 0 writes the workspace.
"""
import argparse
import logging
from pathlib import Path

import torch

from howl_amd import parallel
from howl_amd.data.collate import DeviceCollate
from howl_amd.data.transform.operator import ZmuvTransform
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.model import RegisteredModel
from howl_amd.model.cnn import require_supported_mels
from howl_amd.settings import SETTINGS
from howl_amd.training.data import ClipBank, load_gsc_splits, read_wav16k, synthetic_bank
from howl_amd.training.fused import FusedTrainer
from howl_amd.utils.random_utils import set_random_seed
from howl_amd.workspace import Workspace


def train_epoch(trainer, collate, id_batches, std_transform, writer, epoch_idx, needs_lengths=False, prefetch=0):
    """The loop body of ``training/run/pretrain_gsc.py:120-133`` over one epoch's batches of clip ids: collate (truncate,
    Timeshift, Noise, batchify on the device) -> frontend in train mode (VTLP draw) -> fused training step -> loss logging
    without a host synchronisation.  ``prefetch`` > 0: the host half of the collate (draws, sort, packed staging buffer) runs
    that many batches ahead in a worker thread (``DeviceCollate.prefetch``), as the reference's DataLoader workers do -- measured
    SLOWER than preparing inline once the draws are array operations (round 5, 1xMI355X: 1.190 against 1.168 ms per step at
    512 utterances, 0.391 against 0.334 at 64: 25-47 us of host work per batch do not pay for the hand-over between two Python
    threads), hence off by default.
    (Also measured and removed, round 5: the NEXT batch's upload and collate kernel on a side stream beside the current step --
    bit-identical results, but 1.185 against 1.167 ms per step at 512 utterances and 0.354 against 0.317 at 64: the event record /
    wait pairs between two queues cost more than the 20-60 us of device work they would hide, as every cross-queue experiment on
    this stack has.)
    Returns the number of utterances trained on.  (``bench.py --loop entry`` times exactly this function.)"""
    batches = collate.prefetch(id_batches, depth=prefetch) if prefetch else (collate(ids) for ids in id_batches)
    seen = 0
    for batch in batches:
        if needs_lengths:
            loss = trainer.step(batch.audio_data, batch.labels, std_transform.compute_lengths(batch.lengths),
                                int(std_transform.compute_lengths(torch.tensor(batch.audio_data.shape[-1]))))      # (the batch's padded length)
        else:
            loss = trainer.step(batch.audio_data, batch.labels)
        writer.add_scalar("Training/Loss", loss.detach(), epoch_idx)     # stays on the device until flush
        seen += batch.audio_data.shape[0]
    return seen


COLLATE_SEED_SALT = 0x5DEECE66D     # collate stream = Random(seed ^ salt + 1000003 rank): never the global Random(seed)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", type=str, choices=RegisteredModel.registered_names(), default="res8")
    ap.add_argument("--workspace", type=str, default=str(Path("workspaces") / "default"))
    ap.add_argument("--load-weights", action="store_true")
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--synthetic", type=int, default=0, help="train on this many generated clips (no dataset needed)")
    args = ap.parse_args(argv)

    device = torch.device(SETTINGS.training.device)
    rank, world, device = parallel.init_from_env(device)
    main_rank = rank == 0
    zmuv_on_disk = (Path(args.workspace) / "zmuv.pt.bin").exists()      # asked before rank 0 writes anything
    if main_rank:
        workspace = Workspace(Path(args.workspace), delete_existing=not args.eval)
    parallel.barrier()
    if not main_rank:
        workspace = Workspace(Path(args.workspace), delete_existing=False, writable=False)
    writer = workspace.summary_writer
    set_random_seed(SETTINGS.training.seed)
    sample_rate = SETTINGS.audio.sample_rate
    max_len = int(SETTINGS.training.max_window_size_seconds * sample_rate)
    num_labels = 30   # pretrain_gsc.py:91 hard-codes the head size

    if args.synthetic:
        train = synthetic_bank(args.synthetic, max_len, num_labels, device, seed=1)
        dev = synthetic_bank(max(args.synthetic // 8, 16), max_len, num_labels, device, seed=2)
        test = synthetic_bank(max(args.synthetic // 8, 16), max_len, num_labels, device, seed=3)
    else:
        splits = load_gsc_splits(Path(SETTINGS.dataset.dataset_path), SETTINGS.training.vocab)
        train, dev, test = (ClipBank([read_wav16k(p) for p, _ in s], [l for _, l in s], max_len, device) for s in splits)

    std_transform = StandardAudioTransform().to(device).eval()
    zmuv_transform = ZmuvTransform().to(device)
    model = RegisteredModel.find_registered_class(args.model)(num_labels).to(device)
    require_supported_mels(model)      # res8 with NUM_MELS other than 40 / 80: an error here, not at the first batch
    params = [p for p in model.parameters() if p.requires_grad]
    logging.info(f"{sum(p.numel() for p in params)} parameters")

    zmuv_path = workspace.path / "zmuv.pt.bin"
    if zmuv_on_disk:
        zmuv_transform.load_state_dict(torch.load(str(zmuv_path), map_location="cpu"))
    else:
        if main_rank:
            gen = torch.Generator().manual_seed(SETTINGS.training.seed)
            perm = torch.randperm(len(train), generator=gen)[:2001]        # prep_dl: batch size 1, shuffled, <= 2001 clips
            for i in perm.tolist():
                n = int(train.lengths_host[i])
                zmuv_transform.update(std_transform(train.audio[i:i + 1, :n]))
        parallel.broadcast_([zmuv_transform.total, zmuv_transform.mean, zmuv_transform.mean2])
        if main_rank:   # only freshly computed statistics are written: the other ranks may still be reading an existing file
            torch.save({k: v.cpu() for k, v in zmuv_transform.state_dict().items()}, str(zmuv_path))

    def evaluate_accuracy(bank, prefix, epoch_idx=None, save=False):
        std_transform.eval()
        model.eval()
        counts = torch.zeros(2, device=device)            # [correct, total]; batches dealt round-robin to the ranks
        with torch.no_grad():
            for k, ids in enumerate(bank.index_batches(SETTINGS.training.batch_size, shuffle=False, drop_last=False)):
                if k % world != rank:
                    continue
                batch = bank.batch(torch.tensor(ids))
                scores = model(std_transform.log_mel_for_model(batch.audio_data, zmuv_transform),
                               std_transform.compute_lengths(batch.lengths))
                counts[0] += (scores.max(1)[1] == batch.labels).float().sum()
                counts[1] += scores.size(0)
        parallel.allreduce_scalars_(counts)
        correct, total = counts.tolist()
        acc = correct / max(total, 1)
        if save and not args.eval:
            writer.add_scalar(f"{prefix}/Metric/acc", acc, epoch_idx)
            workspace.increment_model(model, acc / 10)
        return acc

    if args.load_weights or args.eval:
        workspace.load_model(model, best=True)
        model.to(device)
    if args.eval:
        dev_acc, test_acc = evaluate_accuracy(dev, "Dev"), evaluate_accuracy(test, "Test")
        if main_rank:
            print("dev_acc: ", dev_acc)
            print("test_acc: ", test_acc)
        return

    workspace.write_args(args)
    workspace.save_settings(SETTINGS)
    writer.add_scalar("Meta/Parameters", sum(p.numel() for p in params))
    if not hasattr(model, "hot_parameters"):
        raise NotImplementedError(f"{args.model}: no fused MI355X training step (res8, mobilenet, lstm have one)")
    trainer = FusedTrainer(model, std_transform, zmuv_transform, SETTINGS.training.learning_rate,
                           weight_decay=SETTINGS.training.weight_decay)
    trainer.broadcast_parameters()                                     # rank 0's initial weights and BatchNorm buffers
    needs_lengths = getattr(model, "NEEDS_LENGTHS", False)
    # train_comp = compose(truncate, Timeshift.train(), Noise.train(), batchify) (pretrain_gsc.py:78-80), on the device.  The
    # collate owns its `random` stream, as each of the reference's DataLoader workers does (a worker thread prepares the next
    # batches while the main thread -- whose global `random` VTLP draws from -- launches the current one).  Its seed is derived so
    # that it cannot coincide with the global stream's (set_random_seed(seed) seeds `random` with the same integer: with
    # Random(seed) here, VTLP's gate and alpha would replay values the Timeshift / Noise draws had consumed); the single-process
    # order of draws therefore differs from the reference's one shared stream ("Random streams")
    train_collate = DeviceCollate(train.audio, train.lengths_host, train.labels, max_len, sr=sample_rate,
                                  seed=SETTINGS.training.seed ^ COLLATE_SEED_SALT, replica=rank)
    dev_acc = 0

    def shards(id_batches):
        for ids in id_batches:
            ids = parallel.shard(ids)
            if not len(ids):
                raise RuntimeError(f"BATCH_SIZE={SETTINGS.training.batch_size} leaves rank {rank} of {world} without utterances")
            yield ids

    for epoch_idx in range(SETTINGS.training.num_epochs):
        model.train()
        std_transform.train()
        gen = torch.Generator().manual_seed(SETTINGS.training.seed + 7919 * (epoch_idx + 1))   # the same order on every rank
        train_epoch(trainer, train_collate,
                    shards(train.index_batches(SETTINGS.training.batch_size, shuffle=True, drop_last=True, generator=gen)),
                    std_transform, writer, epoch_idx, needs_lengths)
        trainer.decay_lr(SETTINGS.training.lr_decay)
        dev_acc = evaluate_accuracy(dev, "Dev", epoch_idx, save=True)
    test_acc = evaluate_accuracy(test, "Test")
    writer.close()
    parallel.barrier()
    if main_rank:
        print("model: ", args.model)
        print("dev_acc: ", dev_acc)
        print("test_acc: ", test_acc)


if __name__ == "__main__":
    main()
