"""Training driver over the MI355X engine, with the reference's function surface
(reference train.py:20-290): ``reduce_tensor``, ``init_distributed``, ``prepare_dataloaders``,
``prepare_directories_and_logger``, ``load_model``, ``warm_start_model``, ``load_checkpoint``,
``save_checkpoint``, ``validate``, ``train`` and the same command line
(``-o -l -c --warm_start --n_gpus --rank --group_name --hparams``).  The checkpoint is the
reference's dict ``{iteration, state_dict, optimizer, learning_rate}`` with the reference's
``state_dict`` keys, so checkpoints move between the two code bases in both directions.

What is different:
  * no TensorFlow / Apex: ``fp16_run=True`` selects the engine's bf16 compute mode (bf16 MFMA
    operands, f32 master weights and optimiser: no loss scaling, no overflow skipping);
  * one process per GPU over RCCL; launched by ``tacotron2_amd.multiproc``, by
    ``torch.distributed.run`` (RANK / WORLD_SIZE / LOCAL_RANK in the environment win over the
    ``--rank/--n_gpus`` flags), or by hand with the reference's flags;
  * the loss and gradient norm are read back once per iteration AFTER the whole step has been
    enqueued (the reference's ``loss.item()`` sits between forward and backward, train.py:221,
    and stalls the queue there);
  * gradient exchange: three buckets overlapped with the backward (``distributed.GradSync``).
There is no CPU compute path: without an MI355X ``load_model`` raises.
"""
import argparse
import math
import os
import time

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from . import engine, native
from .data_utils import TextMelCollate, TextMelLoader
from .distributed import apply_gradient_allreduce, reduce_tensor
from .hparams import create_hparams
from .logger import Tacotron2Logger
from .loss_function import Tacotron2Loss
from .model import Tacotron2

MAX_NONFINITE_STEPS = 10        # consecutive iterations with a non-finite gradient norm before the run is declared diverged

__all__ = ['reduce_tensor', 'init_distributed', 'prepare_dataloaders', 'prepare_directories_and_logger',
           'load_model', 'warm_start_model', 'load_checkpoint', 'save_checkpoint', 'validate', 'train']

_CKPT_KEYS = ('iteration', 'state_dict', 'optimizer', 'learning_rate')


def _env_rank_world(n_gpus, rank):
    """torch.distributed.run exports RANK / WORLD_SIZE / LOCAL_RANK; they win over the flags."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        return int(os.environ['WORLD_SIZE']), int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', 0)), True
    return n_gpus, rank, rank, False


def init_distributed(hparams, n_gpus, rank, group_name):
    """Bind this process to its GPU and join the process group (reference train.py:27-39).
    ``hparams.dist_backend`` "nccl" is RCCL on ROCm; ``group_name`` is accepted for signature
    compatibility (torch >= 1.x ignores it)."""
    world, rank, local_rank, from_env = _env_rank_world(n_gpus, rank)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    elif not native.validate_only():                          # CPU tests drive the loop with kernels switched off
        raise AssertionError("Distributed mode requires an MI355X (torch.cuda.is_available() is False).")
    if dist.is_initialized():
        return
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
    if from_env:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=hparams.dist_backend, init_method='env://', world_size=world, rank=rank)
    else:
        dist.init_process_group(backend=hparams.dist_backend, init_method=hparams.dist_url,
                                world_size=world, rank=rank)


def prepare_dataloaders(hparams):
    """``(train_loader, valset, collate_fn)`` (reference train.py:42-60).  Items whose mel is
    computed on the GPU inside the dataset (wav input) and synthetic utterances are produced in-process (a forked
    worker must not touch the GPU, and generated items need none); ``.npy`` mels keep the reference's single worker."""
    trainset = TextMelLoader(hparams.training_files, hparams)
    valset = TextMelLoader(hparams.validation_files, hparams)
    collate_fn = TextMelCollate(hparams.n_frames_per_step)
    in_process = trainset.synthetic is not None or not hparams.load_mel_from_disk    # GPU mel / generated items
    sampler = DistributedSampler(trainset) if hparams.distributed_run else None
    train_loader = DataLoader(trainset, num_workers=0 if in_process else 1, shuffle=sampler is None,
                              sampler=sampler, batch_size=hparams.batch_size, pin_memory=False,
                              drop_last=True, collate_fn=collate_fn)
    return train_loader, valset, collate_fn


def prepare_directories_and_logger(output_directory, log_directory, rank):
    if rank != 0:
        return None
    if not os.path.isdir(output_directory):
        os.makedirs(output_directory)
        os.chmod(output_directory, 0o775)
    return Tacotron2Logger(os.path.join(output_directory, log_directory))


def load_model(hparams):
    """Build the model on the current GPU (reference train.py:73-81).  ``fp16_run`` -> bf16 compute mode.
    ``T2AMD_PRECISION=fp32|bf16|bf16x3`` in the environment picks the engine's compute mode instead (the reference's
    hparams have no word for the f32-class 'bf16x3' mode; the variable keeps the 48-field surface unchanged)."""
    model = Tacotron2(hparams)
    if torch.cuda.is_available():
        model = model.cuda()
    elif not native.validate_only():
        raise native.NativeError("load_model: no MI355X visible and the engine has no CPU path")
    if hparams.fp16_run:
        model.precision = 'bf16'
        model.decoder.attention_layer.score_mask_value = float(torch.finfo(torch.float16).min)
    prec = os.environ.get('T2AMD_PRECISION', '')
    if prec:
        if prec not in ('fp32', 'bf16', 'bf16x3'):
            raise native.NativeError("T2AMD_PRECISION must be fp32, bf16 or bf16x3, got %r" % (prec,))
        model.precision = prec
    if hparams.distributed_run:
        model = apply_gradient_allreduce(model)
    return model


def _read_checkpoint(checkpoint_path):
    if not os.path.isfile(checkpoint_path):
        raise AssertionError("no checkpoint at '%s'" % checkpoint_path)
    try:
        ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
    except TypeError:                                           # torch without the weights_only keyword
        ckpt = torch.load(checkpoint_path, map_location='cpu')
    if 'state_dict' not in ckpt:
        raise KeyError("'%s' is not a Tacotron 2 checkpoint (no 'state_dict')" % checkpoint_path)
    return ckpt


def warm_start_model(checkpoint_path, model, ignore_layers):
    """Weights only; keys in ``ignore_layers`` keep the model's fresh initialisation
    (reference train.py:84-96; default ignore list: ``embedding.weight``, hparams.py:22)."""
    ckpt = _read_checkpoint(checkpoint_path)
    print("Warm starting model from checkpoint '{}'".format(checkpoint_path))
    incoming = ckpt['state_dict']
    if ignore_layers:
        skip = set(ignore_layers)
        merged = model.state_dict()
        merged.update({k: v for k, v in incoming.items() if k not in skip})
        incoming = merged
    model.load_state_dict(incoming)
    return model


def load_checkpoint(checkpoint_path, model, optimizer):
    """-> ``(model, optimizer, learning_rate, iteration)`` (reference train.py:99-109)."""
    ckpt = _read_checkpoint(checkpoint_path)
    print("Loading checkpoint '{}'".format(checkpoint_path))
    model.load_state_dict(ckpt['state_dict'])
    optimizer.load_state_dict(ckpt['optimizer'])
    print("Loaded checkpoint '{}' from iteration {}".format(checkpoint_path, ckpt['iteration']))
    return model, optimizer, ckpt['learning_rate'], ckpt['iteration']


def save_checkpoint(model, optimizer, learning_rate, iteration, filepath):
    """The reference's checkpoint dict (train.py:112-118), written atomically."""
    print("Saving model and optimizer state at iteration {} to {}".format(iteration, filepath))
    payload = dict(zip(_CKPT_KEYS, (iteration, model.state_dict(), optimizer.state_dict(), learning_rate)))
    tmp = filepath + '.tmp'
    torch.save(payload, tmp)
    os.replace(tmp, filepath)


def validate(model, criterion, valset, iteration, batch_size, n_gpus, collate_fn, logger, distributed_run, rank):
    """Mean validation loss over ``valset`` in eval mode (reference train.py:121-146): BatchNorm on
    running statistics, only the prenet dropout active.  One host read at the end, not one per batch."""
    model.eval()
    sampler = DistributedSampler(valset) if distributed_run else None
    loader = DataLoader(valset, sampler=sampler, num_workers=0 if valset.synthetic is not None or
                        not valset.load_mel_from_disk else 1, shuffle=False, batch_size=batch_size,
                        pin_memory=False, collate_fn=collate_fn)
    total, batches, y, y_pred = None, 0, None, None
    with torch.no_grad():
        for batch in loader:
            x, y = model.parse_batch(batch)
            y_pred = model(x)
            loss = criterion(y_pred, y).detach()
            if distributed_run:
                loss = reduce_tensor(loss, n_gpus)
            total = loss if total is None else total + loss
            batches += 1
    model.train()
    val_loss = float(total.item()) / batches if batches else float('nan')
    if rank == 0:
        print("Validation loss {}: {:9f}  ".format(iteration, val_loss))
        if logger is not None:
            logger.log_validation(val_loss, model, y, y_pred, iteration)
    return val_loss


def train(output_directory, log_directory, checkpoint_path, warm_start, n_gpus, rank, group_name, hparams,
          max_iterations=None, fused_optimizer=False):
    """The reference's loop (train.py:149-255): Adam(lr, weight_decay), global-norm clipping, validation
    + checkpoint every ``iters_per_checkpoint`` iterations.  ``max_iterations`` (not in the
    reference) bounds the run for smoke tests and benchmarks.  ``fused_optimizer`` runs clipping + Adam as
    two HIP launches (``optim.FusedAdam``: same state, same checkpoint format) instead of torch's
    ``clip_grad_norm_`` + ``Adam.step``.  Returns the last iteration index."""
    if hparams.distributed_run:
        init_distributed(hparams, n_gpus, rank, group_name)
        n_gpus, rank, _, _ = _env_rank_world(n_gpus, rank)

    torch.manual_seed(hparams.seed)                       # every rank: same seed (train.py:165-166)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(hparams.seed)

    model = load_model(hparams)
    learning_rate = hparams.learning_rate
    if fused_optimizer:
        from .optim import FusedAdam
        optimizer = FusedAdam(model.parameters(), lr=learning_rate, weight_decay=hparams.weight_decay)
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=learning_rate, weight_decay=hparams.weight_decay)
    criterion = Tacotron2Loss()
    logger = prepare_directories_and_logger(output_directory, log_directory, rank)
    train_loader, valset, collate_fn = prepare_dataloaders(hparams)
    if len(train_loader) == 0:
        raise ValueError("training set smaller than one batch of %d (drop_last=True)" % hparams.batch_size)

    iteration, first_epoch = 0, 0
    if checkpoint_path is not None:
        if warm_start:
            model = warm_start_model(checkpoint_path, model, hparams.ignore_layers)
        else:
            model, optimizer, saved_lr, iteration = load_checkpoint(checkpoint_path, model, optimizer)
            if hparams.use_saved_learning_rate:
                learning_rate = saved_lr
            iteration += 1                                # resume at the iteration after the saved one
            first_epoch = max(0, iteration // len(train_loader))

    model.train()
    done = False
    settled = False
    bad_steps = 0
    for epoch in range(first_epoch, hparams.epochs):
        print("Epoch: {}".format(epoch))
        if hparams.distributed_run and isinstance(train_loader.sampler, DistributedSampler):
            train_loader.sampler.set_epoch(epoch)
        for batch in train_loader:
            start = time.perf_counter()
            for group in optimizer.param_groups:
                group['lr'] = learning_rate
            model.zero_grad()
            x, y = model.parse_batch(batch)
            loss = criterion(model(x), y)
            shown = reduce_tensor(loss, n_gpus) if hparams.distributed_run else loss.detach()
            loss.backward()
            if fused_optimizer:
                # a non-finite norm skips the update inside the kernel; the host learns about it below
                grad_norm = optimizer.step(clip_norm=hparams.grad_clip_thresh)
                reduced_loss, grad_norm = (float(v) for v in torch.stack([shown.float(), grad_norm.float()]).tolist())
                finite = math.isfinite(grad_norm)
                if not finite:
                    optimizer.undo_step_count()
            else:
                grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), hparams.grad_clip_thresh)
                # one host read per iteration either way; taken BEFORE the update so that an overflowed step
                # (bf16 compute mode has no loss scaler) is skipped instead of poisoning weights and moments
                reduced_loss, grad_norm = (float(v) for v in torch.stack([shown.float(), grad_norm.float()]).tolist())
                finite = math.isfinite(grad_norm)
                if finite:
                    optimizer.step()
            if not finite:
                bad_steps += 1
                print("Warning: non-finite gradient norm at iteration {} (loss {}): optimiser step skipped "
                      "({} in a row)".format(iteration, reduced_loss, bad_steps), flush=True)
                # was it an abandoned in-launch hand-off (a GPU shared with another job)?  Then say so and go on with the
                # separate-launch forms instead of counting towards divergence
                if not native.validate_only() and engine.handle_nonfinite_step(lambda m: print(m, flush=True)) > 0:
                    bad_steps -= 1
                if bad_steps >= MAX_NONFINITE_STEPS:
                    raise FloatingPointError("%d consecutive iterations with a non-finite gradient norm: the run "
                                             "has diverged (last loss %r)" % (bad_steps, reduced_loss))
            else:
                bad_steps = 0
                engine.note_clean_step()       # what counts towards re-promoting forms a give-up demoted (engine.note_clean_step)
            if finite and rank == 0:
                duration = time.perf_counter() - start
                print("Train loss {} {:.6f} Grad Norm {:.6f} {:.2f}s/it".format(
                    iteration, reduced_loss, grad_norm, duration))
                if logger is not None:
                    logger.log_training(reduced_loss, grad_norm, learning_rate, duration, iteration)
            if finite and iteration % hparams.iters_per_checkpoint == 0:
                validate(model, criterion, valset, iteration, hparams.batch_size, n_gpus, collate_fn, logger,
                         hparams.distributed_run, rank)
                if rank == 0:
                    save_checkpoint(model, optimizer, learning_rate, iteration,
                                    os.path.join(output_directory, "checkpoint_{}".format(iteration)))
            if not settled and not native.validate_only():
                # this loop owns its process: after its first complete iteration everything that exists by now is collected
                # once and frozen out of later garbage collections (engine.settle_gc: a full collection walks ~1.5 M objects
                # once torch is imported, 60-130 ms with the GPU idle behind it); T2AMD_GC_FREEZE_TRAIN=0 leaves the collector alone
                settled = True
                if os.environ.get('T2AMD_GC_FREEZE_TRAIN', '1') != '0':
                    engine.settle_gc()
            iteration += 1
            if max_iterations is not None and iteration >= max_iterations:
                done = True
                break
        if done:
            break
    if logger is not None:
        logger.close()
    return iteration - 1


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('-o', '--output_directory', type=str, help='directory to save checkpoints')
    ap.add_argument('-l', '--log_directory', type=str, help='directory (under -o) for the run log')
    ap.add_argument('-c', '--checkpoint_path', type=str, default=None, help='checkpoint path')
    ap.add_argument('--warm_start', action='store_true', help='load model weights only, ignore specified layers')
    ap.add_argument('--n_gpus', type=int, default=1, help='number of gpus')
    ap.add_argument('--rank', type=int, default=0, help='rank of current gpu')
    ap.add_argument('--group_name', type=str, default='group_name', help='Distributed group name')
    ap.add_argument('--hparams', type=str, help='comma separated name=value pairs')
    ap.add_argument('--max_iterations', type=int, default=None, help='stop after this many iterations')
    ap.add_argument('--fused_optimizer', action='store_true',
                    help='clip + Adam as two HIP launches (tacotron2_amd.optim.FusedAdam) instead of the torch pair')
    args = ap.parse_args(argv)
    hparams = create_hparams(args.hparams)
    print("bf16 compute mode (fp16_run):", hparams.fp16_run)
    print("Distributed Run:", hparams.distributed_run)
    return train(args.output_directory, args.log_directory, args.checkpoint_path, args.warm_start, args.n_gpus,
                 args.rank, args.group_name, hparams, max_iterations=args.max_iterations,
                 fused_optimizer=args.fused_optimizer)


if __name__ == '__main__':
    main()
