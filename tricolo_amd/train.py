"""Minimal fit loop for TriCoLoNet when PyTorch-Lightning is not installed (it is not on the build / GPU boxes).

The reference drives its step with Lightning (`/root/reference/train.py:26-52`): Trainer.fit + ModelCheckpoint on
`val_eval/RR@5` + LrDecayCallback (`tricolo/callback/lr_decay_callback.py:6-17`).  Orchestration is outside the hot path
this package accelerates, so this is deliberately small: it runs the SAME module hooks (`training_step`,
`validation_step`, `on_validation_epoch_end`, `configure_optimizers`) in the same order, applies the same cosine decay,
keeps the best checkpoint by the same monitor, and resumes from a checkpoint including the optimizer state.  With
Lightning installed, use the reference's own train.py with the `_target_` overrides of INTEGRATION.md instead.

    python -m tricolo_amd.train data=synthetic model.text_encoder=BiGRUEncoder model.voxel_encoder=SparseCNNEncoder \
        experiment_name=demo trainer.max_epochs=4
"""
from __future__ import annotations

import os
import sys
from math import cos, pi

import torch

from . import config as tcfg
from . import ops
from .checkpoint import extract_state_dict, save_reference_checkpoint


def cosine_lr(cfg, epoch: int) -> float | None:
    """lr_decay_callback.py:6-17, evaluated at the END of `epoch`: None before lr_decay.start_epoch."""
    start, end, clip = cfg.lr_decay.start_epoch, cfg.trainer.max_epochs, 1e-6
    if epoch < start:
        return None
    return clip + 0.5 * (cfg.optimizer.lr - clip) * (1 + cos(pi * ((epoch - start) / (end - start))))


def fit(net, cfg, train_batches, val_batches=None, device="cuda", ckpt_path: str | None = None, out_dir: str | None = None,
        log=print):
    """train_batches / val_batches: callables epoch -> iterable of batch dicts (data_module.py:40-65 layout).
    Returns {'epoch', 'global_step', 'best': (monitor value, path) | None, 'history': [...]}."""
    net = net.to(device)
    opt = net.configure_optimizers()
    start_epoch, step = 0, 0
    if ckpt_path is not None:                                         # train.py:41-45 -> trainer.fit(ckpt_path=...)
        assert os.path.exists(ckpt_path), "Error: Checkpoint path does not exists."
        doc = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        net.load_state_dict({k: v for k, v in extract_state_dict(doc).items() if ".clip_model." not in k}, strict=True)
        if doc.get("optimizer_states"):
            opt.load_state_dict(doc["optimizer_states"][0])
        start_epoch, step = int(doc.get("epoch", -1)) + 1, int(doc.get("global_step", 0))
    best, history = None, []
    every = int(cfg.trainer.get("check_val_every_n_epoch", 1) or 1)
    epoch_start_step = step                                           # global step the first epoch of THIS run starts from (resume: the checkpoint's)
    for epoch in range(start_epoch, int(cfg.trainer.max_epochs)):
        net.train()
        last = None
        for batch in train_batches(epoch):
            opt.zero_grad(set_to_none=True)
            loss = net.training_step(batch, step)
            loss.backward(gradient=ops.one(loss.device))
            opt.step()
            last, step = loss, step + 1
        rec = {"epoch": epoch, "global_step": step, "train_loss": float(last.item()) if last is not None else None,
               "lr": float(opt.param_groups[0]["lr"])}
        if hasattr(opt, "skipped_steps"):                             # FusedAdam's overflow guard: one device read per epoch
            rec["skipped_steps"], rec["skipped_elements"] = opt.skipped_steps(), opt.nonfinite_skipped()
            seen = (history[-1].get("skipped_steps", 0) + history[-1].get("skipped_elements", 0)) if history else 0
            if rec["skipped_steps"] + rec["skipped_elements"] > seen:
                log(f"[tricolo_amd.train] WARNING: {rec['skipped_steps']} optimizer step(s) / {rec['skipped_elements']} gradient element(s) skipped so far "
                    "because of inf / NaN gradients (f16 activation-gradient overflow?): lower ops.F16_GRAD_SCALE or train in bf16x3")
            # (a resumed run starts with an empty history: its first epoch counts from the checkpoint's global step - FusedAdam.load_state_dict
            #  zeroes the skip record -, and one skipped step of a one-step epoch is a transient, not a frozen run: ADVICE r5)
            prev_sk, prev_step = (history[-1].get("skipped_steps", 0), history[-1]["global_step"]) if history else (0, epoch_start_step)
            if step - prev_step >= 2 and rec["skipped_steps"] - prev_sk >= step - prev_step:
                # the static scale does not back off like GradScaler: a persistent overflow would freeze the weights silently
                raise RuntimeError(f"every optimizer step of epoch {epoch} was skipped for non-finite gradients: lower TRICOLO_F16_GRAD_SCALE "
                                   "(a power of two) or train in bf16x3")
        lr = cosine_lr(cfg, epoch)
        if lr is not None:
            for g in opt.param_groups:
                g["lr"] = lr
            if hasattr(opt, "sync_lr"):
                opt.sync_lr()                                         # device-side scalar of FusedAdam (graph replays follow it)
        if val_batches is not None and (epoch + 1) % every == 0:
            net.eval()
            with torch.no_grad():
                for i, batch in enumerate(val_batches(epoch)):
                    net.validation_step(batch, i)
                m = net.on_validation_epoch_end()
            rec["val_eval/RR@1"], rec["val_eval/RR@5"] = float(m["recall_rate"][0] * 100), float(m["recall_rate"][4] * 100)
            if out_dir is not None and (best is None or rec["val_eval/RR@5"] > best[0]):      # ModelCheckpoint(monitor=RR@5, mode=max)
                os.makedirs(out_dir, exist_ok=True)
                path = os.path.join(out_dir, f"epoch={epoch}-RR5={rec['val_eval/RR@5']:.2f}.ckpt")
                save_reference_checkpoint(net, path, epoch=epoch, global_step=step, optimizer=opt)
                if best is not None and os.path.exists(best[1]):
                    os.remove(best[1])
                best = (rec["val_eval/RR@5"], path)
        if out_dir is not None:
            os.makedirs(out_dir, exist_ok=True)
            save_reference_checkpoint(net, os.path.join(out_dir, "last.ckpt"), epoch=epoch, global_step=step, optimizer=opt)
        history.append(rec)
        log(rec)
    return {"epoch": int(cfg.trainer.max_epochs) - 1, "global_step": step, "best": best, "history": history}


def main(argv=None):
    from .data import synthetic as syn
    from .model.tricolo_net import TriCoLoNet
    cfg = tcfg.compose(overrides=list(argv if argv is not None else sys.argv[1:]))
    if cfg.data.get("dataset") != "Synthetic":
        raise SystemExit("tricolo_amd.train only ships the synthetic data source (no dataset can be downloaded here); with real "
                         "data use the reference's train.py + DataModule and the `_target_` overrides of INTEGRATION.md")
    torch.manual_seed(cfg.train_seed)
    net = TriCoLoNet(cfg)
    V = cfg.data.voxel_size if cfg.model.voxel_encoder else None
    nv = cfg.data.num_views if cfg.model.image_encoder else None
    B = min(int(cfg.data.get("batch_size", 32)), 64)
    train = syn.make_factor_retrieval_set(1024, 2, V or 32, nv, cfg.data.image_size, seed=syn.BASE_SEED + 301, distinct=False)
    held = syn.make_factor_retrieval_set(256, 2, V or 32, nv, cfg.data.image_size, seed=syn.BASE_SEED + 302, distinct=True)

    def batches(items, shuffle_seed=None):
        def gen(epoch):
            import numpy as np
            order = np.arange(len(items)) if shuffle_seed is None else np.random.default_rng(shuffle_seed + epoch).permutation(len(items))
            for i in range(0, len(order) - B + 1, B):
                yield syn.batch_to_device(syn.collate_items([items[j] for j in order[i:i + B]], voxel=V is not None, views=nv is not None), "cuda")
        return gen
    out = os.path.join(cfg.experiment_output_path, "training") if cfg.get("experiment_name") else None
    ckpt = os.path.join(out, cfg.ckpt_name) if (out and cfg.get("ckpt_name")) else None
    return fit(net, cfg, batches(train, cfg.train_seed), batches(held), ckpt_path=ckpt, out_dir=out)


if __name__ == "__main__":
    main()
