"""Trainer entry point: the counterpart of
/root/reference/movenet/pytorch_lightning_trainer.py.

    python -m movenet_amd.pytorch_lightning_trainer --dataset synthetic://clips=8,frames=8000 \
        --use_video 0 --input_channels 64 --residual_channels 16 --skip_channels 16 \
        --layer_size 2 --stack_size 2 --batch_size 2 --n_epochs 1

Same ``Dance2Music`` / ``train_model`` / CLI surface (:24-267 there): same loss
(``cross_entropy`` applied to the model's PROBABILITIES), same
accuracy, same optimizer and scheduler factories and kwargs, schedulers stepped
per optimizer step.  pytorch_lightning is not available offline, so the loop
that ``Trainer.fit`` would run is written out in ``Trainer`` below with the
constructor arguments the reference passes (:233-243).  Multi-GPU is plain data
parallel over clips (movenet_amd/parallel.py) when launched with torchrun.
"""
from __future__ import annotations

import json
import math
import os
import time
from dataclasses import asdict
from pathlib import Path
from typing import Optional

import torch
import torch.nn as nn

from .config import TrainingConfig, arg_parser, config_from_args
from .dataset import get_dataloader
from .optim import FlatAdamW, order_like_backward
from .generation import HostWords
from .utils.host import cap_torch_threads
from .parallel import FlatGradSync, contiguous_grad_span, init_distributed
from .wavenet import WaveNet


class Dance2Music(nn.Module):
    def __init__(self, dataset_fp: str, config: TrainingConfig):
        super().__init__()
        self.learning_rate = config.learning_rate
        self.dataset_fp = dataset_fp
        self.config = config
        self.model = WaveNet(**asdict(config.model_config))
        self.current_epoch = 0
        self.precision = 32
        self.rank, self.world_size = 0, 1
        self.logged_raw = {}  # name -> 0-dim device tensor or float, as logged (no host sync)

    @property
    def device(self) -> torch.device:
        return next(self.model.parameters()).device

    @property
    def logged(self) -> dict:
        """The logged values as floats.  Reading this waits for the GPU; ``log`` does not."""
        return {k: float(v) for k, v in self.logged_raw.items()}

    def log(self, name: str, value, batch_size: Optional[int] = None) -> None:
        self.logged_raw[name] = value.detach() if torch.is_tensor(value) else float(value)

    def forward(self, audio, video, **kwargs):
        return self.model(audio, video, **kwargs)

    def generate(self, audio, video):
        if (self.config.log_samples_every is not None
                and (self.current_epoch + 1) % self.config.log_samples_every == 0):
            return self.model.generate(
                audio, video, n_samples=self.config.generate_n_samples,
                temperature=self.config.generate_temperature).detach()
        return None

    def _shared_step(self, batch, prefix: str):
        audio, video, contexts, fps, info = batch
        dtype = getattr(torch, f"float{self.precision}")
        audio = audio.type(dtype).to(self.device)
        if self.config.use_video:
            video = video.type(dtype).to(self.device)
        # output = self(audio, video) (probabilities, Q1); target = audio[:, :, RF:].argmax(1);
        # loss = cross_entropy(output, target) on those probabilities (Q2); accuracy -- the
        # reference's lines 62-66 as ONE autograd node: softmax + loss + accuracy in one pass
        # over the head's logits, their gradient in one pass back (ops.wavenet_forward_loss)
        # (through both modules' __call__, like the reference's self(audio, video): forward
        # hooks and overrides keep firing)
        loss, acc, output = self(audio, video if self.config.use_video else None, return_loss=True)
        self.log(f"{prefix}_loss", loss, batch_size=self.config.batch_size)
        self.log(f"{prefix}_acc", acc, batch_size=self.config.batch_size)
        return loss, output, audio, video

    def training_step(self, batch, batch_idx):
        loss, output, audio, video = self._shared_step(batch, "train")
        return {"loss": loss, "output": output.detach(),
                "generated_output": self.generate(audio, video)}

    def validation_step(self, batch, batch_idx):
        _, output, audio, video = self._shared_step(batch, "val")
        return {"output": output.detach(), "generated_output": self.generate(audio, video)}

    def _loader(self, train: bool):
        c = self.config
        return get_dataloader(
            self.dataset_fp, input_channels=c.model_config.input_channels,
            batch_size=c.batch_size if train else c.val_batch_size, train=train,
            rank=self.rank, world_size=self.world_size, shuffle=train, pin_memory=c.pin_memory,
            num_workers=c.num_workers if train else c.val_num_workers, use_video=c.use_video,
            batch_subsample_frac=c.batch_subsample_frac if train else c.val_batch_subsample_frac,
            # row F2: class indices cross PCIe (4 bytes per sample), the (B,Q,T) one-hot the Batch
            # contract asks for is formed on the device
            device=self.device if self.device.type == "cuda" else None)

    def train_dataloader(self):
        print("using full audio samples." if self.config.batch_subsample_frac is None
              else f"using {self.config.batch_subsample_frac} of audio samples.")
        return self._loader(True)

    def val_dataloader(self):
        return self._loader(False)

    def configure_optimizers(self):
        c = self.config
        base = {"lr": self.learning_rate, "weight_decay": c.weight_decay}
        opt_kw = {"Adam": base, "AdamW": base, "SGD": {**base, "momentum": c.momentum},
                  "RMSprop": {**base, "momentum": c.momentum}}
        if c.optimizer not in opt_kw:
            raise ValueError(f"optimizer {c.optimizer} not recognized. "
                             f"Must be one of {opt_kw.keys()}")
        if c.optimizer in ("Adam", "AdamW") and self.device.type == "cuda":
            # same update rule as torch.optim.Adam / AdamW, one HIP kernel over one flat buffer
            optimizer = FlatAdamW(order_like_backward(self.model, bool(c.use_video)), decoupled=c.optimizer == "AdamW",
                                  **opt_kw[c.optimizer])
        else:
            optimizer = getattr(torch.optim, c.optimizer)(self.model.parameters(), **opt_kw[c.optimizer])
        print(f"using optimizer: {optimizer}")
        optimizers = {"optimizer": optimizer}
        if c.scheduler is not None:
            n_updates = math.ceil(len(self._loader(True)) / c.accumulation_steps)
            sched_kw = {
                "OneCycleLR": dict(max_lr=c.max_learning_rate, epochs=c.n_epochs,
                                   steps_per_epoch=n_updates, pct_start=c.lr_pct_start,
                                   three_phase=True),
                "CyclicLR": dict(base_lr=c.base_learning_rate, max_lr=c.max_learning_rate,
                                 step_size_up=c.scheduler_step_size_up,
                                 step_size_down=c.scheduler_step_size_down,
                                 mode=c.scheduler_cyclic_mode, gamma=c.scheduler_cyclic_gamma,
                                 cycle_momentum=c.scheduler_cycle_momentum),
                "StepLR": dict(step_size=c.scheduler_step_size, gamma=c.scheduler_step_gamma),
                "MultiStepLR": dict(milestones=c.scheduler_milestones,
                                    gamma=c.scheduler_step_gamma),
            }
            if c.scheduler not in sched_kw:
                raise ValueError(f"scheduler {c.scheduler} not recognized. "
                                 f"Must be one of {sched_kw.keys()}")
            optimizers["lr_scheduler"] = {
                "scheduler": getattr(torch.optim.lr_scheduler, c.scheduler)(
                    optimizer, **sched_kw[c.scheduler]),
                "interval": "step",
            }
            print(f"using scheduler: {optimizers['lr_scheduler']}")
        return optimizers


class _DeferredRecord:
    """One optimizer step's log record whose tensor values are still being computed.  The device
    scalars are gathered into one tensor and published to pinned host memory by a kernel queued
    behind them (generation.HostWords); ``resolve()`` -- called a step later -- polls for it.
    (``float(tensor)`` is a synchronous copy on the current stream: it waits for everything
    enqueued so far, i.e. it would serialise the host's enqueueing of step k + 1 with the GPU's
    execution of it.)"""

    def __init__(self, rec: dict):
        self.rec = rec
        self.keys = [k for k, v in rec.items() if torch.is_tensor(v)]
        self.token, self.host = None, None
        if self.keys:
            vals = torch.stack([rec[k].detach().to(torch.float32).reshape(()) for k in self.keys])
            if vals.is_cuda:
                self.token = HostWords.publish(vals)
            else:
                self.host = vals.tolist()

    def resolve(self) -> dict:
        out = dict(self.rec)
        if self.keys:
            out.update(zip(self.keys, HostWords.read(self.token) if self.token is not None else self.host))
        return out


class Trainer:
    """The part of ``pytorch_lightning.Trainer.fit`` the reference relies on."""

    def __init__(self, max_epochs: int, default_root_dir=None, gradient_clip_val: Optional[float] = 0.0,
                 accumulate_grad_batches: int = 1, logger=None, log_every_n_steps: int = 1,
                 num_sanity_val_steps: int = 0, callbacks=None, track_grad_norm: int = 2,
                 limit_train_batches: Optional[int] = None, device: Optional[str] = None,
                 enable_checkpointing: bool = True, limit_val_batches: Optional[int] = None):
        self.max_epochs = max_epochs
        self.root = Path(default_root_dir) if default_root_dir is not None else None
        self.clip = gradient_clip_val or 0.0
        self.accum = max(1, accumulate_grad_batches)
        self.log_every = max(1, log_every_n_steps)
        self.track_grad_norm = track_grad_norm
        self.limit = limit_train_batches
        # (Lightning's limit_val_batches; default: the training limit, as before)
        self.limit_val = limit_val_batches if limit_val_batches is not None else limit_train_batches
        self.epoch_seconds = []  # wall time of each epoch's training loop
        self.device = device
        self.checkpointing = enable_checkpointing
        self.callbacks = list(callbacks or [])
        self.current_epoch = 0
        self.rank = 0
        self.history = []      # one dict per optimizer step
        self.global_step = 0

    def fit(self, model: Dance2Music) -> None:
        rank, world, local_rank = init_distributed(model.config.dist_backend
                                                   if torch.cuda.is_available() else "gloo",
                                                   model.config.dist_port)
        model.rank, model.world_size = rank, world
        self.rank = rank
        cap_torch_threads()  # the container's CPU share, not the visible cores (utils/host.py)
        dev = torch.device(self.device) if self.device else torch.device("cuda", local_rank)
        if dev.type != "cuda":
            raise RuntimeError("movenet_amd trains on MI355X devices only (no CPU path)")
        model.to(dev)
        opt_cfg = model.configure_optimizers()
        optimizer = opt_cfg["optimizer"]
        scheduler = opt_cfg.get("lr_scheduler", {}).get("scheduler")
        sync = FlatGradSync(model.model.parameters(), world)
        sync.broadcast_parameters(0)
        log_f = None
        if self.root is not None and rank == 0:
            self.root.mkdir(parents=True, exist_ok=True)
            log_f = open(self.root / "metrics.jsonl", "a")
        pending = []  # records whose values are still device tensors

        def flush_records():
            """Resolve the queued records (waiting only for the step that produced them -- which
            is why a record is resolved after the NEXT step has been enqueued), append them to
            history, print / write them."""
            while pending:
                rec = pending.pop(0).resolve()
                self.history.append(rec)
                if rank == 0 and (rec["step"] + 1) % self.log_every == 0:
                    print(json.dumps(rec), flush=True)
                    if log_f:
                        log_f.write(json.dumps(rec) + "\n")
                        log_f.flush()

        for epoch in range(self.max_epochs):
            model.current_epoch = self.current_epoch = epoch
            model.train()
            loader = model.train_dataloader()
            loader.set_epoch(epoch)
            n_batches = len(loader) if self.limit is None else min(len(loader), self.limit)
            optimizer.zero_grad(set_to_none=True)
            t0 = time.perf_counter()
            for batch_idx, batch in enumerate(loader):
                if batch_idx >= n_batches:
                    break
                out = model.training_step(batch, batch_idx)
                (out["loss"] / self.accum).backward()
                for cb in self.callbacks:
                    cb.on_train_batch_end(self, model, out, batch, batch_idx)
                if (batch_idx + 1) % self.accum == 0 or batch_idx + 1 == n_batches:
                    sync.sync_gradients()
                    rec = {"epoch": epoch, "step": self.global_step, **model.logged_raw}
                    used = [p for p in model.model.parameters() if p.grad is not None]
                    # every gradient is a view of ONE flat buffer (ops._run_backward): the total
                    # norm is one reduction over that span (parameters without a gradient are
                    # zero-filled gaps), not one launch per parameter; and nothing here reads a
                    # value back -- the record is resolved a step later (flush_records)
                    # (... only while the gaps really are zero: mvn_backward writes a gradient for EVERY decoder parameter
                    # into the buffer and autograd merely withholds the view of a frozen one -- its slot is a gap with
                    # data in it, which torch's clip_grad_norm_ and the reference do not count.  Then: per parameter.)
                    all_trainable = all(p.requires_grad for p in model.model.parameters())
                    span = (contiguous_grad_span(used)
                            if (self.track_grad_norm or self.clip > 0) and used and all_trainable else None)
                    if self.track_grad_norm and used:
                        if span is not None:
                            rec["grad_norm_total"] = span.norm(self.track_grad_norm)
                        else:
                            rec["grad_norm_total"] = torch.stack(
                                [p.grad.detach().norm(self.track_grad_norm) for p in used]).norm(self.track_grad_norm)
                    if self.clip > 0 and used:
                        if span is not None:  # torch.nn.utils.clip_grad_norm_'s rule on the span
                            total = (rec["grad_norm_total"] if self.track_grad_norm == 2 else span.norm(2))
                            span.mul_((self.clip / (total + 1e-6)).clamp(max=1.0))
                        else:
                            torch.nn.utils.clip_grad_norm_(model.model.parameters(), self.clip)
                    optimizer.step()
                    optimizer.zero_grad(set_to_none=True)
                    rec["lr"] = optimizer.param_groups[0]["lr"]
                    if scheduler is not None:
                        scheduler.step()
                    self.global_step += 1
                    flush_records()          # the PREVIOUS step's values: ready by now
                    pending.append(_DeferredRecord(rec))
            flush_records()
            torch.cuda.synchronize(dev)
            epoch_s = time.perf_counter() - t0
            self.epoch_seconds.append(epoch_s)
            model.eval()
            # epoch means of val_loss / val_acc, weighted by batch size and summed over the
            # ranks' shards: what Lightning's self.log(..., on_epoch=True) reports for
            # validation_step (pytorch_lightning_trainer.py:91-92), not the last batch's values
            val_dev = torch.zeros(3, dtype=torch.float64, device=dev)  # loss * n, acc * n, n
            with torch.no_grad():
                for batch_idx, batch in enumerate(model.val_dataloader()):
                    if self.limit_val is not None and batch_idx >= self.limit_val:
                        break
                    out = model.validation_step(batch, batch_idx)
                    n = float(out["output"].shape[0])
                    val_dev += torch.stack([torch.as_tensor(model.logged_raw["val_loss"], device=dev).double() * n,
                                            torch.as_tensor(model.logged_raw["val_acc"], device=dev).double() * n,
                                            torch.tensor(n, dtype=torch.float64, device=dev)])
                    for cb in self.callbacks:
                        cb.on_validation_batch_end(self, model, out, batch, batch_idx, 0)
            if world > 1:
                import torch.distributed as dist
                t = val_dev if dist.get_backend() == "nccl" else val_dev.cpu()
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                val_dev = t
            val_sum = val_dev.cpu()
            if val_sum[2] > 0:
                model.logged_raw["val_loss"] = float(val_sum[0] / val_sum[2])
                model.logged_raw["val_acc"] = float(val_sum[1] / val_sum[2])
            self.val_epoch_means = {k: v for k, v in model.logged.items() if k.startswith("val")}
            if rank == 0:
                print(json.dumps({"epoch": epoch, "epoch_seconds": epoch_s,
                                  **{k: v for k, v in model.logged.items() if k.startswith("val")}}),
                      flush=True)
                if self.root is not None and self.checkpointing:
                    ck = self.root / "checkpoints"
                    ck.mkdir(parents=True, exist_ok=True)
                    # Lightning's layout: "state_dict" with the LightningModule's "model." prefix
                    torch.save({"epoch": epoch, "global_step": self.global_step,
                                "state_dict": {f"model.{k}": v.detach().cpu()
                                               for k, v in model.model.state_dict().items()}},
                               ck / f"epoch={epoch}-step={self.global_step}.ckpt")
        if log_f:
            log_f.close()


def train_model(dataset: str, config: TrainingConfig, logger_name: Optional[str] = None,
                log_video: bool = False, wandb_project: Optional[str] = None,
                limit_train_batches: Optional[int] = None) -> Trainer:
    model = Dance2Music(dataset, config)
    if logger_name == "wandb":
        raise NotImplementedError("wandb logging is a SaaS integration and out of scope "
                                  "(SURVEY.md section 2); metrics go to <model_output_path>/metrics.jsonl")
    print("Using logger: None")
    callbacks = []
    if config.log_samples_every:
        # the reference attaches this callback only under wandb (:223-229); without wandb the
        # decoded samples go to <model_output_path>/samples/ as .wav files
        from .callbacks import LogSamplesCallback
        callbacks.append(LogSamplesCallback(log_every_n_epochs=config.log_samples_every, log_video=log_video))
    trainer = Trainer(
        max_epochs=config.n_epochs, default_root_dir=config.model_output_path,
        gradient_clip_val=config.gradient_clipping,
        accumulate_grad_batches=config.accumulation_steps, logger=None, log_every_n_steps=1,
        num_sanity_val_steps=0, callbacks=callbacks, track_grad_norm=2,
        limit_train_batches=(limit_train_batches if limit_train_batches is not None
                             else config.n_steps_per_epoch))
    trainer.fit(model=model)
    return trainer


if __name__ == "__main__":
    import logging

    logging.basicConfig(level=logging.INFO,
                        format="%(asctime)s: %(levelname)s: %(name)s: %(message)s")
    args = arg_parser().parse_args()
    train_model(args.dataset, config_from_args(args), logger_name=args.logger,
                log_video=args.log_video, wandb_project=args.wandb_project)
