"""Model / training configuration and CLI flags of the trainer entry point.

Same dataclass fields, defaults and flag names as the reference
(/root/reference/movenet/config.py:11-94 dataclasses, :149-240 argparse), so a
command line written for ``movenet/pytorch_lightning_trainer.py`` parses here
unchanged.  ``dataclasses_json`` is not available offline; ``to_json`` /
``from_json`` are provided directly.  Kept quirk: the
``--gradient_clipping`` flag is parsed but never copied into the config.
"""
from __future__ import annotations

import argparse
import json
from dataclasses import asdict, dataclass, field, fields
from datetime import datetime
from pathlib import Path
from typing import List, Optional


@dataclass
class ModelConfig:
    layer_size: int = 2
    stack_size: int = 2
    input_channels: int = 256
    residual_channels: int = 16
    skip_channels: int = 16


@dataclass
class TrainingConfig:
    model_config: ModelConfig = field(default_factory=ModelConfig)

    batch_size: int = 3
    val_batch_size: int = 3
    checkpoint_every: int = 25
    optimizer: str = "AdamW"
    learning_rate: float = 0.0001
    momentum: float = 0.9
    accumulation_steps: int = 1
    num_workers: int = 0
    val_num_workers: int = 0
    pin_memory: bool = False
    weight_decay: float = 0.0
    n_epochs: int = 100
    n_steps_per_epoch: Optional[int] = None
    use_video: bool = True
    gradient_clipping: Optional[float] = 0.0
    batch_subsample_frac: Optional[float] = None
    val_batch_subsample_frac: Optional[float] = None

    generate_n_samples: Optional[int] = None
    generate_temperature: float = 1.0

    scheduler: Optional[str] = "OneCycleLR"
    lr_pct_start: float = 0.45
    base_learning_rate: float = 0.0003
    scheduler_step_size_up: int = 1000
    scheduler_step_size_down: Optional[int] = None
    scheduler_cyclic_mode: str = "triangular"
    scheduler_cyclic_gamma: float = 1.0
    scheduler_cycle_momentum: bool = False
    max_learning_rate: float = 0.003
    scheduler_step_size: int = 10
    scheduler_step_gamma: float = 0.1
    scheduler_milestones: Optional[List[int]] = None

    dist_backend: Optional[str] = None
    dist_port: str = "8888"

    pretrained_model_path: Optional[Path] = None
    model_output_path: Path = Path("models")
    tensorboard_dir: Path = Path("tensorboard_logs")
    log_samples_every: Optional[int] = None

    def to_dict(self) -> dict:
        d = asdict(self)
        for k, v in d.items():
            if isinstance(v, Path):
                d[k] = str(v)
        return d

    def to_json(self, **kw) -> str:
        return json.dumps(self.to_dict(), **kw)

    @classmethod
    def from_json(cls, text: str) -> "TrainingConfig":
        d = json.loads(text)
        d["model_config"] = ModelConfig(**d["model_config"])
        for k in ("pretrained_model_path", "model_output_path", "tensorboard_dir"):
            if d.get(k) is not None:
                d[k] = Path(d[k])
        known = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in known})


def _flag(x) -> bool:
    return bool(int(x))


def _opt_path(x):
    return None if x is None or x == "" else Path(x)


def arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("--dataset", type=str)
    a("--batch_size", type=int, default=3)
    a("--val_batch_size", type=int, default=3)
    a("--optimizer", type=str, default="AdamW")
    a("--learning_rate", type=float, default=0.001)
    a("--momentum", type=float, default=0.9)
    a("--weight_decay", type=float, default=0.000)
    a("--scheduler", type=str, default=None)
    a("--lr_pct_start", type=float, default=0.45)
    a("--base_learning_rate", type=float, default=0.0003)
    a("--scheduler_step_size_up", type=int, default=1000)
    a("--scheduler_step_size_down", type=int, default=None)
    a("--scheduler_cyclic_mode", type=str, default="triangular")
    a("--scheduler_cyclic_gamma", type=float, default=1.0)
    a("--scheduler_cycle_momentum", type=_flag, default=False)
    a("--max_learning_rate", type=float, default=0.003)
    a("--scheduler_step_size", type=int, default=10)
    a("--scheduler_step_gamma", type=float, default=0.1)
    a("--scheduler_milestones", type=lambda x: [int(i) for i in json.loads(x)], default=None)
    a("--accumulation_steps", type=int, default=1)
    a("--num_workers", type=int, default=1)
    a("--val_num_workers", type=int, default=1)
    a("--pin_memory", type=_flag, default=False)
    a("--generate_n_samples", type=lambda x: x if x is None else int(x), default=None)
    a("--generate_temperature", type=float, default=1.0)
    a("--n_epochs", type=int, default=10)
    a("--n_steps_per_epoch", type=int, default=None)
    a("--use_video", type=_flag, default=True)
    a("--batch_subsample_frac", type=float, default=None)
    a("--val_batch_subsample_frac", type=float, default=None)
    a("--gradient_clipping", type=float, default=0.0)
    a("--checkpoint_every", type=int, default=1)
    a("--input_channels", type=int, default=16)
    a("--residual_channels", type=int, default=16)
    a("--skip_channels", type=int, default=8)
    a("--layer_size", type=int, default=3)
    a("--stack_size", type=int, default=3)
    a("--dist_backend", type=str, default="nccl")
    a("--dist_port", type=str, default="8888")
    a("--pretrained_model_path", type=_opt_path, default=None)
    a("--pretrained_run_exp_name", type=lambda x: None if x is None or x == "" else x, default=None)
    a("--model_output_path", type=Path,
      default=Path("models") / datetime.now().strftime("%Y%m%d%H%M%S"))
    a("--training_logs_path", type=Path, default=Path("training_logs"))
    a("--grid_user_name", type=str, default="")
    a("--grid_api_key", type=str, default="")
    a("--logger", default=None, type=str, choices=["wandb"])
    a("--log_samples_every", type=int, default=None)
    a("--log_video", type=_flag, default=False)
    a("--wandb_api_key", type=str, default="")
    a("--wandb_project", type=str, default="dance2music-pl-testing")
    return p


def config_from_args(args) -> TrainingConfig:
    copied = (
        "batch_size val_batch_size checkpoint_every optimizer learning_rate momentum scheduler "
        "lr_pct_start base_learning_rate scheduler_step_size_up scheduler_step_size_down "
        "scheduler_cyclic_mode scheduler_cyclic_gamma scheduler_cycle_momentum max_learning_rate "
        "scheduler_step_size scheduler_step_gamma scheduler_milestones weight_decay "
        "generate_n_samples generate_temperature accumulation_steps num_workers val_num_workers "
        "pin_memory n_epochs n_steps_per_epoch use_video batch_subsample_frac "
        "val_batch_subsample_frac dist_backend dist_port model_output_path log_samples_every"
    ).split()
    kw = {k: getattr(args, k) for k in copied}  # NB: gradient_clipping is not among them (Q11)
    return TrainingConfig(
        model_config=ModelConfig(
            input_channels=args.input_channels, residual_channels=args.residual_channels,
            skip_channels=args.skip_channels, layer_size=args.layer_size,
            stack_size=args.stack_size),
        pretrained_model_path=(args.pretrained_model_path
                               if args.pretrained_model_path and args.pretrained_run_exp_name
                               else None),
        tensorboard_dir=args.training_logs_path,
        **kw,
    )
