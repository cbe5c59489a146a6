"""Sample logging without wandb (row F4 of SURVEY.md section 8).

Mirrors ``movenet.callbacks.LogSamplesCallback`` (/root/reference/movenet/callbacks.py:24-134):
same constructor, same hooks and the same epoch gate; every ``log_every_n_epochs`` epochs
the model's one-step predictions (``outputs["output"]``) and its free-running generation
(``outputs["generated_output"]``) are turned back into waveforms -- argmax over the class
axis, then mu-law decoding (:66-76) -- and logged.  The reference uploads a wandb table
with the clips resampled to the source video's rate (librosa / torchvision, both absent
offline); here the decoded waveforms are written as 16-bit PCM ``.wav`` files at the model
rate (MAX_AUDIO_FRAMES / 10 = 16 kHz) under ``<default_root_dir>/samples/<split>/`` with
one JSON line per clip in ``samples/index.jsonl`` (the table's columns).  The decoding runs
on the GPU (``mvn_mu_law_decode``: formula of RESEARCH.md:156-163, parity unpinned).
"""
from __future__ import annotations

import json
import wave
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .ops import mu_law_decode
from .wavenet import MAX_AUDIO_FRAMES

COLUMNS = ["split", "epoch", "batch_idx", "fp", "origin_audio", "pred_audio", "gen_audio"]
SAMPLE_RATE = MAX_AUDIO_FRAMES // 10  # 16,000 (callbacks.py:88)


def write_wav(path: Path, waveform: np.ndarray, sample_rate: int = SAMPLE_RATE) -> None:
    """waveform in [-1, 1] -> mono 16-bit PCM."""
    pcm = (np.clip(np.asarray(waveform, dtype=np.float64), -1.0, 1.0) * 32767.0).round().astype("<i2")
    path.parent.mkdir(parents=True, exist_ok=True)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(pcm.tobytes())


class LogSamplesCallback:
    def __init__(self, log_every_n_epochs: int = 10, log_video: bool = True, temperature: float = 1.0,
                 out_dir: Optional[str] = None):
        self.log_every_n_epochs = log_every_n_epochs
        self.log_video = log_video  # kept for signature parity: there is no video to re-attach
        self.temperature = temperature
        self.out_dir = Path(out_dir) if out_dir is not None else None
        self.columns = list(COLUMNS)

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx):
        if (trainer.current_epoch + 1) % self.log_every_n_epochs != 0:
            return
        self.log_samples("train", trainer, pl_module, outputs, batch, batch_idx)

    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        if (trainer.current_epoch + 1) % self.log_every_n_epochs != 0:
            return
        self.log_samples("validation", trainer, pl_module, outputs, batch, batch_idx)

    @staticmethod
    def _decode(one_hot_like: torch.Tensor, classes: int) -> np.ndarray:
        """(B, Q, S) probabilities / one-hot -> (B, S) waveforms in [-1, 1]."""
        return mu_law_decode(one_hot_like.argmax(1).to(torch.int32), classes).cpu().numpy()

    def log_samples(self, split, trainer, pl_module, outputs, batch, batch_idx):
        if getattr(trainer, "rank", 0) != 0:
            return
        root = self.out_dir or (Path(trainer.root) / "samples" if trainer.root is not None else None)
        if root is None:
            return
        audio, _, _, fps, _infos = batch
        Q = pl_module.config.model_config.input_channels
        origin = self._decode(audio.to(outputs["output"].device), Q)
        pred = self._decode(outputs["output"], Q)
        gen = None
        if outputs.get("generated_output", None) is not None:
            gen = self._decode(outputs["generated_output"], Q)
        rows = []
        for i, fp in enumerate(fps):
            stem = f"epoch={trainer.current_epoch}-batch={batch_idx}-clip={i}"
            files = {"origin_audio": root / split / f"{stem}-origin.wav",
                     "pred_audio": root / split / f"{stem}-pred.wav"}
            write_wav(files["origin_audio"], origin[i])
            write_wav(files["pred_audio"], pred[i])
            if gen is not None:
                files["gen_audio"] = root / split / f"{stem}-gen.wav"
                write_wav(files["gen_audio"], gen[i])
            rows.append({"split": split, "epoch": trainer.current_epoch, "batch_idx": batch_idx, "fp": fp,
                         **{k: str(v.relative_to(root)) for k, v in files.items()}})
        with open(root / "index.jsonl", "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
