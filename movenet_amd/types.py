"""Layout contract of the model API (drop-in for /root/reference/movenet/types.py:4-5).

The reference spells these with ``torchtyping.TensorType[...]``, which it uses in annotations only
(movenet/wavenet.py:15, :162).  torchtyping is not a dependency here: the aliases are
``typing.Annotated[torch.Tensor, <dimension names>]`` -- the same information for a reader or a
type checker, nothing at run time.  Layouts:

* ``AudioTensor``: ``(batch, channels, frames)`` -- one-hot mu-law classes on ``channels``,
  time contiguous;
* ``VideoTensor``: ``(batch, frames, height, width, channels)`` -- 64 x 64 grayscale frames.
"""
from typing import Annotated

import torch

AudioTensor = Annotated[torch.Tensor, ("batch", "channels", "frames")]
VideoTensor = Annotated[torch.Tensor, ("batch", "frames", "height", "width", "channels")]

__all__ = ["AudioTensor", "VideoTensor"]
