"""The reference's five block classes as PARAMETER HOLDERS (drop-in names for
/root/reference/movenet/modules.py:15-142).

``movenet_amd.wavenet.WaveNet`` is assembled from these exactly as the reference assembles its model
(movenet/wavenet.py:119-123): same constructor signatures, same attribute names, same registration
order -- so ``state_dict`` keys, parameter shapes and the default initialisation drawn under a given
``torch.manual_seed`` are the reference's.  What they do NOT have is arithmetic of their own: the
layers of this build are fused HIP kernels that run a whole forward / backward / generation step
(``mvn_forward``, ``mvn_backward``, ``mvn_generate``), not a module-by-module graph, so calling one
of these blocks directly raises.  Code that only constructs, inspects, initialises, freezes or
checkpoints the blocks (``isinstance`` checks, ``named_parameters``, weight surgery) works unchanged.
"""
from __future__ import annotations

from typing import List

import torch.nn as nn

__all__ = ["CausalConv1d", "DilatedCausalConv1d", "GatedResidualConv1d", "ResidualConvStack", "DenseConv"]


class _Holder(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover
        raise RuntimeError(
            f"{type(self).__name__} holds parameters under the reference's names; its arithmetic runs inside the fused "
            "HIP kernels of movenet_amd.wavenet.WaveNet.forward / .generate (see movenet_amd/modules.py)")


class CausalConv1d(_Holder):
    """movenet/modules.py:15-30: ``Conv1d(in, out, k, padding=1, bias)`` whose last output column is dropped."""

    def __init__(self, input_channels, out_channels, kernel_size=2, bias=False):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv = nn.Conv1d(input_channels, out_channels, kernel_size, padding=1, bias=bias)


class DilatedCausalConv1d(_Holder):
    """movenet/modules.py:33-46: ``Conv1d(C, C, k, dilation=d, padding=0, bias)``."""

    def __init__(self, channels, dilation=1, kernel_size=2, bias=False):
        super().__init__()
        self.conv = nn.Conv1d(channels, channels, kernel_size, dilation=dilation, padding=0, bias=bias)


class GatedResidualConv1d(_Holder):
    """movenet/modules.py:49-93: filter / gate dilated convs, the two context 1x1 convs, residual and skip 1x1 convs."""

    def __init__(self, residual_channels, skip_channels, dilation):
        super().__init__()
        self.conv_filter = DilatedCausalConv1d(residual_channels, dilation=dilation)
        self.conv_gate = DilatedCausalConv1d(residual_channels, dilation=dilation)
        self.context_conv_filter = nn.Conv1d(residual_channels, residual_channels, 1)
        self.context_conv_gate = nn.Conv1d(residual_channels, residual_channels, 1)
        self.conv_residual = nn.Conv1d(residual_channels, residual_channels, 1)
        self.conv_skip = nn.Conv1d(residual_channels, skip_channels, 1)


class ResidualConvStack(_Holder):
    """movenet/modules.py:96-130: ``stack_size`` cycles of dilations 1, 2, ..., 2^(layer_size - 1)."""

    def __init__(self, layer_size, stack_size, residual_channels, skip_channels):
        super().__init__()
        self.layer_size = layer_size
        self.stack_size = stack_size
        self.conv_layers = nn.ModuleList([
            GatedResidualConv1d(residual_channels, skip_channels, dilation) for dilation in self.dilations
        ])

    @property
    def dilations(self) -> List[int]:
        return [2 ** x for _ in range(self.stack_size) for x in range(self.layer_size)]


class DenseConv(_Holder):
    """movenet/modules.py:133-142: two 1x1 convs behind leaky ReLUs (the mu-law head)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv1d(in_channels, out_channels, 1)
        self.conv2 = nn.Conv1d(out_channels, out_channels, 1)
