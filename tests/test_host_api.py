"""CPU: the drop-in module's host-side contract: ctor,
attributes, constants, state_dict keys/shapes, shape arithmetic, exceptions,
and that CPU tensors are refused loudly (no fallback)."""
import pytest
import torch

from movenet_amd import wavenet as W
from movenet_amd.utils.weights import make_state_dict, parameter_shapes


def test_constants_and_solver():
    assert (W.MAX_AUDIO_FRAMES, W.MAX_VIDEO_FRAMES, W.VIDEO_KERNEL_SIZE, W.UPSAMPLE_STRIDE) == \
        (160000, 160, (1, 64, 64), 10)
    assert W.upsample_kernel_size_solver(160, 1600, stride=10) == (10,)
    assert W.upsample_kernel_size_solver(16000, 160000, stride=10) == (10,)


def test_state_dict_contract():
    cfg = dict(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    m = W.WaveNet(**cfg)
    want = parameter_shapes(**cfg)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got) == list(want)          # same keys, same registration order
    assert got == dict(want)
    assert sum(v.numel() for v in m.state_dict().values()) == 1491200  # SURVEY 2.2 C2 [probed]
    m.load_state_dict(make_state_dict(**cfg, seed=0), strict=True)
    for attr in ("layer_size", "stack_size", "input_channels", "residual_channels", "skip_channels"):
        assert getattr(m, attr) == cfg[attr]
    assert m.receptive_fields == 3072
    assert m.residual_conv_stack.dilations[:11] == [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1]
    # Lightning / DDP prefixes are plain key prefixes
    pref = {"model." + k: v for k, v in m.state_dict().items()}
    m.load_state_dict({k[len("model."):]: v for k, v in pref.items()})


def test_default_ctor_and_output_size():
    m = W.WaveNet(2, 2, 64)
    assert (m.residual_channels, m.skip_channels) == (16, 16)
    assert m.receptive_fields == 8
    x = torch.zeros(1, 64, 8)
    assert m.compute_output_size(x) == 1
    with pytest.raises(ValueError, match="receptive"):
        m.compute_output_size(torch.zeros(1, 64, 7))


def test_cpu_tensors_are_refused():
    m = W.WaveNet(2, 2, 64)
    x = torch.zeros(1, 64, 16)
    x[:, 0] = 1
    with pytest.raises(RuntimeError, match="MI355X"):
        m.generate(x, n_samples=20, temperature=0.0)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(x)


def test_reference_checkpoint_flavours_load(tmp_path):
    """Plain state_dict, DDP `module.` prefix, Lightning `{"state_dict": {"model.…"}}`."""
    from movenet_amd.checkpoint import load_into, load_state_dict_file
    cfg = dict(layer_size=2, stack_size=2, input_channels=16, residual_channels=8, skip_channels=8)
    sd = make_state_dict(**cfg, seed=3)
    torch.save(sd, tmp_path / "model.pth")
    torch.save({"module." + k: v for k, v in sd.items()}, tmp_path / "ddp.pth")
    torch.save({"epoch": 3, "state_dict": {"model." + k: v for k, v in sd.items()}}, tmp_path / "pl.ckpt")
    for name in ("model.pth", "ddp.pth", "pl.ckpt"):
        got = load_state_dict_file(tmp_path / name)
        assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
        m = W.WaveNet(**cfg)
        load_into(m, tmp_path / name)
        assert torch.equal(m.state_dict()["dense_conv.conv2.bias"], sd["dense_conv.conv2.bias"])
    torch.save([1, 2, 3], tmp_path / "junk.pth")
    with pytest.raises(ValueError):
        load_state_dict_file(tmp_path / "junk.pth")


def test_reference_block_and_type_names():
    """movenet/modules.py:15-142 and movenet/types.py:4-5 by name: the five block classes (same constructor signatures,
    same attribute / parameter names, parameter holders without arithmetic of their own) and the two layout aliases."""
    import typing

    from movenet_amd import modules as M
    from movenet_amd import types as T
    m = W.WaveNet(3, 2, 32, residual_channels=8, skip_channels=4)
    assert isinstance(m.causal_conv, M.CausalConv1d) and isinstance(m.dense_conv, M.DenseConv)
    assert isinstance(m.residual_conv_stack, M.ResidualConvStack)
    layer = m.residual_conv_stack.conv_layers[4]
    assert isinstance(layer, M.GatedResidualConv1d) and isinstance(layer.conv_filter, M.DilatedCausalConv1d)
    assert layer.conv_filter.conv.dilation == (2,) and tuple(layer.conv_skip.weight.shape) == (4, 8, 1)
    # the reference's positional / keyword signatures
    c = M.CausalConv1d(32, 8, kernel_size=2, bias=False)
    assert tuple(c.conv.weight.shape) == (8, 32, 2) and c.conv.padding == (1,) and c.conv.bias is None and c.kernel_size == 2
    d = M.DilatedCausalConv1d(8, dilation=4, kernel_size=2, bias=False)
    assert d.conv.dilation == (4,) and d.conv.padding == (0,)
    g = M.GatedResidualConv1d(8, 4, 16)
    assert [n for n, _ in g.named_children()] == ["conv_filter", "conv_gate", "context_conv_filter", "context_conv_gate",
                                                  "conv_residual", "conv_skip"]
    s = M.ResidualConvStack(3, 2, 8, 4)
    assert s.dilations == [1, 2, 4, 1, 2, 4] and len(s.conv_layers) == 6
    h = M.DenseConv(4, 32)
    assert tuple(h.conv1.weight.shape) == (32, 4, 1) and tuple(h.conv2.weight.shape) == (32, 32, 1)
    with pytest.raises(RuntimeError, match="holds parameters"):
        h(torch.zeros(1, 4, 3))
    # layout aliases: (batch, channels, frames) / (batch, frames, height, width, channels)
    assert typing.get_args(T.AudioTensor) == (torch.Tensor, ("batch", "channels", "frames"))
    assert typing.get_args(T.VideoTensor)[1] == ("batch", "frames", "height", "width", "channels")
    assert W.AudioTensor is T.AudioTensor and W.VideoTensor is T.VideoTensor
