"""CPU: the C-ABI library loads without a GPU and exports every symbol that
include/movenet_hip.h declares; host-side shape arithmetic and error mapping."""
import os
import re

import pytest

from movenet_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "movenet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    names = _declared()
    assert len(names) >= 10
    lib = N.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in movenet_hip.h but not exported"
    assert sorted(N.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.mvn_abi_version() == 2


def test_receptive_fields_and_output_size():
    lib = N.lib()
    d2 = N.make_dims(10, 3, 256, 64, 64)
    assert lib.mvn_receptive_fields(d2) == 3072
    assert lib.mvn_receptive_fields(N.make_dims(10, 6, 256, 128, 128)) == 6144
    assert lib.mvn_receptive_fields(N.make_dims(2, 2, 64, 16, 16)) == 8
    assert lib.mvn_output_size(d2, 3072) == 1
    assert lib.mvn_output_size(d2, 16000) == 16000 - 3072 + 1
    rc = lib.mvn_output_size(d2, 3071)
    assert rc == N.MVN_ERR_TOO_SHORT
    with pytest.raises(ValueError, match="receptive"):
        N.check(rc, "mvn_output_size")
    assert lib.mvn_receptive_fields(N.make_dims(0, 3, 256, 64, 64)) == N.MVN_ERR_BAD_DIMS


def test_generate_sizes_and_variants():
    lib = N.lib()
    d2 = N.make_dims(10, 3, 256, 64, 64)
    # FOLD: 16 eleven-stage pipelines inside XCDs + 7 across them (r3), each serving up to 8 sequences in turn
    assert lib.mvn_gen_variant(d2, N.GEN_AUTO, 16) == N.GEN_FOLD
    assert lib.mvn_gen_variant(d2, N.GEN_AUTO, 17) == N.GEN_FOLD
    assert lib.mvn_gen_variant(d2, N.GEN_AUTO, 184) == N.GEN_FOLD
    assert lib.mvn_gen_variant(d2, N.GEN_AUTO, 185) == N.GEN_PIPE   # 24 pipelines x 8 rounds
    assert lib.mvn_gen_variant(d2, N.GEN_AUTO, 192) == N.GEN_PIPE
    assert lib.mvn_gen_variant(d2, N.GEN_AUTO, 193) == N.GEN_STREAM
    assert [lib.mvn_gen_launch_pipelines(d2, N.GEN_FOLD, n) for n in (1, 16, 17, 80, 81, 184)] == [1, 16, 16, 16, 23, 23]
    assert [lib.mvn_gen_launch_pipelines(d2, N.GEN_PIPE, n) for n in (5, 24, 100)] == [5, 24, 24]
    assert lib.mvn_gen_launch_pipelines(d2, N.GEN_STREAM, 4) == 0
    # PIPE: one workgroup per CU, 32 CUs per XCD: 3 nine-stage pipelines per XCD, 24 in all, each
    # serving up to 8 sequences in turn (r3)
    assert lib.mvn_gen_variant(d2, N.GEN_PIPE, 24) == N.GEN_PIPE
    assert lib.mvn_gen_variant(d2, N.GEN_PIPE, 192) == N.GEN_PIPE
    assert lib.mvn_gen_variant(d2, N.GEN_PIPE, 193) == N.MVN_ERR_UNSUPPORTED
    # BASELINE config 5 (60 layers, C=K=128): 61 stages span 2 XCDs -> 4 pipelines of up to 16 sequences
    d5 = N.make_dims(10, 6, 256, 128, 128)
    assert lib.mvn_gen_variant(d5, N.GEN_AUTO, 4) == N.GEN_PIPE
    assert lib.mvn_gen_variant(d5, N.GEN_AUTO, 64) == N.GEN_PIPE
    assert lib.mvn_gen_variant(d5, N.GEN_AUTO, 65) == N.GEN_GENERIC
    n5 = 2 * 256 * 128 + 60 * (6 * 128 * 128 + 2 * 128) + 256 * 128 + 256 + 256 * 256 + 256
    assert lib.mvn_gen_weights_floats(d5, N.GEN_PIPE) == n5 + 60 * (2 * 128 * 128 + 256)
    assert lib.mvn_gen_weights_floats(d5, N.GEN_GENERIC) == n5 + 60 * (2 * 128 * 128 + 256)
    assert lib.mvn_gen_state_floats(d5, 1) == 6138 * 128 + 61 * 512 + 128
    assert lib.mvn_gen_status_offset(d5, 1) == 6138 * 128 + 61 * 512
    # C=K=64: the hand-off area is sized for the largest pipelined variant (FOLD: 11 stages of
    # 192 granules), the status word follows its granules
    assert lib.mvn_gen_status_offset(d2, 16) == 16 * (3069 * 64 + 11 * 384)
    assert lib.mvn_gen_variant(d2, N.GEN_FOLD, 184) == N.GEN_FOLD
    assert lib.mvn_gen_variant(d2, N.GEN_FOLD, 185) == N.MVN_ERR_UNSUPPORTED
    assert lib.mvn_gen_variant(d5, N.GEN_PIPE_F16, 64) == N.GEN_PIPE_F16   # 8 pipelines x 8 rounds
    assert lib.mvn_gen_variant(d5, N.GEN_PIPE_F16, 65) == N.MVN_ERR_UNSUPPORTED
    assert lib.mvn_gen_variant(d2, N.GEN_PIPE_F16, 1) == N.MVN_ERR_UNSUPPORTED
    d1 = N.make_dims(2, 2, 64, 16, 16)
    assert lib.mvn_gen_variant(d1, N.GEN_AUTO, 2) == N.GEN_GENERIC
    assert lib.mvn_gen_variant(d1, N.GEN_STREAM, 2) == N.MVN_ERR_UNSUPPORTED
    assert lib.mvn_gen_variant(d1, N.GEN_PIPE, 2) == N.MVN_ERR_UNSUPPORTED
    # audio-path parameters of the 30-layer model
    # + the context-conv section: 30 layers x (128x64 weights + 128 biases)
    ctx = 30 * (128 * 64 + 128)
    assert lib.mvn_gen_weights_floats(d2, N.GEN_GENERIC) == 856320 + ctx
    assert lib.mvn_gen_weights_floats(d2, N.GEN_STREAM) == 856320 + ctx
    assert lib.mvn_gen_weights_floats(d2, N.GEN_PIPE) == 856320 + ctx
    # dilation queues: D*C floats per sequence (786 KB fp32) + the pipelined variants'
    # hand-off area (the largest: FOLD's 11 stages x 192 eight-byte granules per sequence + flags)
    assert lib.mvn_gen_state_floats(d2, 1) == 3069 * 64 + 11 * 384 + 64
    assert lib.mvn_gen_state_floats(d2, 16) == 16 * (3069 * 64 + 11 * 384) + 192
    assert lib.mvn_gen_state_floats(d1, 2) == 2 * 6 * 16
    assert lib.mvn_gen_status_offset(d1, 2) == 2 ** 64 - 1


def test_bad_arguments_are_refused_before_any_launch():
    lib = N.lib()
    d2 = N.make_dims(10, 3, 256, 64, 64)
    rc = lib.mvn_generate(d2, N.GEN_STREAM, None, None, None, 1, 10, 10, 1, 0, 5, 0.0, 0, None, None, 0, None, None)
    assert rc == N.MVN_ERR_BAD_ARG
    with pytest.raises(ValueError):
        N.check(rc, "mvn_generate")
    assert "bad argument" in N.last_error()


def test_loss_kernels_argument_checks():
    """Row F3 entry points: sizes and refusals that need no GPU."""
    lib = N.lib()
    assert lib.mvn_ce_parts(16, 12928) == 16 * 202     # one partial sum per 64-column workgroup
    assert lib.mvn_ce_parts(0, 100) == 0 and lib.mvn_ce_parts(3, 0) == 0 and lib.mvn_ce_parts(-1, 5) == 0
    assert lib.mvn_ce_on_probs_forward(None, None, 2, 64, 10, None, None, None) == N.MVN_ERR_BAD_ARG
    assert "mvn_ce_on_probs_forward" in N.last_error()
    assert lib.mvn_ce_on_probs_backward(None, None, 2, 64, 10, 1.0, None, None, None) == N.MVN_ERR_BAD_ARG


def test_pipelined_launch_form_policy():
    """The pipelined generators launch cooperatively unless a profiler's tool library is attached or the caller
    opts out (csrc/pipe_common.h; decided once per process, so each case is a process of its own)."""
    import subprocess
    import sys
    code = "from movenet_amd import _native as N; print(N.lib().mvn_gen_launch_is_cooperative())"
    base = {k: v for k, v in os.environ.items()
            if k not in ("MOVENET_PIPE_COOPERATIVE_LAUNCH", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "LD_PRELOAD")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra, want in (({}, "1"), ({"MOVENET_PIPE_COOPERATIVE_LAUNCH": "0"}, "0"),
                        ({"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"}, "0"),
                        ({"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so",
                          "MOVENET_PIPE_COOPERATIVE_LAUNCH": "1"}, "1")):
        out = subprocess.run([sys.executable, "-c", code], env={**base, **extra}, cwd=root, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == want, (extra, out.stdout, out.stderr)
