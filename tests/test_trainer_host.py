"""CPU: trainer-side host logic -- CLI flags, config round trip, synthetic
batches, optimizer/scheduler factories, and the world_size-2 data-parallel
gradient exchange over gloo."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from movenet_amd.config import ModelConfig, TrainingConfig, arg_parser, config_from_args
from movenet_amd.dataset import get_dataloader, mu_law_decoding, mu_law_encoding


def test_cli_defaults_match_reference_parser():
    # defaults of /root/reference/movenet/config.py:149-240
    a = arg_parser().parse_args([])
    assert (a.batch_size, a.val_batch_size, a.optimizer, a.learning_rate) == (3, 3, "AdamW", 0.001)
    assert (a.input_channels, a.residual_channels, a.skip_channels, a.layer_size, a.stack_size) == \
        (16, 16, 8, 3, 3)
    assert a.scheduler is None and a.lr_pct_start == 0.45 and a.max_learning_rate == 0.003
    assert a.use_video is True and a.dist_backend == "nccl" and a.dist_port == "8888"
    assert a.n_epochs == 10 and a.accumulation_steps == 1 and a.gradient_clipping == 0.0


def test_config_from_args_and_q11_quirk():
    a = arg_parser().parse_args(
        "--dataset synthetic://clips=4,frames=100 --use_video 0 --input_channels 64 "
        "--residual_channels 16 --skip_channels 16 --layer_size 2 --stack_size 2 --batch_size 2 "
        "--n_epochs 1 --gradient_clipping 5.0 --scheduler_milestones [1,2] --optimizer SGD".split())
    c = config_from_args(a)
    assert c.model_config == ModelConfig(2, 2, 64, 16, 16)
    assert c.use_video is False and c.batch_size == 2 and c.optimizer == "SGD"
    assert c.scheduler_milestones == [1, 2]
    assert c.gradient_clipping == 0.0  # Q11: the flag is parsed but never copied
    back = TrainingConfig.from_json(c.to_json())
    assert back.model_config == c.model_config and back.batch_size == 2
    assert json.loads(c.to_json())["model_config"]["input_channels"] == 64
    # dataclass defaults (config.py:11-94)
    d = TrainingConfig()
    assert (d.learning_rate, d.scheduler, d.n_epochs, d.checkpoint_every) == (1e-4, "OneCycleLR", 100, 25)


def test_synthetic_batches_and_sharding():
    ld = get_dataloader("synthetic://clips=10,frames=50,seed=3", input_channels=16, batch_size=4,
                        use_video=False)
    batches = list(ld)
    assert len(ld) == 3 and len(batches) == 3
    audio, video, contexts, fps, info = batches[0]
    assert audio.shape == (4, 16, 50) and video is None and len(contexts) == 4
    assert torch.equal(audio.sum(1), torch.ones(4, 50))
    assert batches[-1].audio.shape[0] == 2
    again = list(get_dataloader("synthetic://clips=10,frames=50,seed=3", 16, 4, use_video=False))
    assert torch.equal(again[1].audio, batches[1].audio)
    # two ranks see disjoint clips covering the set
    seen = []
    for r in range(2):
        l = get_dataloader("synthetic://clips=10,frames=50,seed=3", 16, 5, use_video=False,
                           rank=r, world_size=2)
        seen.append({f for b in l for f in b.filepaths})
    assert not (seen[0] & seen[1]) and len(seen[0] | seen[1]) == 10
    crop = next(iter(get_dataloader("synthetic://clips=2,frames=100", 16, 2, use_video=False,
                                    batch_subsample_frac=0.25)))
    assert crop.audio.shape == (2, 16, 25)
    with pytest.raises(ValueError):
        get_dataloader("synthetic://clips=2,frames=100", 16, 2, use_video=True)
    vb = next(iter(get_dataloader("synthetic://clips=2,frames=2000", 16, 2, use_video=True)))
    assert vb.video.shape == (2, 2, 64, 64, 1) and vb.audio.shape == (2, 16, 2000)
    with pytest.raises(ValueError):
        get_dataloader("/data/kinetics", 16, 2, use_video=False)


def test_mu_law_roundtrip_formula():
    x = torch.linspace(-1, 1, 1001)
    q = mu_law_encoding(x, 256)
    assert q.min() == 0 and q.max() == 255 and torch.all(q[1:] >= q[:-1])
    y = mu_law_decoding(q, 256)
    assert (y - x).abs().max() < 0.03 and abs(float(mu_law_decoding(torch.tensor([128]), 256))) < 0.01


def test_optimizer_and_scheduler_factories():
    from movenet_amd.pytorch_lightning_trainer import Dance2Music
    cfg = TrainingConfig(model_config=ModelConfig(2, 2, 16, 8, 8), batch_size=2, n_epochs=3,
                         use_video=False, scheduler="OneCycleLR", accumulation_steps=2)
    m = Dance2Music("synthetic://clips=8,frames=40", cfg)
    o = m.configure_optimizers()
    assert isinstance(o["optimizer"], torch.optim.AdamW)
    sch = o["lr_scheduler"]["scheduler"]
    assert isinstance(sch, torch.optim.lr_scheduler.OneCycleLR) and o["lr_scheduler"]["interval"] == "step"
    assert sch.total_steps == 3 * 2  # epochs * ceil(4 batches / 2 accumulation)
    for name, klass in (("SGD", torch.optim.SGD), ("RMSprop", torch.optim.RMSprop), ("Adam", torch.optim.Adam)):
        cfg2 = TrainingConfig(model_config=ModelConfig(2, 2, 16, 8, 8), optimizer=name, scheduler="StepLR",
                              use_video=False)
        assert isinstance(Dance2Music("synthetic://clips=2,frames=40", cfg2).configure_optimizers()["optimizer"], klass)
    with pytest.raises(ValueError, match="optimizer"):
        Dance2Music("synthetic://clips=2,frames=40",
                    TrainingConfig(optimizer="LBFGS", use_video=False)).configure_optimizers()
    with pytest.raises(ValueError, match="scheduler"):
        Dance2Music("synthetic://clips=2,frames=40",
                    TrainingConfig(scheduler="Cosine", use_video=False)).configure_optimizers()


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from movenet_amd.parallel import FlatGradSync, init_distributed
    r, w, _ = init_distributed("gloo", str(port))
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different initial weights per rank on purpose
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3),
                              torch.nn.Linear(3, 3))
    for p in net[3].parameters():
        p.requires_grad_(True)
    sync = FlatGradSync(net.parameters(), w)
    sync.broadcast_parameters(0)
    torch.manual_seed(7 + rank)
    x = torch.randn(4, 5)
    net[2](net[1](net[0](x))).pow(2).sum().backward()  # net[3] unused: its grads stay None
    local = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
    sent = sync.sync_gradients()
    # second exchange: the same gradients as views of ONE flat buffer with a gap (what
    # ops.py's backward hands to autograd) -> reduced in place, no staging copies
    used = [p for p in net.parameters() if p.grad is not None]
    flat = torch.zeros(sum(p.numel() for p in used) + 11)
    off = 0
    for i, (p, g) in enumerate(zip(used, [x for x in local if x is not None])):
        if i == 2:
            off += 11  # a parameter without gradient in between
        p.grad = flat[off:off + p.numel()].view_as(p)
        p.grad.copy_(g)
        off += p.numel()
    assert FlatGradSync._contiguous_span(used) is not None
    sent_flat = sync.sync_gradients()
    torch.save({"params": [p.detach().clone() for p in net.parameters()], "local": local,
                "synced": [p.grad.clone() if p.grad is not None else None for p in net.parameters()],
                "sent": sent, "sent_flat": sent_flat},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_exchange_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=True) for i in range(world)]
    for a, b in zip(r[0]["params"], r[1]["params"]):
        assert torch.equal(a, b)  # rank 0's weights everywhere
    assert r[0]["sent"] == r[1]["sent"] == 5 * 7 + 7 + 7 * 3 + 3
    assert r[0]["sent_flat"] == r[1]["sent_flat"] == 5 * 7 + 7 + 7 * 3 + 3 + 11  # the span, gap included
    for i, (g0, g1) in enumerate(zip(r[0]["synced"], r[1]["synced"])):
        if g0 is None:
            assert g1 is None and r[0]["local"][i] is None  # unused parameters are left alone
            continue
        assert torch.equal(g0, g1)
        assert torch.allclose(g0, (r[0]["local"][i] + r[1]["local"][i]) / 2, atol=1e-6)


def _conditioned_layout_worker(rank: int, world: int, port: int, out_dir: str):
    """BASELINE configs[3]'s gradient layout on two ranks: the conditioned decoder's gradients
    in ops.decoder_param_names(with_context=True) order (NOT registration order: the context
    convs come after the skip conv there, before the residual conv in the module) with the eight
    video-encoder gradients behind them in one flat buffer -- what ops._run_backward +
    VideoGradSlot hand to autograd -- must travel as ONE in-place all-reduce."""
    import hashlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from movenet_amd.ops import VIDEO_PARAMS, decoder_param_names
    from movenet_amd.parallel import FlatGradSync, init_distributed
    from movenet_amd.wavenet import WaveNet
    init_distributed("gloo", str(port))
    torch.manual_seed(50 + rank)
    model = WaveNet(layer_size=2, stack_size=2, input_channels=8, residual_channels=4, skip_channels=4)
    sync = FlatGradSync(model.parameters(), world)
    sync.broadcast_parameters(0)
    lookup = dict(model.named_parameters())
    L = 4
    names = decoder_param_names(L, with_context=True) + list(VIDEO_PARAMS)
    assert sorted(names) == sorted(lookup)  # the layout covers every parameter of the model
    last = f"residual_conv_stack.conv_layers.{L - 1}.conv_residual."
    flat = torch.zeros(sum(lookup[n].numel() for n in names))
    gen = torch.Generator().manual_seed(900 + rank)
    off = 0
    for n in names:
        p = lookup[n]
        if not n.startswith(last):  # the last layer's residual conv gets no gradient: a gap
            p.grad = flat[off:off + p.numel()].view_as(p)
            p.grad.copy_(torch.randn(p.shape, generator=gen))
        off += p.numel()
    local = {n: lookup[n].grad.clone() for n in names if lookup[n].grad is not None}
    sent = sync.sync_gradients()
    with torch.no_grad():
        for p in model.parameters():
            if p.grad is not None:
                p.add_(p.grad, alpha=-0.1)
    digest = hashlib.sha256(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy().tobytes()).hexdigest()
    torch.save({"path": sync.last_path, "sent": sent, "digest": digest, "n": flat.numel(), "local": local,
                "synced": {n: lookup[n].grad.clone() for n in local}}, os.path.join(out_dir, f"c{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_conditioned_gradient_layout_is_one_message_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_conditioned_layout_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"c{i}.pt", weights_only=True) for i in range(world)]
    assert r[0]["path"] == r[1]["path"] == "contiguous-span"
    assert r[0]["sent"] == r[1]["sent"] == r[0]["n"]  # the whole buffer, gap included, one message
    assert r[0]["digest"] == r[1]["digest"]           # identical replicas after the step
    for n, g in r[0]["synced"].items():
        assert torch.allclose(g, (r[0]["local"][n] + r[1]["local"][n]) / 2, atol=1e-6), n


def _config4_worker(rank: int, world: int, port: int, out_dir: str):
    """BASELINE configs[3] (video-conditioned, global batch 64 = 8 clips per rank on 8 ranks) as far as a CPU can take
    it: the REAL model's parameter set (30 layers, C = K = 64, Q = 256: 1 491 200 floats), every rank's shard of a
    64-clip synthetic dataset, the per-rank input seeds of SURVEY 8d (1234 + rank / 4321 + rank), the conditioned
    gradient layout as ONE flat span, two optimizer steps."""
    import hashlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    import numpy as np
    from movenet_amd.dataset import SyntheticLoader
    from movenet_amd.ops import VIDEO_PARAMS, decoder_param_names
    from movenet_amd.parallel import FlatGradSync, init_distributed
    from movenet_amd.utils.weights import synthetic_indices
    from movenet_amd.wavenet import WaveNet
    r, w, _ = init_distributed("gloo", str(port))
    assert (r, w) == (rank, world)
    torch.manual_seed(50 + rank)  # different initial weights per rank on purpose: the broadcast must fix that
    model = WaveNet(layer_size=10, stack_size=3, input_channels=256, residual_channels=64, skip_channels=64)
    sync = FlatGradSync(model.parameters(), world)
    sync.broadcast_parameters(0)
    # the rank's shard: 8 of the 64 clips, one batch of 8 (config 3's per-GPU batch)
    loader = SyntheticLoader("synthetic://clips=64,frames=2000,seed=1234", 256, batch_size=8, rank=rank, world_size=world,
                             use_video=True)
    batches = list(loader)
    clip_ids = [int(f.rsplit("/", 1)[1]) for b in batches for f in b.filepaths]
    assert len(batches) == 1 and tuple(batches[0].audio.shape) == (8, 256, 2000) and tuple(batches[0].video.shape) == (8, 2, 64, 64, 1)
    # bench.py's synthetic inputs: seeds 1234 + rank (audio classes) and 4321 + rank (frames) differ by rank
    audio_idx = synthetic_indices(8, 64, 256, 1234 + rank)
    frames = np.random.default_rng(4321 + rank).random((2, 4), dtype=np.float32)
    lookup = dict(model.named_parameters())
    L = 30
    names = decoder_param_names(L, with_context=True) + list(VIDEO_PARAMS)
    assert sorted(names) == sorted(lookup)
    last = f"residual_conv_stack.conv_layers.{L - 1}.conv_residual."
    n_total = sum(lookup[n].numel() for n in names)
    digests, sent, paths = [], [], []
    for step in range(2):
        flat = torch.zeros(n_total)  # what ops._run_backward + VideoGradSlot hand to autograd: one zero-filled buffer
        gen = torch.Generator().manual_seed(1000 * step + 1234 + rank)
        off = 0
        for n in names:
            p = lookup[n]
            if not n.startswith(last):  # the last layer's residual conv gets no gradient: a zero gap
                p.grad = flat[off:off + p.numel()].view_as(p)
                p.grad.copy_(torch.randn(p.shape, generator=gen) * 1e-2)
            off += p.numel()
        sent.append(sync.sync_gradients())
        paths.append(sync.last_path)
        with torch.no_grad():
            for p in model.parameters():
                if p.grad is not None:
                    p.add_(p.grad, alpha=-0.1)
        digests.append(hashlib.sha256(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy().tobytes()).hexdigest())
    torch.save({"clips": clip_ids, "audio_sum": int(audio_idx.sum()), "frames_sum": float(frames.sum()), "n_total": n_total,
                "sent": sent, "paths": paths, "digests": digests}, os.path.join(out_dir, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_config4_world_size_eight_gloo(tmp_path):
    """BASELINE configs[3] without the hardware: 8 ranks, global batch 64 -> 8 clips per rank, disjoint shards whose
    union is the dataset, rank-offset seeds, the conditioned gradient span = every parameter of the model as ONE
    message on every rank, identical replicas after two steps.  Precedent: movenet/trainer.py:226-238 (DDP),
    movenet/dataset.py:78-86 (DistributedSampler)."""
    world, port = 8, _free_port()
    mp.spawn(_config4_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"w{i}.pt", weights_only=True) for i in range(world)]
    shards = [set(x["clips"]) for x in r]
    assert all(len(x["clips"]) == 8 == len(s) for x, s in zip(r, shards))
    assert set().union(*shards) == set(range(64)) and sum(len(s) for s in shards) == 64   # a partition of the dataset
    assert len({x["audio_sum"] for x in r}) == world and len({x["frames_sum"] for x in r}) == world  # per-rank inputs
    for x in r:
        assert x["n_total"] == 1491200 and x["sent"] == [1491200, 1491200] and x["paths"] == ["contiguous-span"] * 2
        assert x["digests"] == r[0]["digests"]   # identical replicas after each step
    assert r[0]["digests"][0] != r[0]["digests"][1]


def test_contiguous_span_rejects_overlap_and_foreign_storage():
    from movenet_amd.parallel import contiguous_grad_span
    a, b = torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(4))
    flat = torch.zeros(12)
    a.grad, b.grad = flat[6:10], flat[0:4]           # decreasing order, a gap: fine
    span = contiguous_grad_span([a, b])
    assert span is not None and span.numel() == 10 and span.data_ptr() == flat.data_ptr()
    a.grad, b.grad = flat[2:6], flat[0:4]            # overlapping views
    assert contiguous_grad_span([a, b]) is None
    a.grad, b.grad = flat[0:4], torch.zeros(4)       # two storages
    assert contiguous_grad_span([a, b]) is None


def test_bench_host_side_arithmetic():
    """bench.py's host-side pieces that no GPU run exercises here: the byte and FLOP counts of a train step and
    the CPU baseline of the train step (one step of the oracle at a reduced length; the bench line's sample is B=2, T=16000)."""
    import bench
    flop = bench.train_flop_per_step(bench.CFG, 16, 16000)
    assert abs(flop - 1.1256e12) / 1.1256e12 < 1e-3          # 1.126e12 FLOP per config-2 step
    nb = bench.train_bytes_per_step(bench.CFG, 16, 16000)
    assert nb["floor_bytes"] < nb["design_bytes"] < 3 * nb["floor_bytes"]
    roof = bench.train_roofline(flop, nb, 10e-3, "train_config2")
    assert roof["bound"] == "hbm" and roof["traffic"] is None and 0 < roof["frac"] < 1 and 0 < roof["mfma"]["frac"] < 1
    from movenet_amd.utils.weights import make_state_dict
    r = bench.cpu_train_baseline(make_state_dict(**bench.CFG, seed=0), batch=1, t_len=3072 + 256, n_timed=1, n_warm=0)
    assert r["unit"] == "tokens/s" and r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1
