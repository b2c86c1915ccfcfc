"""Data-parallel training path on CPU: 2 ranks over `gloo`, each running the fused DSM step on its shard (kernels under
the test-only emulator), against ONE process on the full batch.  Checks the single exchange of the design
(DESIGN.md section 6): gradients pre-scaled by 1/world in the loss head + one SUM all-reduce of the flat gradient
buffer == full-batch gradient, and the replicas stay bit-identical after the optimizer step.  Sampling shards by
replication with no collective, so there is nothing to test there beyond per-rank seeds."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import emu
import _util

pytestmark = pytest.mark.skipif(not emu.available(), reason="emulator needs x86-64 + ROCm's clang++")


def _build(world_batch):
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    from score_sde_pytorch_amd import losses, sde_lib
    cfg = _util.small_config("ncsnpp")
    cfg.optim.warmup = 0
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    optimize_fn = losses.optimization_manager(cfg)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    g = torch.Generator().manual_seed(3)
    batch = torch.rand(world_batch, 3, 16, 16, generator=g)
    t = torch.rand(world_batch, generator=g) * (1 - 1e-5) + 1e-5
    z = torch.randn(world_batch, 3, 16, 16, generator=g)
    return cfg, model, state, step_fn, optimize_fn, batch, t, z


def _one_step(state, step_fn, optimize_fn, batch, t, z):
    fs = step_fn.fused_for(state, batch)
    loss = fs.loss_and_grads(batch, t=t, z=z).clone()
    grads_local = fs.flat.grad.clone()
    fs.optimizer_step(state["optimizer"], state["ema"], state["step"], optimize_fn.ssde_hyper)
    return float(loss), grads_local, fs.flat.grad.clone(), fs.flat.data.clone()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ["SSDE_GRAD_BUCKET_MB"] = "0.05"       # ~13k floats: the small test model splits into many buckets
    from score_sde_pytorch_amd import parallel
    parallel.init_from_env(backend="gloo")
    assert parallel.world_size() == world
    per = 2 if world == 2 else 1                      # samples per rank
    _, model, state, step_fn, optimize_fn, batch, t, z = _build(world * per)
    with emu.emulated():
        parallel.broadcast_parameters(model)
        sl = slice(rank * per, rank * per + per)
        assert torch.equal(parallel.shard_batch(batch), batch[sl])
        loss, g_local, g_sum, params = _one_step(state, step_fn, optimize_fn, batch[sl].clone(), t[sl].clone(), z[sl].clone())
    torch.save(dict(loss=loss, g_sum=g_sum, params=params), os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])            # 8 = the node the driver's scaling run uses (one sample per rank here)
def test_gradient_allreduce_matches_full_batch(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    r0 = rs[0]
    # replicas agree bit for bit after the exchange and the update
    for r in rs[1:]:
        assert torch.equal(r0["g_sum"], r["g_sum"]) and torch.equal(r0["params"], r["params"])
    # one process, full batch
    _, model, state, step_fn, optimize_fn, batch, t, z = _build(4 if world == 2 else world)
    with emu.emulated():
        loss, g_full, _, params = _one_step(state, step_fn, optimize_fn, batch, t, z)
    assert abs(sum(r["loss"] for r in rs) / world - loss) / abs(loss) < 1e-6
    scale = float(g_full.abs().max())
    assert float((r0["g_sum"] - g_full).abs().max()) / scale < 2e-5
    assert float((r0["params"] - params).abs().max()) / float(params.abs().max()) < 1e-5


def test_gradient_buckets_are_final_when_announced():
    """TrainEngine.run_backward_bucketed: at the moment a bucket is announced (the point where its all-reduce is issued)
    the slice already holds its final value, the buckets tile the flat buffer from its end to its start, and the segmented
    run equals the one-shot backward."""
    from score_sde_pytorch_amd import backward as B
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.small_config("ncsnpp", image_size=16, ch_mult=(1, 2), attn=(8,))
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    g = torch.Generator().manual_seed(1)
    x, sig, gout = torch.randn(2, 3, 16, 16, generator=g), torch.tensor([0.7, 3.0]), torch.randn(2, 3, 16, 16, generator=g)
    with emu.emulated():
        eng = B.TrainEngine(model, 2, 16, 16, torch.device("cpu"), dropout=False)
        eng.forward_train(x, sig)
        eng.backward(gout)
        ref = eng.flat.grad.clone()
        buckets = eng.grad_buckets(10000)
        assert len(buckets) > 3 and buckets[0][1] == eng.flat.numel and buckets[-1][0] == 0
        assert all(a[0] == b[1] for a, b in zip(buckets[:-1], buckets[1:]))            # contiguous, end -> start
        assert all(a[2] <= b[2] for a, b in zip(buckets[:-1], buckets[1:]))            # announced in program order
        eng.forward_train(x, sig)
        eng.gout.tensor[: gout.numel()].copy_(gout.reshape(-1))
        seen = []
        eng.run_backward_bucketed(lambda lo, hi: seen.append((lo, hi, eng.flat.grad[lo:hi].clone())), 10000)
        assert [(lo, hi) for lo, hi, _ in seen] == [(lo, hi) for lo, hi, _ in buckets]
        for lo, hi, snap in seen:
            assert torch.equal(snap, ref[lo:hi]), (lo, hi)
        assert torch.equal(eng.flat.grad, ref)


# ---- bench.py's rank logic (the code the driver's N > 1 run enters first), with a CPU device standing in for the GPU ----
def _bench_rank_worker(rank, world, port, out_dir):
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, _util.ROOT)
    import bench
    args = bench.parse(["--gpus", str(world), "--dist-backend", "gloo", "--share-device"])
    w, r, dev, d, ranks = bench.rank_setup(args, os.environ, n_devices=1, on_gpu=False)     # both ranks on "GPU 0"
    assert (w, r, ranks) == (world, rank, world) and d is not None
    sync_all = bench._sync_factory(dev, d)
    sync_all()
    # max over ranks of a per-rank wall time (the contract's "take the MAX over ranks"), every rank's own time beside it
    per_rank = []
    mx = bench._max_over_ranks(1.0 + rank, dev, d, per_rank)
    assert per_rank == [1.0 + i for i in range(world)]
    torch.save(dict(mx=mx, rank=r, world=w), os.path.join(out_dir, "bench_rank%d.pt" % rank))
    sync_all()
    d.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_rank_setup_gloo_ranks_share_one_device(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_bench_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), "bench_rank%d.pt" % r))
        assert got == dict(mx=float(world), rank=r, world=world)


def test_bench_launch_line_and_rank_guards():
    import sys
    sys.path.insert(0, _util.ROOT)
    import bench
    cmd = bench._launch_cmd(4, 29512, ["--gpus", "4", "--steps", "7"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    # world 1: no process group, device 0
    args = bench.parse([])
    assert bench.rank_setup(args, {}, n_devices=1, on_gpu=False)[:2] == (1, 0)
    assert args.matrix == os.environ.get("SSDE_MATRIX", "bf16x6") and not args.train_only
    assert bench.parse(["--train-only", "--matrix", "f32"]).train_only
    # a launcher whose world size disagrees with --gpus, a local rank without a GPU, RCCL with two ranks per device
    with pytest.raises(AssertionError):
        bench.rank_setup(bench.parse(["--gpus", "2"]), {"WORLD_SIZE": "4"}, n_devices=8, on_gpu=False)
    with pytest.raises(AssertionError):
        bench.rank_setup(bench.parse(["--gpus", "2"]), {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, n_devices=1, on_gpu=False)
    with pytest.raises(AssertionError):
        bench.rank_setup(bench.parse(["--gpus", "2", "--share-device"]), {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, n_devices=1, on_gpu=False)
