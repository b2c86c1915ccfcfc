#!/usr/bin/env python
"""Headline benchmark: PC-sampler throughput of NCSN++ cont. VE-SDE on CIFAR-10 (BASELINE configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (per GPU): configs/ve/cifar10_ncsnpp_continuous, batch 256, reverse_diffusion predictor +
Langevin corrector, snr 0.16, N=1000 discretisation steps.  One "step" = ONE full PC iteration on
the batch = 2 U-Net evaluations + rocRAND noise + norms + Langevin update + predictor update, replayed
as one hipGraph.  All N iterations are identical work, so images/sec = n_gpus * batch / (1000 * t_step);
the timed region holds exactly K iterations with the state resident in HBM.  Sampling shards by
replication (one independent sampler per GPU, no collectives) -> "scaling": "weak".
Weights: deterministic random init of the full architecture (every tensor re-randomised);
data: synthetic prior noise.  fp32 end to end (dtype "f32"), as the reference.

The JSON line also carries
  roofline     -- the dominant kernel (conv_mfma: every 3x3 / fused 3x3+1x1 convolution launch of one
                  U-Net evaluation): algorithmic FLOPs / HIP-event time on the launch stream, against
                  the 157.3 TFLOP/s fp32 MFMA peak;
  cpu_baseline -- the CPU oracle (a torch-CPU port of the reference path, oracle/) timed on this host
                  for a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--sde-steps", type=int, default=1000)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--dump-ops", type=str, default="")
    ap.add_argument("--no-train", action="store_true", help="skip the DSM training-step measurement")
    ap.add_argument("--workload", default="cifar10", choices=["cifar10", "ffhq256", "subvp_ode"],
                    help="cifar10 = BASELINE configs[1] (the headline line); ffhq256 = configs[3] (NCSN++ 256x256, N=2000, batch "
                         "16/GPU); subvp_ode = configs[4] (DDPM++ sub-VP, probability-flow ODE sampler with RK45; 1 step = 1 solve)")
    ap.add_argument("--train-batch", type=int, default=128)
    ap.add_argument("--train-steps", type=int, default=0, help="timed training steps (0: same as --steps, capped at 10)")
    ap.add_argument("--dump-train-ops", type=str, default="")
    return ap.parse_args()


def bench_train(args, cfg, dev, dist, world, rank, sync_all):
    """sec/train-step of the fused DSM step (BASELINE configs[2]): batch 128/GPU, dropout 0.1, Adam + clip + EMA,
    one all-reduce of the flat gradient per step when world > 1."""
    import _util
    from score_sde_pytorch_amd import sde_lib, losses, backward as B, engine as E
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.to(dev).train()
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    optimize_fn = losses.optimization_manager(cfg)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, reduce_mean=cfg.training.reduce_mean,
                                 continuous=True, likelihood_weighting=cfg.training.likelihood_weighting)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    Bt, R = args.train_batch, cfg.data.image_size
    torch.manual_seed(100 + rank)
    batch = torch.rand(Bt, 3, R, R, device=dev)
    k = args.train_steps or min(args.steps, 10)
    w = max(1, min(args.warmup, 3))
    for _ in range(w):
        loss = step_fn(state, batch)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(k):
        loss = step_fn(state, batch)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    sec = dt / k
    fs = step_fn.fused_for(state, batch)
    eng = fs.eng
    fl = np.array(eng.program.flops)
    out = {"metric": "sec_per_train_step", "value": sec, "unit": "s/step", "higher_is_better": False,
           "batch_per_gpu": Bt, "global_batch": Bt * world, "images_per_sec": world * Bt / sec, "steps": k, "warmup": w,
           "loss": float(loss), "dropout": float(cfg.model.dropout), "path": "fused (losses.FusedTrainStep)",
           "algorithmic_tflops": float(fl.sum()) / sec / 1e12, "gflop_per_image": float(fl.sum()) / Bt / 1e9,
           "grad_allreduce_mb": eng.flat.numel * 4 / 1e6 if world > 1 else 0.0,
           "arena_gb": eng.b.arena_bytes / 1e9}
    if rank == 0 and not args.no_roofline:
        ms = np.array(eng.program.run_range_timed(0, eng.program.n))
        ms = np.array(eng.program.run_range_timed(0, eng.program.n))
        cls = np.array(eng.program.classes)
        idx = np.arange(eng.program.n)
        by = {}
        for name, c in [("conv3x3_fwd+dgrad", E.FC_CONV3), ("conv1x1_fwd+dgrad", E.FC_CONV1), ("wgrad", B.FC_WGRAD),
                        ("attention", E.FC_ATTN), ("groupnorm_stats", E.FC_GN), ("upfirdn", E.FC_FIR),
                        ("backward_elementwise", B.FC_BWD), ("other", E.FC_OTHER)]:
            m = cls == c
            by[name] = {"launches": int(m.sum()), "ms": float(ms[m].sum()), "gflop": float(fl[m].sum()) / 1e9}
        out["fwd_ms_events"] = float(ms[idx < eng.n_fwd].sum())
        out["bwd_ms_events"] = float(ms[idx >= eng.n_fwd].sum())
        out["by_class"] = by
        wg = cls == B.FC_WGRAD
        out["wgrad_tflops"] = float(fl[wg].sum()) / max(float(ms[wg].sum()), 1e-9) / 1e9
        if args.dump_train_ops:
            rows = []
            for i in range(eng.program.n):
                op = eng.program.ops[i]
                row = {"i": i, "kind": int(op.kind), "cls": int(cls[i]), "ms": float(ms[i]), "gflop": float(fl[i]) / 1e9,
                       "bwd": bool(i >= eng.n_fwd)}
                if op.kind == 1:
                    c = op.u.conv
                    row.update(h=c.h_out, w=c.w_out, cout=c.c_out, cin=c.main.c0 + c.main.c1, caux=c.aux.c0 + c.aux.c1, ks=c.ksize)
                elif op.kind == 15:
                    c = op.u.wgrad
                    row.update(h=c.h_out, w=c.w_out, cout=c.c_out, cin=c.src.c0 + c.src.c1, ks=c.ksize)
                rows.append(row)
            with open(args.dump_train_ops, "w") as f:
                json.dump(rows, f)
    return out


def bench_ode(args, dev, dist, world, rank):
    """BASELINE configs[4]: CIFAR-10 DDPM++ sub-VP, probability-flow ODE sampler (sampling.py:414-485: RK45, rtol = atol =
    1e-5, eps 1e-3) on the device-resident integrator.  One step = one full solve of a batch; the number of function
    evaluations is adaptive and is reported (random-init weights of the architecture: the NFE of a trained net differs)."""
    import _util
    from score_sde_pytorch_amd import sde_lib, sampling
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.cfgs.get_config("subvp/cifar10_ddpmpp_continuous")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.to(dev).eval()
    B, R = args.batch, cfg.data.image_size
    sde = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
    sampler = sampling.get_ode_sampler(sde, (B, 3, R, R), lambda v: v, denoise=cfg.sampling.noise_removal, rtol=1e-5, atol=1e-5,
                                       method="RK45", eps=1e-3, device=dev)
    torch.manual_seed(1234 + rank)
    steps, warm = max(1, min(args.steps, 3)), max(1, min(args.warmup, 1))

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)
    for _ in range(warm):
        sampler(model, z=sde.prior_sampling((B, 3, R, R)).to(dev))
    sync_all()
    t0 = time.perf_counter()
    nfes = []
    for _ in range(steps):
        x, nfe = sampler(model, z=sde.prior_sampling((B, 3, R, R)).to(dev))
        nfes.append(int(nfe))
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = {"metric": "ode_sampler_images_per_sec", "value": world * B * steps / dt, "unit": "images/s", "n_gpus": world, "steps": steps,
           "warmup": warm, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (f64 integrator state)", "data": "synthetic",
           "config": {"workload": "configs/subvp/cifar10_ddpmpp_continuous probability-flow ODE sampler (RK45 rtol=atol=1e-5, eps=1e-3), "
                                  "batch %d/GPU, 32x32; 1 step = 1 solve" % B,
                      "batch_per_gpu": B, "nfe_per_solve": nfes, "state_finite": bool(torch.isfinite(x).all()),
                      "ms_per_nfe": dt / max(sum(nfes), 1) * 1e3, "parallelism": "replicas x%d (no collectives)" % world}}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (the HIP path has no CPU fallback)"
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import _util
    from score_sde_pytorch_amd import sde_lib, sampling, engine as E
    from score_sde_pytorch_amd.models import utils as mutils

    if args.workload == "subvp_ode":
        return bench_ode(args, dev, dist, world, rank)
    if args.workload == "ffhq256":
        cfg = _util.cfgs.get_config("ve/ffhq_256_ncsnpp_continuous")
        if args.batch == 256:
            args.batch = 16
        if args.sde_steps == 1000:
            args.sde_steps = 2000
        args.no_train = args.no_cpu_baseline = True
    else:
        cfg = _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = _util.load_seeded(model, seed=1)
    model = model.to(dev).eval()
    B, R = args.batch, cfg.data.image_size
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, N=args.sde_steps)
    sampler = sampling.get_pc_sampler(sde, (B, 3, R, R), sampling.get_predictor(cfg.sampling.predictor),
                                      sampling.get_corrector(cfg.sampling.corrector), lambda v: v, snr=cfg.sampling.snr,
                                      n_steps=cfg.sampling.n_steps_each, probability_flow=False, continuous=True,
                                      denoise=True, eps=1e-5, device=dev)
    torch.manual_seed(1234 + rank)
    x_T = sde.prior_sampling((B, 3, R, R))
    sampler(model, x_init=x_T, max_steps=0, seed=rank)     # builds the engine, loads the state
    eng = sampler.engine
    prog = eng.step_program(with_rng=True, seed=rank)
    use_graph = not args.no_graph

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    eng.run_steps(prog, args.warmup, use_graph)
    sync_all()
    t0 = time.perf_counter()
    eng.run_steps(prog, args.steps, use_graph)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    finite = bool(torch.isfinite(eng.x).all())
    images_per_sec = world * B / (args.sde_steps * ms_per_step * 1e-3)

    out = {
        "metric": "pc_sampler_images_per_sec", "value": images_per_sec, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs/ve/%s PC sampler (reverse_diffusion+langevin), "
                               "batch %d/GPU, N=%d, %dx%d; 1 step = 1 PC iteration = 2 U-Net evaluations"
                               % ("ffhq_256_ncsnpp_continuous" if args.workload == "ffhq256" else "cifar10_ncsnpp_continuous",
                                  B, args.sde_steps, R, R),
                   "batch_per_gpu": B, "sde_steps": args.sde_steps, "nfe_per_step": eng.nfe_per_step(),
                   "path": eng.last_path, "state_finite": finite,
                   "unet_gflop_per_image": eng.unet.flops_per_forward() / B / 1e9,
                   "parallelism": "replicas x%d (no collectives)" % world},
    }

    if rank == 0 and not args.no_roofline:
        # dominant kernel = conv_mfma launches of one U-Net evaluation, HIP events on the launch stream
        up = eng.unet.program
        reps = 3
        acc = np.zeros(up.n)
        up.run_timed()
        for _ in range(reps):
            acc += np.array(up.run_timed())
        ms = acc / reps
        cls = np.array(up.classes)
        fl = np.array(up.flops)
        conv3 = cls == E.FC_CONV3
        t_conv3 = float(ms[conv3].sum()) * 1e-3
        achieved = float(fl[conv3].sum()) / t_conv3 / 1e12
        by_class = {}
        for name, c in [("conv3x3_fused", E.FC_CONV3), ("conv1x1_gemm", E.FC_CONV1), ("attention", E.FC_ATTN),
                        ("groupnorm_stats", E.FC_GN), ("upfirdn", E.FC_FIR), ("other", E.FC_OTHER)]:
            m = cls == c
            by_class[name] = {"launches": int(m.sum()), "ms": float(ms[m].sum()), "gflop": float(fl[m].sum()) / 1e9}
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure is the
        # per-launch average of the committed rocprofv3 passes over this same command (tools/profile_gpu.sh ->
        # profiles/*_profile_summary.json: (2*FETCH_SIZE + WRITE_SIZE) * 1024, MI355X_MICROARCH.md HBM section)
        traffic, mfma_busy = None, None
        try:
            import glob
            summ = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_profile_summary.json")))[-1]
            with open(summ) as f:
                prof = json.load(f)
            traffic = prof["hbm_traffic"]["conv_wino_kernel"]["hbm_bytes_per_launch"]
            out["config"]["pmc_source"] = os.path.basename(summ)
        except Exception:
            pass
        wino = os.environ.get("SSDE_WINOGRAD", "1") != "0"
        out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                           "kernel": ("conv_wino_kernel (Winograd F(2x2,3x3), fp32 MFMA) + conv_mfma_kernel for the layers it does not "
                                      "take: the %d 3x3 launches of one U-Net evaluation" if wino else
                                      "conv_mfma_kernel (direct 3x3): the %d 3x3 launches of one U-Net evaluation") % int(conv3.sum()),
                           "note": "achieved = algorithmic (direct-form) FLOPs / HIP-event time; Winograd executes 2.25x fewer "
                                   "MFMA FLOPs than that, so frac is against the direct-form fp32 peak and can exceed the "
                                   "matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES in profiles/)",
                           "unet_eval_ms_eager_events": float(ms.sum()), "by_class": by_class}
        if args.dump_ops:
            rows = []
            for i in range(up.n):
                op = up.ops[i]
                row = {"i": i, "kind": int(op.kind), "cls": int(cls[i]), "ms": float(ms[i]), "gflop": float(fl[i]) / 1e9}
                if op.kind == 1:
                    c = op.u.conv
                    row.update(h=c.h_out, w=c.w_out, cout=c.c_out, cin=c.main.c0 + c.main.c1, caux=c.aux.c0 + c.aux.c1,
                               ks=c.ksize, stride=c.stride, pro=c.main.pro_mode)
                    row["tflops"] = row["gflop"] / max(row["ms"], 1e-9)
                rows.append(row)
            with open(args.dump_ops, "w") as f:
                json.dump(rows, f)
        out["config"]["end_to_end_tflops"] = eng.nfe_per_step() * eng.unet.flops_per_forward() / (ms_per_step * 1e-3) / 1e12

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import sampler_oracle
        # bounded sample: oneDNN scales badly past a few dozen threads on small convolutions (a 256-thread run
        # of this sample took minutes), so the port is timed on a fixed, stated number of host threads
        cores = max(1, min(args.cpu_threads, os.cpu_count()))
        torch.set_num_threads(cores)
        full_sd = {k: v.cpu() for k, v in sd.items()}
        full_sd["sigmas"] = model.sigmas.cpu()
        g = torch.Generator().manual_seed(3)
        kw = dict(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=args.sde_steps)

        def cpu_iter(cb, steps):
            x0 = torch.randn(cb, 3, R, R, generator=g) * cfg.model.sigma_max
            nz = torch.randn(steps, 2, cb, 3, R, R, generator=g)
            t0 = time.perf_counter()
            sampler_oracle.pc_sample(cfg, full_sd, "vesde", kw, x0, nz, snr=cfg.sampling.snr, eps=1e-5, max_steps=steps)
            return (time.perf_counter() - t0) / steps
        t_probe = cpu_iter(2, 1)                       # warm-up + per-image cost probe
        cb = int(max(2, min(args.cpu_batch, 20.0 / max(t_probe / 2, 1e-3))))
        t_cpu = cpu_iter(cb, 1)
        out["cpu_baseline"] = {"value": cb / (args.sde_steps * t_cpu), "unit": "images/s", "cores": cores, "kind": "port",
                               "sample": "1 PC iteration (2 U-Net evaluations) at batch %d with the torch-CPU oracle "
                                         "(oracle/sampler_oracle.py) on %d threads, extrapolated to N=%d: %.2f s per iteration"
                                         % (cb, cores, args.sde_steps, t_cpu)}

    if not args.no_train:
        # second headline quantity of BASELINE.json's metric ("... + sec/train-step"): reported inside the same JSON line
        del sampler, eng, prog
        torch.cuda.empty_cache()
        tr = bench_train(args, _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous"), dev, dist, world, rank, sync_all)
        out["train"] = tr

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
