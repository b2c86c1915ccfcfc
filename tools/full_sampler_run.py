#!/usr/bin/env python
"""One COMPLETE PC-sampler run through the reference's entry point (sampling.get_sampling_fn, config ve/cifar10_ncsnpp_continuous:
N = 1000 iterations, 2000 NFE) at batch 256 with seeded random weights: wall-clock images/s of the whole call (engine
lowering, hipGraph capture, 1000 replays, denoising step, inverse scaler) -- a check of bench.py's steady-state number."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402
from score_sde_pytorch_amd import sde_lib, sampling  # noqa: E402
from score_sde_pytorch_amd.models import utils as mutils  # noqa: E402

cfg = _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous")
cfg.eval.batch_size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
model = mutils.get_model("ncsnpp")(cfg)
_util.load_seeded(model, seed=1)
model = model.cuda().eval()
sde = sde_lib.VESDE(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
shape = (cfg.eval.batch_size, 3, cfg.data.image_size, cfg.data.image_size)
fn = sampling.get_sampling_fn(cfg, sde, shape, lambda x: x, 1e-5)
torch.cuda.synchronize()
t0 = time.perf_counter()
x, nfe = fn(model)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("samples %s finite=%s nfe=%d  wall %.2f s  -> %.3f images/s (path %s)  |x| max %.3g" %
      (tuple(x.shape), bool(torch.isfinite(x).all()), nfe, dt, shape[0] / dt, getattr(fn, "last_path", "?"), float(x.abs().max())))
