"""Training losses and the one-step training function (drop-in for the reference's losses.py).

Same names, arguments and error behaviour: `get_optimizer` (:26-35), `optimization_manager` (:38-52),
`get_sde_loss_fn` (:55-101), `get_smld_loss_fn` (:104-125), `get_ddpm_loss_fn` (:128-148),
`get_step_fn` (:151-210).  Two execution paths sit behind `step_fn(state, batch)`:

* generic (any model / optimizer / optimize_fn / loss): the reference's own sequence --
  zero_grad, loss_fn, loss.backward(), optimize_fn, EMA update -- where `model(x, labels)` of our
  NCSNpp differentiates through the HIP backward program (autograd.py);
* fused (selected when the model is our NCSNpp on the GPU, the optimizer came from `get_optimizer`,
  optimize_fn from `optimization_manager`, EMA from models.ema and the loss is the continuous SDE loss):
  perturb kernel -> forward program -> DSM loss head (loss + d loss/d score in one pass) -> backward
  program -> [one RCCL all-reduce of the flat gradient] -> global-norm clip + Adam + EMA in two
  launches over flat buffers.  Random draws use torch's device generator exactly as the reference does
  (torch.rand for t, torch.randn_like for z), so a seeded run perturbs the data identically.

Data parallelism (replaces nn.DataParallel): one process per GPU; when torch.distributed is
initialised the fused step averages gradients with ONE all-reduce per step (parallel.py).
"""
import numpy as np
import torch
import torch.optim as optim

from . import sde_lib
from .models import utils as mutils
from .sde_lib import VESDE, VPSDE


class FusedAdam(optim.Adam):
    """torch.optim.Adam whose state can live in flat buffers shared with libssde_hip's fused update.

    Behaves exactly like torch.optim.Adam (step(), state_dict(), load_state_dict()); `flatten_like`
    re-homes exp_avg / exp_avg_sq into two flat fp32 buffers laid out like backward.FlatParams, after
    which both torch's own step() and the fused kernel update the same memory."""

    def flatten_like(self, flat):
        cached = getattr(self, "_ssde_flat", None)
        if cached is not None and cached[0] is flat:
            return cached[1], cached[2]
        dev = flat.data.device
        m = torch.zeros(flat.numel, dtype=torch.float32, device=dev)
        v = torch.zeros(flat.numel, dtype=torch.float32, device=dev)
        for p in flat.params:
            o, n = flat.index[id(p)]
            st = self.state[p]
            if "exp_avg" in st:
                m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = st.get("step", None)
            st["step"] = step if torch.is_tensor(step) else torch.tensor(float(step or 0.0))
            st["exp_avg"] = m[o:o + n].view(p.shape)
            st["exp_avg_sq"] = v[o:o + n].view(p.shape)
        self._ssde_flat = (flat, m, v)
        return m, v

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._ssde_flat = None          # the loaded exp_avg / exp_avg_sq tensors are re-homed on the next fused step


def get_optimizer(config, params):
    """Returns an Adam optimizer built from `config.optim` (losses.py:26-35)."""
    if config.optim.optimizer == 'Adam':
        optimizer = FusedAdam(params, lr=config.optim.lr, betas=(config.optim.beta1, 0.999), eps=config.optim.eps,
                              weight_decay=config.optim.weight_decay)
    else:
        raise NotImplementedError(f'Optimizer {config.optim.optimizer} not supported yet!')
    return optimizer


def optimization_manager(config):
    """Returns an optimize_fn based on `config` (losses.py:38-52): lr warm-up, global-norm clip, step."""

    def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup, grad_clip=config.optim.grad_clip):
        if warmup > 0:
            for g in optimizer.param_groups:
                g['lr'] = lr * np.minimum(step / warmup, 1.0)
        if grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
        optimizer.step()

    # read by the fused step (the arithmetic above, executed by ssde_adam_clip_ema)
    optimize_fn.ssde_hyper = dict(lr=config.optim.lr, warmup=config.optim.warmup, grad_clip=config.optim.grad_clip)
    return optimize_fn


class _DsmHead(torch.autograd.Function):
    """mean_n [ reduce_i (score * s + z)^2 ]  (or the likelihood-weighted form  reduce_i (score + z / s)^2 * g2)  as ONE
    libssde_hip kernel pair (ssde_dsm_loss); its gradient w.r.t. `score` is produced by the same launch.  This is the
    loss head of the three reference closures (losses.py:77-101, 104-125, 128-148) for models the fused step cannot
    lower: the model itself still runs under torch autograd, the head does not."""

    @staticmethod
    def forward(ctx, score, z, s, g2, reduce_mean, likelihood_weighting):
        from . import hipops
        loss, _, dscore = hipops.dsm_loss(score.detach().float().contiguous(), z.float().contiguous(), s.float().contiguous(),
                                          None if g2 is None else g2.float().contiguous(), reduce_mean=reduce_mean,
                                          likelihood_weighting=likelihood_weighting, want_grad=True)
        ctx.save_for_backward(dscore)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad):
        (dscore,) = ctx.saved_tensors
        return dscore * grad, None, None, None, None, None


def _perturb(x, z, s, a=None):
    """a[n] * x + s[n] * z  (ssde_perturb)."""
    from . import hipops
    return hipops.perturb(x.float().contiguous(), z.float().contiguous(), s.float().contiguous(),
                          None if a is None else a.float().contiguous())


def get_sde_loss_fn(sde, train, reduce_mean=True, continuous=True, likelihood_weighting=True, eps=1e-5):
    """Continuous-time denoising score matching loss (losses.py:55-101)."""
    def loss_fn(model, batch):
        score_fn = mutils.get_score_fn(sde, model, train=train, continuous=continuous)
        t = torch.rand(batch.shape[0], device=batch.device) * (sde.T - eps) + eps
        z = torch.randn_like(batch)
        mean, std = sde.marginal_prob(batch, t)
        score = score_fn(_perturb(mean, z, std), t)                      # mean + std[:, None, None, None] * z
        g2 = sde.sde(torch.zeros_like(batch), t)[1] ** 2 if likelihood_weighting else None
        return _DsmHead.apply(score, z, std, g2, reduce_mean, likelihood_weighting)

    loss_fn.ssde_spec = dict(kind="sde", sde=sde, train=train, reduce_mean=reduce_mean, continuous=continuous,
                             likelihood_weighting=likelihood_weighting, eps=eps)
    return loss_fn


def get_smld_loss_fn(vesde, train, reduce_mean=False):
    """Legacy discrete SMLD (NCSN) loss (losses.py:104-125)."""
    assert isinstance(vesde, VESDE), "SMLD training only works for VESDEs."
    smld_sigma_array = torch.flip(vesde.discrete_sigmas, dims=(0,))

    def loss_fn(model, batch):
        model_fn = mutils.get_model_fn(model, train=train)
        labels = torch.randint(0, vesde.N, (batch.shape[0],), device=batch.device)
        sigmas = smld_sigma_array.to(batch.device)[labels]
        z = torch.randn_like(batch)
        score = model_fn(_perturb(batch, z, sigmas), labels)             # batch + sigma * z
        # (score - target)^2 * sigma^2 with target = -z / sigma  is  (score * sigma + z)^2
        return _DsmHead.apply(score, z, sigmas, None, reduce_mean, False)

    # (score - target)^2 sigma^2 with target = -z / sigma is (score sigma + z)^2: the DSM head with std = sigma[labels]
    loss_fn.ssde_spec = dict(kind="smld", sde=vesde, train=train, reduce_mean=reduce_mean, continuous=False,
                             likelihood_weighting=False, eps=0.0)
    return loss_fn


def get_ddpm_loss_fn(vpsde, train, reduce_mean=True):
    """Legacy discrete DDPM loss (losses.py:128-148)."""
    assert isinstance(vpsde, VPSDE), "DDPM training only works for VPSDEs."

    def loss_fn(model, batch):
        model_fn = mutils.get_model_fn(model, train=train)
        labels = torch.randint(0, vpsde.N, (batch.shape[0],), device=batch.device)
        a = vpsde.sqrt_alphas_cumprod.to(batch.device)[labels]
        s = vpsde.sqrt_1m_alphas_cumprod.to(batch.device)[labels]
        z = torch.randn_like(batch)
        eps_theta = model_fn(_perturb(batch, z, s, a), labels)           # sqrt(abar) * batch + sqrt(1 - abar) * z
        # (eps_theta - z)^2 = (eps_theta * (-1) + z)^2: the same head with s = -1
        return _DsmHead.apply(eps_theta, z, -torch.ones_like(s), None, reduce_mean, False)

    # (eps_theta - z)^2 is (score std + z)^2 for the VP score head score = -eps_theta / std, std = sqrt(1 - alpha_bar)[labels]
    loss_fn.ssde_spec = dict(kind="ddpm", sde=vpsde, train=train, reduce_mean=reduce_mean, continuous=False,
                             likelihood_weighting=False, eps=0.0)
    return loss_fn


# --------------------------------------------------------------------------- fused DSM step
class FusedTrainStep:
    """perturb -> forward -> loss head -> backward -> [all-reduce] -> clip + Adam + EMA, all libssde_hip kernels."""

    def __init__(self, model, spec, batch_shape, device):
        from . import backward as B
        from . import _lib as L
        from . import engine as E
        self.L, self.E = L, E
        self.model, self.spec, self.device = model, spec, device
        sde = spec["sde"]
        self.vp_like = isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE))
        if self.vp_like and model.config.model.scale_by_sigma:
            raise NotImplementedError("scale_by_sigma with a VP score head")
        n, c, h, w = batch_shape
        self.n, self.per = n, c * h * w
        self.eng = B.TrainEngine(model, n, h, w, device, vp_score=self.vp_like, dropout=spec["train"])
        self.flat = self.eng.flat
        f32 = dict(dtype=torch.float32, device=device)
        self.z = torch.zeros(n, c, h, w, **f32)
        self.batch = torch.zeros(n, c, h, w, **f32)
        self.a = torch.zeros(n, **f32)
        self.s = torch.zeros(n, **f32)
        self.g2 = torch.zeros(n, **f32)
        self.losses = torch.zeros(n, **f32)
        self.loss = torch.zeros(1, **f32)
        self.hyper = torch.zeros(12, **f32)
        # ring of pinned staging buffers for the per-step scalars: the H2D copy is asynchronous and nothing else in the
        # step synchronises the host, so a slot is rewritten only after the copy that last read it has completed
        self._hyper_ring = [[torch.zeros(12, dtype=torch.float32, pin_memory=torch.cuda.is_available()), None]
                            for _ in range(4)]
        self._hyper_slot = 0
        self.gnorm = torch.zeros(1, **f32)
        self.partial = torch.zeros(1024, **f32)
        self._opt_prog = None
        self._head = self._build_head()
        self.steps_done = 0

    def _prog(self, entries):
        b = self.E.ProgramBuilder(self.device)
        for kind, fields in entries:
            b.add(kind, fields)
        return b.finalize()

    def _build_head(self):
        L, eng, spec = self.L, self.eng, self.spec
        x_in = eng.x_in.tensor
        perturb = self._prog([(L.OP_PERTURB, dict(x=self.batch, z=self.z, a=self.a, s=self.s, dst=x_in, n=self.n, per=self.per))])
        loss = self._prog([(L.OP_DSM_LOSS, dict(score=eng.out.tensor, z=self.z, s=self.s,
                                                g2=self.g2 if spec["likelihood_weighting"] else None,
                                                dscore=eng.gout.tensor if spec["train"] else None, losses=self.losses,
                                                loss=self.loss, n=self.n, per=self.per, reduce_mean=int(spec["reduce_mean"]),
                                                likelihood_weighting=int(spec["likelihood_weighting"]),
                                                grad_scale=1.0 / self._world()))])
        return perturb, loss

    @staticmethod
    def _world():
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _exchanges(self):
        """Is the gradient exchange on?  Data parallel (world > 1); SSDE_FORCE_GRAD_EXCHANGE=1 also turns it on in a
        one-rank process group, so that the bucketed RCCL path and its stream ordering run on a single GPU
        (tests/test_train_gpu.py).  skip_exchange: bench.py times the step without it."""
        import os
        import torch.distributed as dist
        if getattr(self, "skip_exchange", False):
            return False
        if self._world() > 1:
            return True
        return os.environ.get("SSDE_FORCE_GRAD_EXCHANGE", "0") == "1" and dist.is_available() and dist.is_initialized()

    def _optimizer_program(self, optimizer, ema):
        m, v = optimizer.flatten_like(self.flat)
        ema_buf = ema.flatten_like(self.flat) if ema is not None else None
        key = (m.data_ptr(), v.data_ptr(), ema_buf.data_ptr() if ema_buf is not None else 0)
        if self._opt_prog is None or self._opt_prog[1] is not optimizer or self._opt_prog[2] is not ema or self._opt_prog[3] != key:
            L = self.L
            prog = self._prog([
                (L.OP_SUMSQ_FLAT, dict(x=self.flat.grad, numel=self.flat.numel, partial=self.partial, out=self.gnorm)),
                (L.OP_ADAM, dict(p=self.flat.data, g=self.flat.grad, m=m, v=v, ema=ema_buf, numel=self.flat.numel,
                                 hyper=self.hyper, gnorm_sq=self.gnorm))])
            self._opt_prog = (prog, optimizer, ema, key)
        return self._opt_prog[0]

    def perturb_inputs(self, batch, t, z):
        """Per-sample coefficients with the SDE's own [B]-sized expressions; the per-pixel work is the perturb kernel."""
        sde, spec = self.spec["sde"], self.spec
        if spec["kind"] == "smld":                                  # losses.py:111-118; t holds the integer noise levels
            std = torch.flip(sde.discrete_sigmas, dims=(0,)).to(self.device)[t]
            self.a.fill_(1.0)
            labels = t.to(torch.float32)
            if self.eng.sig is not self.eng.cond:                   # positional models gather sigma for scale_by_sigma
                self.eng.sig.tensor[: self.n].copy_(self.model.sigmas.to(torch.float32)[t])
        elif spec["kind"] == "ddpm":                                # losses.py:135-141
            std = sde.sqrt_1m_alphas_cumprod.to(self.device)[t]
            self.a.copy_(sde.sqrt_alphas_cumprod.to(self.device)[t])
            labels = t.to(torch.float32)
        else:
            ones = torch.ones(self.n, 1, 1, 1, device=self.device)
            mean1, std = sde.marginal_prob(ones, t)                 # mean is linear in x: mean = a[n] * x
            self.a.copy_(mean1.reshape(self.n))
            if spec["likelihood_weighting"]:
                self.g2.copy_(sde.sde(torch.zeros(self.n, 1, 1, 1, device=self.device), t)[1] ** 2)
            labels = t * 999 if self.vp_like else std               # models/utils.py:147-166 (continuous)
        self.s.copy_(std)
        self.eng.cond.tensor[: self.n].copy_(labels)
        if self.vp_like:
            self.eng.std.tensor[: self.n].copy_(std)
        self.batch.copy_(batch)
        self.z.copy_(z)

    def _draw_and_perturb(self, batch, t, z):
        spec = self.spec
        if t is None and spec["kind"] in ("smld", "ddpm"):
            t = torch.randint(0, spec["sde"].N, (batch.shape[0],), device=batch.device)                         # losses.py:116,136
        elif t is None:
            t = torch.rand(batch.shape[0], device=batch.device) * (spec["sde"].T - spec["eps"]) + spec["eps"]   # losses.py:84
        if z is None:
            z = torch.randn_like(batch)                                                                         # losses.py:85
        self.perturb_inputs(batch, t, z)

    def loss_and_grads(self, batch, t=None, z=None, seed=None):
        """Forward + loss (+ backward when built for training); returns the device scalar loss."""
        spec = self.spec
        self._draw_and_perturb(batch, t, z)
        eng = self.eng
        eng.weights.refresh()
        eng.set_dropout_seed(self._dropout_seed() if seed is None else seed)
        self._head[0].run()
        eng.run_forward()
        self._head[1].run()
        self._pending = []
        if spec["train"]:
            if self._exchanges():
                # data parallel: ~32 MB buckets of the flat gradient are all-reduced (RCCL, its own stream) as soon as the
                # backward ops that finalise them are enqueued, overlapping the rest of the backward program
                import torch.distributed as dist
                import os
                bucket = int(float(os.environ.get("SSDE_GRAD_BUCKET_MB", "32")) * 262144)
                eng.run_backward_bucketed(
                    lambda lo, hi: self._pending.append(dist.all_reduce(self.flat.grad[lo:hi], async_op=True)), bucket)
            else:
                eng.run_backward()
        return self.loss

    # ------------------------------------------------------------------ the whole step as ONE hipGraph
    def _step_graph(self, optimizer, ema):
        """perturb -> forward -> loss head -> backward -> clip + Adam + EMA -> weight re-pack as one captured graph (no gradient
        exchange: a single replica, or bench.py's no-exchange leg).  Everything that changes from step to step lives in device
        buffers the host refreshes before the replay (batch, z, per-sample coefficients, the hyper-parameter record, the dropout
        seed word), so the ~1400 launches of a step cost one hipGraphLaunch.  SSDE_TRAIN_GRAPH=0 keeps the program runs."""
        import os
        if os.environ.get("SSDE_TRAIN_GRAPH", "1") == "0" or not self.spec["train"] or self.device.type != "cuda":
            return None
        if getattr(self, "_graph_failed", None) is not None:
            return None
        opt = self._optimizer_program(optimizer, ema)
        ws = self.eng.weights
        if getattr(ws, "_tables", None) is None or \
                ws._tables[1] != tuple(s_.data_ptr() for e in ws.entries if e[4] is not None for s_ in e[1]):
            ws._build_tables()
        key = (id(opt), id(ws._tables))
        if getattr(self, "_graph", None) is not None and self._graph[0] == key:
            return self._graph[1]
        L, E, eng = self.L, self.E, self.eng
        ops = [self._head[0].ops[0]]
        ops += [eng.program.ops[i] for i in range(eng.n_fwd)]
        ops += [self._head[1].ops[0]]
        ops += [eng.program.ops[i] for i in range(eng.n_fwd, eng.program.n)]
        ops += [opt.ops[i] for i in range(opt.n)]
        ops += [L.make_op(L.OP_PACK, args) for args, _ in ws._tables[0]]
        prog = E.Program(L.op_array(ops), [0] * len(ops), [0.0] * len(ops), (self, opt, ws._tables))
        if getattr(self, "_gstream", None) is None:
            self._gstream = torch.cuda.Stream(device=self.device)
        self._gstream.wait_stream(torch.cuda.current_stream())
        try:
            prog.capture(self._gstream)
        except L.SsdeError as exc:
            # a driver that refuses the capture (or the instantiation) must not stop training: the program runs of
            # loss_and_grads + optimizer_step are the same launches, issued one by one.  Remembered, reported once.
            import warnings
            self._graph_failed = exc
            warnings.warn("libssde_hip: capturing the training step as a hipGraph failed (%s); running it as program launches" % exc)
            torch.cuda.current_stream().wait_stream(self._gstream)
            return None
        torch.cuda.current_stream().wait_stream(self._gstream)
        self._graph = (key, prog)
        return prog

    def _step_graph_segments(self, optimizer, ema):
        """The same step with the gradient exchange (data parallel, world > 1): ONE hipGraph PER GRADIENT BUCKET instead of ~1400
        program launches.  Segment k holds the ops up to the point where bucket k of the flat gradient is final
        (TrainEngine.grad_buckets: the tail of the buffer completes first); the host replays segment k, issues the RCCL
        all-reduce of bucket k (its stream waits for the replay, the next segment's replay does not wait for it), and after
        the last bucket a closing graph holds clip + Adam + EMA + weight re-pack behind the collectives.  At the default 32 MB
        bucket that is 9 + 1 graph launches and 8 all-reduces per step.  Returns [(program, (lo, hi) or None)], or None when
        graphs are off / unavailable (the program runs of loss_and_grads + optimizer_step are the fallback)."""
        import os
        if os.environ.get("SSDE_TRAIN_GRAPH", "1") == "0" or not self.spec["train"] or self.device.type != "cuda":
            return None
        if getattr(self, "_graph_failed", None) is not None:
            return None
        opt = self._optimizer_program(optimizer, ema)
        ws = self.eng.weights
        if getattr(ws, "_tables", None) is None or \
                ws._tables[1] != tuple(s_.data_ptr() for e in ws.entries if e[4] is not None for s_ in e[1]):
            ws._build_tables()
        bucket = int(float(os.environ.get("SSDE_GRAD_BUCKET_MB", "32")) * 262144)
        key = (id(opt), id(ws._tables), bucket)
        if getattr(self, "_graph_seg", None) is not None and self._graph_seg[0] == key:
            return self._graph_seg[1]
        L, E, eng = self.L, self.E, self.eng
        head = [self._head[0].ops[0]] + [eng.program.ops[i] for i in range(eng.n_fwd)] + [self._head[1].ops[0]]
        plan, cur = [], eng.n_fwd
        for lo, hi, op_end in eng.grad_buckets(bucket):
            ops = head + [eng.program.ops[i] for i in range(cur, max(op_end, cur))]
            head, cur = [], max(op_end, cur)
            plan.append((ops, (lo, hi)))
        tail = [eng.program.ops[i] for i in range(cur, eng.program.n)]
        tail += [opt.ops[i] for i in range(opt.n)] + [L.make_op(L.OP_PACK, args) for args, _ in ws._tables[0]]
        plan.append((tail, None))
        if getattr(self, "_gstream", None) is None:
            self._gstream = torch.cuda.Stream(device=self.device)
        self._gstream.wait_stream(torch.cuda.current_stream())
        segs = []
        try:
            for ops, span in plan:
                prog = None
                if ops:                              # (a bucket that is final at the same op as its predecessor: no launch)
                    prog = E.Program(L.op_array(ops), [0] * len(ops), [0.0] * len(ops), (self, opt, ws._tables))
                    prog.capture(self._gstream)
                segs.append((prog, span))
        except L.SsdeError as exc:
            import warnings
            self._graph_failed = exc
            warnings.warn("libssde_hip: capturing the training step as hipGraphs failed (%s); running it as program launches" % exc)
            torch.cuda.current_stream().wait_stream(self._gstream)
            return None
        torch.cuda.current_stream().wait_stream(self._gstream)
        self._graph_seg = (key, segs)
        return segs

    def _train_step_exchange_graphs(self, segs, batch, optimizer, ema, step, hyper, t, z, seed):
        import torch.distributed as dist
        self._draw_and_perturb(batch, t, z)
        self.eng.weights.refresh()
        self.eng.set_dropout_seed(self._dropout_seed() if seed is None else seed)
        self._upload_hyper(optimizer, ema, step, hyper)
        s = self._gstream
        s.wait_stream(torch.cuda.current_stream())
        works = []
        with torch.cuda.stream(s):                   # (the collective's stream waits for what `s` holds when it is issued)
            for prog, span in segs:
                if span is None:
                    for w in works:
                        w.wait()                     # `s` waits for the all-reduces; the host does not
                if prog is not None:
                    prog.replay(s)
                if span is not None:
                    works.append(dist.all_reduce(self.flat.grad[span[0]:span[1]], async_op=True))
        self.collectives_last_step = len(works)
        torch.cuda.current_stream().wait_stream(s)
        self._after_update(optimizer, repacked=True)
        return self.loss.clone()

    def train_step(self, batch, optimizer, ema, step, hyper, t=None, z=None, seed=None):
        """One optimisation step (losses.py:179-208's train branch); returns the device scalar loss (of the parameters BEFORE the
        update, as the reference's step_fn does)."""
        if self._exchanges():
            segs = self._step_graph_segments(optimizer, ema)
            if segs is not None:
                return self._train_step_exchange_graphs(segs, batch, optimizer, ema, step, hyper, t, z, seed)
            prog = None
        else:
            prog = self._step_graph(optimizer, ema)
        if prog is None:
            loss = self.loss_and_grads(batch, t=t, z=z, seed=seed)
            loss = loss.clone()
            self.optimizer_step(optimizer, ema, step, hyper)
            return loss
        self._draw_and_perturb(batch, t, z)
        self.eng.weights.refresh()                    # (a no-op unless somebody wrote parameters behind the step's back)
        self.eng.set_dropout_seed(self._dropout_seed() if seed is None else seed)
        self._upload_hyper(optimizer, ema, step, hyper)
        s = self._gstream
        s.wait_stream(torch.cuda.current_stream())
        prog.replay(s)
        torch.cuda.current_stream().wait_stream(s)
        self._after_update(optimizer, repacked=True)
        return self.loss.clone()

    def _dropout_seed(self):
        """Mask stream of this step: torch's seed (torch.manual_seed), the data-parallel rank and a per-call counter
        that a resumed run continues from `state['step']` (the reference's dropout draws from torch's generator, so
        replicas and resumed runs never replay the same masks)."""
        import torch.distributed as dist
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        count = self.steps_done + int(getattr(self, "step_offset", 0))
        word = (int(torch.initial_seed()) * 1000003 + rank * 7919 + 17) & 0xFFFFFFFFFFFF
        return (word * 2654435761 + count * 40503) & 0x7FFFFFFF

    def optimizer_step(self, optimizer, ema, step, hyper):
        import torch.distributed as dist
        if self._exchanges():     # gradients were pre-scaled by 1/world in the loss head
            self.collectives_last_step = len(getattr(self, "_pending", None) or []) or 1
            if getattr(self, "_pending", None):
                for work in self._pending:           # bucketed all-reduces started during the backward program
                    work.wait()
                self._pending = []
            else:
                dist.all_reduce(self.flat.grad)
        self._upload_hyper(optimizer, ema, step, hyper)
        self._optimizer_program(optimizer, ema).run()
        self._after_update(optimizer, repacked=False)

    def _upload_hyper(self, optimizer, ema, step, hyper):
        group = optimizer.param_groups[0]
        lr = hyper["lr"]
        if hyper["warmup"] > 0:
            lr = hyper["lr"] * float(np.minimum(step / hyper["warmup"], 1.0))
            for g in optimizer.param_groups:
                g['lr'] = lr
        # torch.optim.Adam keeps a per-parameter step counter; all parameters share it here
        optimizer.flatten_like(self.flat)
        st0 = optimizer.state[self.flat.params[0]]
        t_adam = float(st0["step"]) + 1.0
        b1, b2 = group["betas"]
        decay = ema.next_decay() if ema is not None else 1.0
        slot = self._hyper_ring[self._hyper_slot % len(self._hyper_ring)]
        self._hyper_slot += 1
        if slot[1] is not None:
            slot[1].synchronize()
        h = slot[0]
        h[0], h[1], h[2], h[3], h[4] = lr, b1, b2, group["eps"], group["weight_decay"]
        h[5] = hyper["grad_clip"]
        h[6], h[7], h[8] = 1.0 - b1 ** t_adam, float(np.sqrt(1.0 - b2 ** t_adam)), 1.0 - decay
        self.hyper.copy_(h, non_blocking=True)
        if self.hyper.is_cuda:
            slot[1] = torch.cuda.Event()
            slot[1].record()

    def _after_update(self, optimizer, repacked):
        for p in self.flat.params:
            optimizer.state[p]["step"] += 1
        self.flat.touch()                            # every engine lowered from this model re-packs on its next refresh
        if repacked:
            self.eng.weights.mark_fresh()            # (this engine's packed copies were re-packed inside the step's graph)
        else:
            self.eng.weights.refresh()               # packed weight copies follow the in-place parameter update
        self.steps_done += 1


def _on_device(t):
    """Does this tensor live in HIP device memory?  (One place to ask, so that the CPU emulator of tests/emu, whose "device"
    memory is host memory, can drive step_fn itself.)"""
    return t.is_cuda


def _fused_candidate(state, loss_fn, optimize_fn, train):
    from .models.ncsnpp import NCSNpp
    from .models.ema import ExponentialMovingAverage
    model = state['model']
    if not isinstance(model, NCSNpp) or getattr(loss_fn, "ssde_spec", None) is None:
        return False
    if not _on_device(next(model.parameters())):
        return False
    if not isinstance(state.get('ema'), ExponentialMovingAverage):
        return False
    if train and (not isinstance(state.get('optimizer'), FusedAdam) or getattr(optimize_fn, "ssde_hyper", None) is None):
        return False
    if train and len(state['optimizer'].param_groups) != 1:
        return False
    return True


def get_step_fn(sde, train, optimize_fn=None, reduce_mean=False, continuous=True, likelihood_weighting=False):
    """Create a one-step training/evaluation function (losses.py:151-210)."""
    if continuous:
        loss_fn = get_sde_loss_fn(sde, train, reduce_mean=reduce_mean, continuous=True,
                                  likelihood_weighting=likelihood_weighting)
    else:
        assert not likelihood_weighting, "Likelihood weighting is not supported for original SMLD/DDPM training."
        if isinstance(sde, VESDE):
            loss_fn = get_smld_loss_fn(sde, train, reduce_mean=reduce_mean)
        elif isinstance(sde, VPSDE):
            loss_fn = get_ddpm_loss_fn(sde, train, reduce_mean=reduce_mean)
        else:
            raise ValueError(f"Discrete training for {sde.__class__.__name__} is not recommended.")
    cache = {}

    def fused_for(state, batch):
        key = (id(state['model']), tuple(batch.shape), batch.device.index)
        fs = cache.get(key)
        if fs is None:
            fs = cache[key] = FusedTrainStep(state['model'], loss_fn.ssde_spec, tuple(batch.shape), batch.device)
        return fs

    def step_fn(state, batch):
        model = state['model']
        if _on_device(batch) and _fused_candidate(state, loss_fn, optimize_fn, train):
            fs = fused_for(state, batch)
            if train:
                fs.step_offset = int(state['step']) - fs.steps_done      # a restored checkpoint continues its mask stream
                loss = fs.train_step(batch, state['optimizer'], state['ema'], state['step'], optimize_fn.ssde_hyper)
                state['step'] += 1
            else:
                ema = state['ema']
                ema.store(model.parameters())
                ema.copy_to(model.parameters())
                fs.eng.weights.refresh(force=True)
                loss = fs.loss_and_grads(batch).clone()
                ema.restore(model.parameters())
                fs.eng.weights.refresh(force=True)
            return loss.reshape(())
        # ---- generic path: the reference's sequence (losses.py:190-207)
        if train:
            optimizer = state['optimizer']
            optimizer.zero_grad()
            loss = loss_fn(model, batch)
            loss.backward()
            optimize_fn(optimizer, model.parameters(), step=state['step'])
            state['step'] += 1
            state['ema'].update(model.parameters())
        else:
            with torch.no_grad():
                ema = state['ema']
                ema.store(model.parameters())
                ema.copy_to(model.parameters())
                loss = loss_fn(model, batch)
                ema.restore(model.parameters())
        return loss

    step_fn.loss_fn = loss_fn
    step_fn.fused_for = fused_for
    return step_fn
