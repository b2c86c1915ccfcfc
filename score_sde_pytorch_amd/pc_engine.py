"""Fused predictor-corrector sampling loop: one hipGraph replay per PC iteration.

Lowers the body of `pc_sampler`'s loop (reference sampling.py:403-407) for the stock
predictors / correctors / SDEs:

    FILL   cond   <- label_tab[step]              vec_t -> labels (models/utils.py:147-173)
    [ U-Net program ]                              score_fn(x, t)         (corrector, sampling.py:274)
    RANDN  z_c                                     rocRAND Philox4x32-10  (torch.randn_like, :275)
    SUMSQ  ||score_n||^2, ||z_n||^2                per-sample norms       (:276-277)
    LANGEVIN  batch-mean norms -> step; x_mean, x  (:278-280)
    [ U-Net program ]                              score_fn(x, t)         (predictor, sde_lib.py:105)
    RANDN  z_p
    PREDICTOR x_mean = a x + b score ; x = x_mean + c z   (sampling.py:181-187 / 195-200)
    STEP_INC

Controllable generation (controllable_generation.py:44-52, 136-144) adds, after the corrector block and after the
predictor block,    RANDN z ; PROJECT x, x_mean <- data-consistency projection (ssde_project_update).

Every per-step scalar (sigma_i, G_i, alpha_i, std_i, dt terms) is precomputed on the host with
the same fp32 torch expressions the reference evaluates, stored in device tables, and indexed by
a device-resident step counter -- the captured graph is replayed with identical arguments.
The Langevin step size needs a batch-wide reduction between the score evaluation and the update
(SURVEY F10), which is why the loop body is a kernel sequence rather than one persistent kernel.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from . import engine as E
from . import sde_lib


def plan_fused(sde, predictor, corrector, model, continuous, x, probability_flow=False):
    """Return a lowering plan when (sde, predictor, corrector, model) are all stock, else None."""
    from . import sampling as S
    from .models.ncsnpp import NCSNpp
    if not isinstance(model, NCSNpp) or not x.is_cuda:
        return None
    if type(sde) not in (sde_lib.VESDE, sde_lib.VPSDE, sde_lib.subVPSDE):
        return None
    pred = {None: "none", S.NonePredictor: "none", S.ReverseDiffusionPredictor: "reverse_diffusion",
            S.EulerMaruyamaPredictor: "euler_maruyama", S.AncestralSamplingPredictor: "ancestral_sampling"}.get(predictor, "?")
    corr = {None: "none", S.NoneCorrector: "none", S.LangevinCorrector: "langevin",
            S.AnnealedLangevinDynamics: "ald"}.get(corrector, "?")
    if pred == "?" or corr == "?":
        return None
    if corr in ("langevin", "ald") and type(sde) is sde_lib.subVPSDE:
        return None    # the reference raises AttributeError (subVPSDE has no `alphas`); keep that behaviour
    if pred == "ancestral_sampling" and (type(sde) is sde_lib.subVPSDE or probability_flow):
        return None    # NotImplementedError / AssertionError from the predictor's constructor (sampling.py:206-210)
    vp_like = type(sde) in (sde_lib.VPSDE, sde_lib.subVPSDE)
    if vp_like and model.config.model.scale_by_sigma:
        return None
    return dict(predictor=pred, corrector=corr, vp_like=vp_like, continuous=continuous)


def step_tables(sde, plan, eps, probability_flow):
    """Per-step scalars, computed on the CPU in fp32 with the reference's own expression order."""
    N = sde.N
    ts = torch.linspace(sde.T, eps, N)                                    # sampling.py:401
    zeros = torch.zeros(N, 1, 1, 1)
    ones = torch.ones(N, 1, 1, 1)
    tabs = {"t": ts}
    if plan["vp_like"]:
        if plan["continuous"] or type(sde) is sde_lib.subVPSDE:            # models/utils.py:147-153
            tabs["label"] = ts * 999
            tabs["std"] = sde.marginal_prob(zeros, ts)[1]
        else:                                                              # models/utils.py:154-158
            lab = ts * (N - 1)
            tabs["label"] = lab
            tabs["std"] = sde.sqrt_1m_alphas_cumprod[lab.long()]
    else:
        if plan["continuous"]:                                             # models/utils.py:165-166
            tabs["label"] = sde.marginal_prob(zeros, ts)[1]
        else:                                                              # models/utils.py:168-171
            tabs["label"] = torch.round((sde.T - ts) * (N - 1)).long().float()
    half = 0.5 if probability_flow else 1.0
    if plan["predictor"] == "reverse_diffusion":                           # sampling.py:195-200, sde_lib.py:102-107
        f, G = sde.discretize(ones, ts)
        f_coef = f[:, 0, 0, 0]
        a = 1.0 - f_coef
        b = G ** 2 * half
        c = torch.zeros_like(G) if probability_flow else G
        tabs["coef"] = torch.stack([a, b, c], dim=1)
    elif plan["predictor"] == "euler_maruyama":                            # sampling.py:181-187, sde_lib.py:93-100
        dt = -1. / N
        drift, diffusion = sde.sde(ones, ts)
        d_coef = drift[:, 0, 0, 0]
        a = 1.0 + d_coef * dt
        b = -(diffusion ** 2) * half * dt
        c = torch.zeros_like(diffusion) if probability_flow else diffusion * np.sqrt(-dt)
        tabs["coef"] = torch.stack([a, b, c], dim=1)
    elif plan["predictor"] == "ancestral_sampling":                        # sampling.py:213-239
        idx = (ts * (N - 1) / sde.T).long()
        if plan["vp_like"]:
            beta = sde.discrete_betas[idx]
            a = 1.0 / torch.sqrt(1. - beta)
            b = beta / torch.sqrt(1. - beta)
            c = torch.sqrt(beta)
        else:
            sigma = sde.discrete_sigmas[idx]
            adj = torch.where(idx == 0, torch.zeros_like(ts), sde.discrete_sigmas[idx - 1])
            a = torch.ones_like(sigma)
            b = sigma ** 2 - adj ** 2
            c = torch.sqrt((adj ** 2 * (sigma ** 2 - adj ** 2)) / (sigma ** 2))
        tabs["coef"] = torch.stack([a, b, c], dim=1)
    if plan["corrector"] in ("langevin", "ald") and plan["vp_like"]:       # sampling.py:267-269, 306-309
        tabs["alpha"] = sde.alphas[(ts * (N - 1) / sde.T).long()]
    if plan["corrector"] == "ald":                                         # sampling.py:311,316: step = (snr * std)^2 * 2 alpha
        # the Langevin kernel forms (snr * mean||z|| / mean||g||)^2 * 2 alpha from per-sample squared norms; feeding it
        # ||z||^2 := std^2 and ||g||^2 := 1 makes it evaluate the annealed step without a second kernel
        tabs["ald_std2"] = sde.marginal_prob(zeros, ts)[1] ** 2
        tabs["ald_one"] = torch.ones(N)
    return {k: v.to(torch.float32).contiguous() for k, v in tabs.items()}


class FusedPCSampler:
    def __init__(self, model, sde, plan, shape, snr, n_steps, probability_flow, eps, device, projection=None):
        if n_steps > 8:
            raise NotImplementedError("fused PC sampler supports n_steps <= 8 corrector steps")
        self.model, self.sde, self.plan, self.shape = model, sde, plan, tuple(shape)
        self.n_steps, self.device = n_steps, device
        B, Cc, H, W = self.shape
        self.unet = E.UNetEngine(model, B, H, W, device, vp_score=plan["vp_like"])
        tabs = step_tables(sde, plan, eps, probability_flow)
        self.tabs = {k: v.to(device) for k, v in tabs.items()}
        per = Cc * H * W
        self.x = self.unet.x_in.tensor[: B * per]
        self.x_mean = torch.zeros(B * per, device=device)
        self.z_c = torch.zeros(B * per, device=device)
        self.z_p = torch.zeros(B * per, device=device)
        self.gss = torch.zeros(B, device=device)
        self.zss = torch.zeros(B, device=device)
        self.step = torch.zeros(1, dtype=torch.int32, device=device)
        # Philox seed word read by every RANDN op at run time: a fresh value per call, ONE program / captured graph
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=device)
        self.snr, self.B, self.per = float(snr), B, per
        # projection: None, or dict(M=[9 floats] | None, invM=...) -- the inpainting / colorization data-consistency step
        self.projection = projection
        if projection is not None:
            ts = tabs["t"]
            mean1, std = sde.marginal_prob(torch.ones(sde.N, 1, 1, 1), ts)         # mean is linear in the data
            self.tabs["proj"] = torch.stack([mean1.reshape(-1), std], dim=1).to(torch.float32).contiguous().to(device)
            self.proj_data = torch.zeros(B * per, device=device)
            self.proj_mask = torch.zeros(B * per, device=device)
            self.z_pc = torch.zeros(B * per, device=device)
            self.z_pp = torch.zeros(B * per, device=device)
        self._programs = {}
        self.last_path = None
        self._stream = None

    # -------------------------------------------------------------- program assembly
    def _assemble(self, with_rng):
        unet_ops = [self.unet.program.ops[i] for i in range(self.unet.program.n)]
        unet_cls, unet_fl = list(self.unet.program.classes), list(self.unet.program.flops)
        ops, classes, flops = [], [], []

        def emit(kind, struct_cls, **fields):
            a = struct_cls()
            for k, v in fields.items():
                if isinstance(v, (list, tuple)):
                    v = (C.c_float * len(v))(*v)
                setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
            ops.append(L.make_op(kind, a)); classes.append(E.FC_OTHER); flops.append(0.0)

        def emit_projection(noise, stream_id):
            pj = self.projection
            if pj is None:
                return
            if with_rng:
                emit(L.OP_RANDN, L.RandnArgs, dst=noise, numel=self.B * self.per, seed=0, seed_ptr=self.seed_word, step_ptr=self.step, stream_id=stream_id)
            Cc = self.shape[1]
            emit(L.OP_PROJECT, L.ProjectArgs, x=self.x, x_mean=self.x_mean, data=self.proj_data, mask=self.proj_mask, noise=noise,
                 coef=self.tabs["proj"], step_ptr=self.step, n=self.B, c=Cc, hw=self.per // Cc,
                 use_matrix=int(pj.get("M") is not None), M=list(pj.get("M") or [0.0] * 9), invM=list(pj.get("invM") or [0.0] * 9))

        # Every evaluation of an iteration runs at the SAME noise level (the label is filled once, above all of them;
        # sampling.py:403-407: corrector and predictor both take vec_t): the conditioning chain of the network -- embedding, the
        # two Linear layers, the batched Dense_0 projections, ~0.2 ms of skinny GEMMs at batch 256 -- is evaluated by the first
        # one only (UNetEngine.cond_only_ops; the same values, bit for bit) when SSDE_PC_SHARE_COND=1.  Off by default: the bench's
        # headline counts two FULL evaluations per iteration, as the reference runs them.
        share = os.environ.get("SSDE_PC_SHARE_COND", "0") == "1"
        n_cond = self.unet.cond_only_ops if share else 0
        emitted = [0]

        def emit_unet():
            lo = n_cond if emitted[0] else 0
            emitted[0] += 1
            ops.extend(unet_ops[lo:]); classes.extend(unet_cls[lo:]); flops.extend(unet_fl[lo:])

        score = self.unet.out.tensor
        emit(L.OP_FILL, L.FillArgs, dst=self.unet.cond.tensor, tab=self.tabs["label"], step_ptr=self.step, n=self.B)
        if self.plan["vp_like"]:
            emit(L.OP_FILL, L.FillArgs, dst=self.unet.std.tensor, tab=self.tabs["std"], step_ptr=self.step, n=self.B)
        if self.plan["corrector"] == "ald":
            emit(L.OP_FILL, L.FillArgs, dst=self.gss, tab=self.tabs["ald_one"], step_ptr=self.step, n=self.B)
            emit(L.OP_FILL, L.FillArgs, dst=self.zss, tab=self.tabs["ald_std2"], step_ptr=self.step, n=self.B)
        if self.plan["corrector"] in ("langevin", "ald"):
            for k in range(self.n_steps):
                emit_unet()
                if with_rng:
                    emit(L.OP_RANDN, L.RandnArgs, dst=self.z_c, numel=self.B * self.per, seed=0, seed_ptr=self.seed_word, step_ptr=self.step, stream_id=k)
                if self.plan["corrector"] == "langevin":
                    emit(L.OP_SUMSQ, L.SumsqArgs, a=score, b=self.z_c, out_a=self.gss, out_b=self.zss, n=self.B, per=self.per)
                emit(L.OP_LANGEVIN, L.LangevinArgs, x=self.x, x_mean=self.x_mean, grad=score, noise=self.z_c,
                     grad_sumsq=self.gss, noise_sumsq=self.zss, alpha_tab=self.tabs.get("alpha"), step_ptr=self.step,
                     n=self.B, per=self.per, snr=self.snr)
        emit_projection(self.z_pc if self.projection is not None else None, 9)
        if self.plan["predictor"] != "none":
            emit_unet()
            if with_rng:
                emit(L.OP_RANDN, L.RandnArgs, dst=self.z_p, numel=self.B * self.per, seed=0, seed_ptr=self.seed_word, step_ptr=self.step, stream_id=8)
            emit(L.OP_PREDICTOR, L.PredictorArgs, x=self.x, x_mean=self.x_mean, score=score, noise=self.z_p,
                 coef=self.tabs["coef"], step_ptr=self.step, numel=self.B * self.per)
        emit_projection(self.z_pp if self.projection is not None else None, 10)
        emit(L.OP_STEP_INC, L.StepIncArgs, step_ptr=self.step, delta=1)
        return E.Program(L.op_array(ops), classes, flops, self)

    def step_program(self, with_rng=True):
        if with_rng not in self._programs:
            self._programs[with_rng] = self._assemble(with_rng)
        return self._programs[with_rng]

    def set_seed(self, seed=None):
        """seed=None: a fresh 62-bit word from torch's default generator, i.e. new noise on every call exactly like the
        reference's torch.randn_like (sampling.py:197,275), reproducible under torch.manual_seed and different across ranks
        that seed differently; an explicit integer pins the noise (tests).  Returns the value used."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self.seed_word.fill_(int(seed))
        self.last_seed = int(seed)
        return self.last_seed

    def nfe_per_step(self):
        return (self.n_steps if self.plan["corrector"] != "none" else 0) + (1 if self.plan["predictor"] != "none" else 0)

    # -------------------------------------------------------------- execution
    def set_projection_inputs(self, data, mask):
        """`data` (already in the projection's space) and the 0/1 `mask` of known entries, both shaped like the state."""
        self.proj_data.copy_(data.reshape(-1).to(torch.float32))
        self.proj_mask.copy_(mask.reshape(-1).to(torch.float32))

    def reset(self, x):
        self.x.copy_(x.reshape(-1).to(torch.float32))
        self.x_mean.copy_(self.x)
        self.step.zero_()

    def run(self, x, noises=None, seed=None, use_graph=True, max_steps=None):
        """Run the loop from state `x`; returns (x, x_mean) clones shaped like `shape`."""
        self.unet.weights.refresh()
        self.reset(x)
        steps = self.sde.N if max_steps is None else int(max_steps)
        if noises is not None:
            prog = self.step_program(with_rng=False)
            nz = noises.to(self.device, torch.float32)
            for i in range(steps):
                self.z_c.copy_(nz[i, 0].reshape(-1))
                self.z_p.copy_(nz[i, 1].reshape(-1))
                if self.projection is not None:
                    self.z_pc.copy_(nz[i, 2].reshape(-1))
                    self.z_pp.copy_(nz[i, 3].reshape(-1))
                prog.run()
            self.last_path = "fused-eager"
        else:
            prog = self.step_program(with_rng=True)
            self.set_seed(seed)
            self.run_steps(prog, steps, use_graph)
        return self.x.clone().view(self.shape), self.x_mean.clone().view(self.shape)

    def run_steps(self, prog, steps, use_graph=True):
        if not use_graph:
            for _ in range(steps):
                prog.run()
            self.last_path = "fused-eager"
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        s = self._stream
        s.wait_stream(torch.cuda.current_stream())
        if prog._graph is None:
            prog.capture(s)
        for _ in range(steps):
            prog.replay(s)
        torch.cuda.current_stream().wait_stream(s)
        self.last_path = "fused-graph"
