"""On-device adaptive Dormand-Prince RK45.

The reference drives its probability-flow ODE sampler and likelihood with `scipy.integrate.solve_ivp(method='RK45')`
(sampling.py:473, likelihood.py:99): every function evaluation converts the fp64 numpy state to an fp32 device tensor
and back (models/utils.py:181-188).  This is the same algorithm -- scipy's `RK45` class: Dormand-Prince 5(4) pair,
FSAL, RMS error norm over the whole state, step factor 0.9 * err^(-1/5) clamped to [0.2, 10], scipy's
`select_initial_step` -- with the state kept as an fp64 tensor on the GPU, so only one scalar (the error norm) crosses
the PCIe bus per step.  `fun(t, y)` receives and returns fp64 device tensors.
"""
import math

import torch

_C = [0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0]
_A = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
      [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
_B = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]
_E = [-71 / 57600, 0.0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40]
SAFETY, MIN_FACTOR, MAX_FACTOR = 0.9, 0.2, 10.0


def _rms(x):
    return float(torch.sqrt(torch.mean(x * x)))


def _initial_step(fun, t0, y0, f0, direction, rtol, atol):
    """scipy.integrate._ivp.common.select_initial_step with order = 4 (runs once per solve)."""
    scale = atol + torch.abs(y0) * rtol
    d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    y1 = y0 + h0 * direction * f0
    f1 = fun(t0 + h0 * direction, y1)
    d2 = _rms((f1 - f0) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1 / 5)
    return min(100 * h0, h1)


class _TorchStages:
    """Stage arithmetic with torch ops.  Reached only with host tensors (the CPU unit tests of the step-size controller
    against scipy on analytic right-hand sides): the model's right-hand side cannot run there."""

    def __init__(self, n, like):
        self.K = torch.empty(7, n, dtype=torch.float64, device=like.device)
        self.x32 = None

    def combine(self, y, coefs, dst):
        acc = None
        for j, c in enumerate(coefs):
            if c != 0.0:
                acc = self.K[j] * c if acc is None else acc + self.K[j] * c
        dst.copy_(y if acc is None else y + acc)

    def error_norm(self, y, y_new, coefs, atol, rtol):
        err = None
        for j, c in enumerate(coefs):
            if c != 0.0:
                err = self.K[j] * c if err is None else err + self.K[j] * c
        scale = atol + torch.maximum(torch.abs(y), torch.abs(y_new)) * rtol
        return _rms(err / scale)


class _HipStages:
    """Stage arithmetic on the device (libssde_hip: ssde_rk_combine, ssde_rk_error_norm): one launch forms a stage
    argument together with its fp32 copy -- written straight into the U-Net's input buffer when the right-hand side is
    the fused drift (FusedDrift) -- and the error norm comes back as ONE scalar per step."""

    def __init__(self, n, like, x32=None, n32=0):
        from . import _lib as L
        self.L, self.lib, self.n, self.n32 = L, L.load(), n, int(n32)
        self.K = torch.empty(7, n, dtype=torch.float64, device=like.device)
        self.partial = torch.empty(1024, dtype=torch.float64, device=like.device)
        self.out = torch.empty(1, dtype=torch.float64, device=like.device)
        self.x32 = x32

    def _stream(self):
        from . import hipops
        return hipops._stream()

    def combine(self, y, coefs, dst):
        import ctypes as C
        a = self.L.RkCombineArgs()
        terms = max([j + 1 for j, c in enumerate(coefs) if c != 0.0], default=0)
        a.y, a.k, a.n, a.terms, a.dst = y.data_ptr(), self.K.data_ptr(), self.n, terms, dst.data_ptr()
        a.dst32 = self.x32.data_ptr() if self.x32 is not None else None
        a.n32 = self.n32
        for j in range(terms):
            a.coef[j] = coefs[j]
        self.L.check(self.lib.ssde_rk_combine(C.byref(a), self._stream()), "ssde_rk_combine")

    def error_norm(self, y, y_new, coefs, atol, rtol):
        import ctypes as C
        a = self.L.RkErrorArgs()
        a.y, a.y_new, a.k, a.n, a.atol, a.rtol = y.data_ptr(), y_new.data_ptr(), self.K.data_ptr(), self.n, atol, rtol
        a.partial, a.partial_len, a.out = self.partial.data_ptr(), self.partial.numel(), self.out.data_ptr()
        for j in range(7):
            a.coef[j] = coefs[j]
        self.L.check(self.lib.ssde_rk_error_norm(C.byref(a), self._stream()), "ssde_rk_error_norm")
        return float(self.out.item())          # the one host read of the step


def solve_rk45(fun, t_span, y0, rtol=1e-5, atol=1e-5, stages=None):
    """Integrate dy/dt = fun(t, y) from t_span[0] to t_span[1]; returns (y_final, nfev).

    `fun(t, y)` returns the fp64 slope, or -- for right-hand sides that write their result in place (FusedDrift) --
    `fun(t, y, out=K_row)` fills `out`.  `stages`: the arithmetic backend (device kernels for CUDA tensors)."""
    t, t_bound = float(t_span[0]), float(t_span[1])
    direction = 1.0 if t_bound >= t else -1.0
    y = y0.to(torch.float64).clone()
    n = y.numel()
    if stages is None:
        stages = _HipStages(n, y) if y.is_cuda else _TorchStages(n, y)
    K = stages.K
    in_place = getattr(fun, "writes_out", False)

    def evaluate(tt, yy, row):
        if in_place:
            fun(tt, yy, out=K[row])
        else:
            K[row].copy_(fun(tt, yy))
    y_stage, y_new = torch.empty_like(y), torch.empty_like(y)
    stages.combine(y, [], y_stage)                       # stage argument of the first evaluation (and its fp32 copy)
    evaluate(t, y_stage, 0)
    nfev = 1
    h_abs = _initial_step((lambda tt, yy: _once(fun, stages, tt, yy, in_place)), t, y, K[0].clone(), direction, rtol, atol)
    nfev += 1
    while direction * (t - t_bound) < 0:
        min_step = 10 * abs(math.nextafter(t, direction * math.inf) - t)
        h_abs = max(h_abs, min_step)
        rejected = False
        while True:
            if h_abs < min_step:
                raise RuntimeError("solve_rk45: step size underflow (scipy: 'Required step size is less than spacing')")
            h = h_abs * direction
            t_new = t + h
            if direction * (t_new - t_bound) > 0:
                t_new = t_bound
            h = t_new - t
            h_abs = abs(h)
            for s_ in range(1, 6):
                stages.combine(y, [a * h for a in _A[s_]], y_stage)
                evaluate(t + _C[s_] * h, y_stage, s_)
            stages.combine(y, [b * h for b in _B], y_new)
            if stages.x32 is not None:                   # the fp32 copy must hold y_new for the seventh evaluation
                pass                                     # (combine wrote it: dst32 accompanies every combine)
            evaluate(t + h, y_new, 6)
            nfev += 6
            error_norm = stages.error_norm(y, y_new, [e * h for e in _E], atol, rtol)
            if error_norm < 1:
                factor = MAX_FACTOR if error_norm == 0 else min(MAX_FACTOR, SAFETY * error_norm ** -0.2)
                if rejected:
                    factor = min(1.0, factor)
                h_abs *= factor
                break
            h_abs *= max(MIN_FACTOR, SAFETY * error_norm ** -0.2)
            rejected = True
        t = t_new
        y, y_new = y_new, y                              # accept: swap buffers
        K[0].copy_(K[6])                                 # FSAL: the last slope is the next step's first
    return y, nfev


def _once(fun, stages, t, yy, in_place):
    """One extra evaluation outside the stage table (scipy's select_initial_step probes f(t0 + h0, y0 + h0 f0))."""
    if not in_place:
        return fun(t, yy)
    out = torch.empty_like(yy)
    tmp = torch.empty_like(yy)
    stages.combine(yy, [], tmp)                          # refreshes the fp32 copy the fused right-hand side reads
    fun(t, tmp, out=out)
    return out


def rhs_cache_get(model, key, make, limit=None):
    """The fused right-hand sides of a model (lowered program + arena + packed weights + hipGraph each), newest last.
    Bounded PER KIND (key[0]: "drift" of the ODE sampler, "likelihood"): building samplers / likelihood closures with fresh SDE
    objects or varying batch shapes evicts the oldest entry of that kind (its graph and arena are released with it) instead of
    growing until the device runs out of memory -- and a loop that alternates sampling and likelihood evaluation over a few
    shapes does not make the two kinds evict each other (every eviction is a re-lowering and a re-capture on the next call).
    limit: entries kept per kind (default 4; SSDE_ODE_RHS_CACHE=<n> overrides)."""
    import os
    if limit is None:
        limit = max(1, int(os.environ.get("SSDE_ODE_RHS_CACHE", "4")))
    cache = model.__dict__.setdefault("_ode_rhs", {})
    rhs = cache.pop(key, None)
    fresh = rhs is None
    if fresh:
        rhs = make()
    cache[key] = rhs                                   # (re-)inserted as the most recently used
    kind = key[0] if isinstance(key, tuple) and key else None
    same = [k for k in cache if (k[0] if isinstance(k, tuple) and k else None) == kind]     # oldest first
    for k in same[:max(0, len(same) - limit)]:
        cache.pop(k)
    return rhs, fresh


class _FusedRhs:
    """Common part of the fused right-hand sides: the per-evaluation scalars live in a 24-byte DEVICE record
    (include/ssde.h: ssde_ode_dyn -- label, std, drift coefficient, g^2, the slope row to fill), uploaded before every
    evaluation, and every launch of an evaluation is an op of ONE program that reads them from there.  On the GPU that
    program is captured into a hipGraph once and replayed per evaluation (SSDE_ODE_GRAPH=0: launched op by op): an
    adaptive solve is ~500 evaluations of ~230 (sampler) to ~1100 (likelihood) launches each."""
    writes_out = True
    _RING = 32

    def _init_dyn(self, device):
        import os
        self.device = torch.device(device)
        self.dyn = torch.zeros(24, dtype=torch.uint8, device=self.device)
        on_gpu = self.device.type == "cuda"
        self._host = [torch.zeros(24, dtype=torch.uint8).pin_memory() if on_gpu else torch.zeros(24, dtype=torch.uint8)
                      for _ in range(self._RING if on_gpu else 1)]
        self._slot = 0
        self._slot_events = [None] * len(self._host)     # the copy out of a pinned slot, recorded when it was enqueued
        self.use_graph = on_gpu and os.environ.get("SSDE_ODE_GRAPH", "1") != "0"
        self.graph_stream = torch.cuda.Stream(device=self.device) if self.use_graph else None
        self.nfev = 0
        self.last_path = None

    def _scalars(self, t):
        """(label, std, a, g2) with the SDE's own fp32 torch expressions (f(x, t) is linear in x: a = f(1, t))."""
        sde = self.sde
        tv = torch.full((1,), float(t), dtype=torch.float32)
        one = torch.ones(1, 1, 1, 1)
        drift1, diffusion = sde.sde(one, tv)
        std = sde.marginal_prob(torch.zeros(1, 1, 1, 1), tv)[1]
        label = tv * 999 if self.vp_like else std                         # models/utils.py:147-166 (continuous labels)
        eng = self.unet
        second = float(std)
        if eng.sig is not eng.cond:
            # discrete-label (positional embedding) VE model with scale_by_sigma: the output is divided by
            # sigmas[labels.long()] (ncsnpp.py:245,377-379) -- the same table lookup UNetEngine.load_inputs does; the
            # value rides in the record's `std` slot (a VE network has no std head)
            second = float(eng.model.sigmas[int(label.reshape(-1)[0])])
        return float(label), second, float(drift1.reshape(-1)[0]), float((diffusion ** 2).reshape(-1)[0])

    def _upload(self, t, out):
        import struct
        label, second, a, g2 = self._scalars(t)
        i = self._slot % len(self._host)
        h = self._host[i]
        self._slot += 1
        # a driver that enqueues more than _RING evaluations without reading anything back (a fixed-step integrator) must
        # not overwrite a record whose upload has not executed yet (as losses.FusedTrainStep's hyper ring)
        if self._slot_events[i] is not None:
            self._slot_events[i].synchronize()
        import numpy as np
        h.numpy()[:] = np.frombuffer(struct.pack("<ffffQ", label, second, a, g2, out.data_ptr()), dtype=np.uint8)
        self.dyn.copy_(h, non_blocking=True)
        if self.device.type == "cuda":
            ev = self._slot_events[i] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._slot_events[i] = ev

    def _head_ops(self, emit):
        from . import _lib as L
        eng, n = self.unet, self.shape[0]
        p = self.dyn.data_ptr()
        emit(L.OP_FILL, L.FillArgs, dst=eng.cond.tensor, tab=p, step_ptr=None, n=n)
        if eng.sig is not eng.cond:
            emit(L.OP_FILL, L.FillArgs, dst=eng.sig.tensor, tab=p + 4, step_ptr=None, n=n)
        if self.vp_like:
            emit(L.OP_FILL, L.FillArgs, dst=eng.std.tensor, tab=p + 4, step_ptr=None, n=n)

    def _build(self, assemble):
        import ctypes as C
        from . import _lib as L, engine as E
        ops = []

        def emit(kind, struct_cls, **fields):
            a = struct_cls()
            for k, v in fields.items():
                setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
            ops.append(L.make_op(kind, a))
        assemble(emit, ops)
        self.program = E.Program(L.op_array(ops), [E.FC_OTHER] * len(ops), [0.0] * len(ops), self)

    def __call__(self, t, y, out):
        self.unet.weights.refresh()
        self._upload(t, out)
        if self.use_graph:
            cur = torch.cuda.current_stream()
            s = cur if cur.cuda_stream != 0 else self.graph_stream      # (capture needs a non-default stream)
            if s is not cur:
                s.wait_stream(cur)
            if self.program._graph is None:
                self.program.capture(s)
            self.program.replay(s)
            if s is not cur:
                cur.wait_stream(s)
            self.last_path = "graph"
        else:
            self.program.run()
            self.last_path = "eager"
        self.nfev += 1

    @staticmethod
    def applies(model, sde, x):
        from . import sde_lib
        from .models.ncsnpp import NCSNpp
        if not isinstance(model, NCSNpp) or not x.is_cuda or type(sde) not in (sde_lib.VESDE, sde_lib.VPSDE, sde_lib.subVPSDE):
            return False
        return not (type(sde) is not sde_lib.VESDE and model.config.model.scale_by_sigma)


class FusedDrift(_FusedRhs):
    """Right-hand side of the probability-flow ODE for an NCSNpp model and a stock SDE, without torch arithmetic:
    drift = f(x, t) - g(t)^2 score(x, t) / 2 (sde_lib.py:93-97 with probability_flow=True; score_fn models/utils.py:129-178).
    The integrator's combine kernel writes the fp32 state straight into the U-Net program's input buffer, the program
    runs (score head included: -h / std for VP / sub-VP), ssde_pf_drift forms the fp64 slope."""

    def __init__(self, model, sde, shape, device):
        from . import engine as E, sde_lib, _lib as L
        self.sde, self.shape = sde, tuple(shape)
        self.vp_like = isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE))
        self.unet = E.UNetEngine(model, shape[0], shape[2], shape[3], device, vp_score=self.vp_like)
        self.n = int(torch.tensor(self.shape).prod())
        self.x32, self.n32 = self.unet.x_in.tensor[: self.n], self.n
        self._init_dyn(device)

        def assemble(emit, ops):
            self._head_ops(emit)
            ops.extend(self.unet.program.ops[i] for i in range(self.unet.program.n))
            emit(L.OP_PF_DRIFT, L.PfDriftArgs, x=self.x32, score=self.unet.out.tensor, dst=None, numel=self.n, a=0.0, g2=0.0,
                 dyn=self.dyn)
        self._build(assemble)


class FusedLikelihoodRhs(_FusedRhs):
    """Right-hand side of the likelihood ODE (likelihood.py:59-67): d/dt [x, delta log p] = [drift, eps^T (d drift / d x) eps]
    with the Hutchinson-Skilling probe eps fixed for the whole solve (likelihood.py:76-81).  One program per evaluation:
    labels -> U-Net forward (activations resident) -> drift -> input-gradient program with the probe as the cotangent
    (backward.TrainEngine without weight-gradient kernels) -> per-sample divergence (ssde_hutch_div).  The reference gets
    the same vector-Jacobian product from torch.autograd.grad (likelihood.py:29-35)."""

    def __init__(self, model, sde, shape, probe, device):
        from . import backward as B, sde_lib, _lib as L
        self.sde, self.shape = sde, tuple(shape)
        self.vp_like = isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE))
        eng = self.unet = B.TrainEngine(model, shape[0], shape[2], shape[3], device, vp_score=self.vp_like, input_grad=True,
                                        dropout=False, param_grads=False)
        self.n = int(torch.tensor(self.shape).prod())
        self.per = self.n // self.shape[0]
        self.x32, self.n32 = eng.x_in.tensor[: self.n], self.n
        self.eps = probe.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        eng.gout.tensor[: self.n].copy_(self.eps)        # the cotangent never changes: d(sum(score * eps)) / d score = eps
        self._init_dyn(device)

        def assemble(emit, ops):
            self._head_ops(emit)
            ops.extend(eng.program.ops[i] for i in range(eng.n_fwd))
            emit(L.OP_PF_DRIFT, L.PfDriftArgs, x=self.x32, score=eng.out.tensor, dst=None, numel=self.n, a=0.0, g2=0.0, dyn=self.dyn)
            ops.extend(eng.program.ops[i] for i in range(eng.n_fwd, eng.program.n))
            emit(L.OP_HUTCH_DIV, L.HutchDivArgs, gx=eng.gx.tensor, eps=self.eps, dst=None, dst_off=self.n, n=self.shape[0],
                 per=self.per, a=0.0, g2=0.0, dyn=self.dyn)
        self._build(assemble)

    def set_probe(self, probe):
        """a new Hutchinson probe for the next solve (the reference draws one per likelihood_fn call, likelihood.py:76-81)"""
        self.eps.copy_(probe.detach().to(self.eps.device, torch.float32).reshape(-1))
        self.unet.gout.tensor[: self.n].copy_(self.eps)


def solve_host(fun, t_span, y0, rtol=1e-5, atol=1e-5, method="RK45"):
    """The reference's integrator, scipy.integrate.solve_ivp on the host, around the same tensor right-hand side as
    solve_rk45: `fun(t, y)` takes / returns an fp64 tensor on y0's device; every evaluation crosses to numpy and back
    (what models/utils.py:181-188 does in the reference).  Used for methods other than RK45 and with SSDE_HOST_ODE=1."""
    import numpy as np
    from scipy import integrate
    dev = y0.device

    def rhs(t, y_np):
        y = torch.from_numpy(np.ascontiguousarray(y_np)).to(dev)
        return fun(float(t), y).detach().to("cpu", torch.float64).numpy()
    sol = integrate.solve_ivp(rhs, (float(t_span[0]), float(t_span[1])), y0.detach().to("cpu", torch.float64).numpy().reshape(-1),
                              rtol=rtol, atol=atol, method=method)
    return torch.from_numpy(sol.y[:, -1].copy()).to(dev), int(sol.nfev)


def integrate_ode(fun, t_span, y0, rtol, atol, method):
    """Device RK45 when the state lives on the GPU and nothing asks for the host path, else scipy on the host.
    A right-hand side with `writes_out` (FusedDrift) gets the fp32 copy of every stage argument written into its input."""
    import os
    if method == "RK45" and y0.is_cuda and os.environ.get("SSDE_HOST_ODE", "0") != "1":
        side = getattr(fun, "graph_stream", None)
        if side is None:
            stages = _HipStages(y0.numel(), y0, x32=getattr(fun, "x32", None), n32=getattr(fun, "n32", 0))
            return solve_rk45(fun, t_span, y0, rtol=rtol, atol=atol, stages=stages)
        # graph-captured right-hand side: the whole solve (stage kernels, graph replays, the one scalar read per step)
        # runs on the side stream the graph was captured on
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            stages = _HipStages(y0.numel(), y0, x32=getattr(fun, "x32", None), n32=getattr(fun, "n32", 0))
            res = solve_rk45(fun, t_span, y0, rtol=rtol, atol=atol, stages=stages)
        torch.cuda.current_stream().wait_stream(side)
        return res
    if getattr(fun, "writes_out", False):
        inner = fun

        def fun(t, y):                                   # host scipy loop around the fused right-hand side
            inner.x32.copy_(y.to(torch.float32))
            out = torch.empty_like(y)
            inner(t, y, out=out)
            return out
    return solve_host(fun, t_span, y0, rtol=rtol, atol=atol, method=method)

