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
    """scipy.integrate._ivp.common.select_initial_step with order = 4."""
    scale = atol + torch.abs(y0) * rtol
    d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    y1 = y0 + h0 * direction * f0
    f1 = fun(t0 + h0 * direction, y1)
    d2 = _rms((f1 - f0) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1 / 5)
    return min(100 * h0, h1)


def solve_rk45(fun, t_span, y0, rtol=1e-5, atol=1e-5):
    """Integrate dy/dt = fun(t, y) from t_span[0] to t_span[1]; returns (y_final, nfev)."""
    t, t_bound = float(t_span[0]), float(t_span[1])
    direction = 1.0 if t_bound >= t else -1.0
    y = y0.to(torch.float64)
    f = fun(t, y)
    nfev = 1
    h_abs = _initial_step(fun, t, y, f, direction, rtol, atol)
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
            K = [f]
            for s_ in range(1, 6):
                dy = K[0] * (_A[s_][0] * h)
                for j in range(1, s_):
                    dy = dy + K[j] * (_A[s_][j] * h)
                K.append(fun(t + _C[s_] * h, y + dy))
            y_new = y
            for j in range(6):
                if _B[j] != 0.0:
                    y_new = y_new + K[j] * (_B[j] * h)
            f_new = fun(t + h, y_new)
            K.append(f_new)
            nfev += 6
            err = K[0] * _E[0]
            for j in range(1, 7):
                if _E[j] != 0.0:
                    err = err + K[j] * _E[j]
            scale = atol + torch.maximum(torch.abs(y), torch.abs(y_new)) * rtol
            error_norm = _rms(err * h / scale)
            if error_norm < 1:
                factor = MAX_FACTOR if error_norm == 0 else min(MAX_FACTOR, SAFETY * error_norm ** -0.2)
                if rejected:
                    factor = min(1.0, factor)
                h_abs *= factor
                break
            h_abs *= max(MIN_FACTOR, SAFETY * error_norm ** -0.2)
            rejected = True
        t, y, f = t_new, y_new, f_new
    return y, nfev


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
    """Device RK45 when the state lives on the GPU and nothing asks for the host path, else scipy on the host."""
    import os
    if method == "RK45" and y0.is_cuda and os.environ.get("SSDE_HOST_ODE", "0") != "1":
        return solve_rk45(fun, t_span, y0, rtol=rtol, atol=atol)
    return solve_host(fun, t_span, y0, rtol=rtol, atol=atol, method=method)

