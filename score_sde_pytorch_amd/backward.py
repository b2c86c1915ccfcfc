"""Backward lowering: reverse-mode differentiation of a U-Net program into a program of HIP kernels.

The reference gets its backward from torch autograd over ~900 eager ops (losses.py:196
`loss.backward()`).  Here the forward is already a flat list of fused ops (engine.py); `TrainEngine`
walks that list in reverse ONCE and emits, per forward op, the kernels of its adjoint:

  conv (k x k main + 1 x 1 aux + bias + temb addend + residual, scaled)
      -> column sums            bias / Dense_0 addend gradients           (backward.hip)
         weight gradients       pixel-split MFMA GEMM, prologue recomputed (wgrad.hip)
         input gradient         the FORWARD conv kernel on the output gradient with
                                transposed + 180-degree-rotated weights    (conv_mfma.hip)
         prologue backward      GroupNorm / SiLU / dropout adjoint, fused  (backward.hip)
         residual               scaled accumulate
  upfirdn2d  -> upfirdn2d with up/down swapped, the flipped kernel and the gradient pads of
                op/upfirdn2d.py:111-116 (UpFirDn2d.backward)
  attention  -> two kernels that recompute the probabilities (attention.hip)
  NCHW/NHWC boundary ops -> each other.

Parameter gradients land in a flat fp32 buffer laid out exactly like the parameters themselves
(`FlatParams`), so `p.grad` is a view for torch users and the fused optimizer (losses.py) sees two
flat arrays.  Forward and backward share ONE liveness-planned arena: a forward activation stays
resident exactly until its last backward consumer has been enqueued.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import engine as E
from .engine import _src

FC_WGRAD, FC_BWD = 6, 7


def trainable_params(model):
    """Parameters that are trainable by design, in model.parameters() order: everything but the frozen Fourier
    frequencies (layerspp.py:38 `requires_grad=False`).  Independent of requires_grad flags a caller may toggle
    temporarily (likelihood.py freezes everything while it differentiates w.r.t. the input)."""
    frozen = {id(p) for m in model.modules() if getattr(m, "kind", "") == "fourier" for p in m.parameters(recurse=False)}
    return [p for p in model.parameters() if id(p) not in frozen]


class FlatParams:
    """All trainable parameters of a model as views into one flat fp32 buffer (+ a same-shaped gradient buffer)."""

    def __init__(self, model, device):
        # trainable-by-design parameters (the frozen Fourier frequencies are not): independent of requires_grad flags a
        # caller may toggle temporarily (likelihood.py freezes everything while it differentiates w.r.t. the input)
        params = trainable_params(model)
        # parameters the kernels treat as ONE operand (the Dense_0 projections of all residual blocks form one GEMM,
        # engine.py) are laid out back to back, so their gradient is one weight-gradient launch as well
        groups = model.flat_param_groups() if hasattr(model, "flat_param_groups") else []
        grouped = [p for g in groups for p in g]
        seen = {id(p) for p in grouped}
        self.params = grouped + [p for p in params if id(p) not in seen]
        self.model_order = params            # model.parameters() order (what optimizers / EMA objects were built from)
        self.index, off = {}, 0
        for p in self.params:
            self.index[id(p)] = (off, p.numel())
            off += (p.numel() + 3) // 4 * 4          # 16-byte aligned starts
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p in self.params:
                o, n = self.index[id(p)]
                self.data[o:o + n].copy_(p.detach().reshape(-1).to(device, torch.float32))
                p.data = self.data[o:o + n].view(p.shape)
                p._ssde_flat = self
        self.device = device
        # bumped by whoever writes `data` as a whole (fused optimizer kernel, EMA swap): per-tensor version counters do
        # not see such writes, engine.WeightStore stamps read this instead
        self.generation = 0

    def touch(self):
        self.generation += 1

    def owns(self, model):
        ps = trainable_params(model)
        if len(ps) != len(self.params):
            return False
        base = self.data.data_ptr()
        for p in ps:
            e = self.index.get(id(p))
            if e is None or p.data_ptr() != base + e[0] * 4:
                return False
        return True

    def owns_params(self, params):
        """True when every tensor of `params` still is the view of `data` this object gave it."""
        base = self.data.data_ptr()
        for p in params:
            e = self.index.get(id(p))
            if e is None or p.data_ptr() != base + e[0] * 4:
                return False
        return True

    def grad_view(self, p):
        o, n = self.index[id(p)]
        return self.grad[o:o + n].view(p.shape)

    def contiguous(self, params):
        """True when `params` occupy one gap-free run of the flat buffers, in this order."""
        for a, b in zip(params[:-1], params[1:]):
            oa, na = self.index[id(a)]
            if self.index[id(b)][0] != oa + na:
                return False
        return True

    def grad_run(self, params):
        o = self.index[id(params[0])][0]
        n = sum(self.index[id(p)][1] for p in params)
        # remembered for TrainEngine.grad_buckets: a kernel handed this pointer writes n floats, not one parameter
        runs = self.__dict__.setdefault("runs", {})
        runs[o] = max(runs.get(o, 0), n)
        return self.grad[o:o + n]

    def write_extent(self, off):
        """Floats a backward op holding a pointer at float offset `off` of the gradient buffer may write: the recorded
        multi-parameter run starting there, else the rest of the parameter containing `off`."""
        n = self.__dict__.get("runs", {}).get(off, 0)
        if getattr(self, "_starts", None) is None:
            self._starts = sorted((o, ln) for o, ln in self.index.values())
        import bisect
        i = bisect.bisect_right(self._starts, (off, float("inf"))) - 1
        if i >= 0:
            o, ln = self._starts[i]
            if off < o + ln:
                n = max(n, o + ln - off)
        return max(n, 1)

    def attach_grads(self):
        """Expose the flat gradient buffer through `p.grad` (views, no copies)."""
        for p in self.params:
            p.grad = self.grad_view(p)


def flat_params_of(model, device):
    fp = getattr(model, "_flat_params", None)
    if fp is None or fp.device != device or not fp.owns(model):
        fp = FlatParams(model, device)
        model._flat_params = fp
    return fp


class TrainEngine(E.UNetEngine):
    """Forward (train mode) + backward program of NCSNpp at a fixed (batch, H, W)."""

    def __init__(self, model, batch, height, width, device, vp_score=False, input_grad=False, dropout=True,
                 param_grads=True):
        # param_grads=False: only d out / d x is needed (likelihood.py's Hutchinson divergence): the backward program
        # carries no weight-gradient / bias-gradient kernels (a third of its FLOPs)
        self.param_grads = param_grads
        self.flat = flat_params_of(model, device)
        super().__init__(model, batch, height, width, device, vp_score=vp_score, train=bool(dropout), input_grad=input_grad,
                         finalize=False)
        self.n_fwd = len(self.b.specs)
        # d loss / d out in the boundary layout (NCHW), written by the caller (loss head or autograd)
        self.gout = self.b.buf(batch, self.channels, height, width, name="gout", persistent=True)
        self.gx = self.b.buf(batch, self.channels, height, width, name="gx", persistent=True) if input_grad else None
        self._G = {}
        # the small second stages of column sums and GroupNorm backward passes (~200 launches of 5-8 us per step) are deferred
        # and finished up to L.FINISH_JOBS at a time by one launch (ssde_colsum_finish / ssde_gn_bwd_finish); SSDE_DEFER_FINISH=0
        # keeps the per-call launches (A/B runs, tests of both forms)
        import os
        self.defer_finish = os.environ.get("SSDE_DEFER_FINISH", "1") != "0"
        self._jobs = {L.OP_COLSUM_FINISH: [], L.OP_GN_BWD_FINISH: []}
        self._pending_per = set()
        self._lower_backward()
        self.program = self.b.finalize()
        self.n_fwd = self.program.spec_start[self.n_fwd]      # spec count -> op count
        self.n_bwd = self.program.n - self.n_fwd

    # ------------------------------------------------------------------ helpers
    def _needs(self, buf):
        if buf is None:
            return False
        if isinstance(buf, tuple):
            buf = buf[0]
        return id(buf) not in self._nograd

    def _gentry(self, buf):
        e = self._G.get(id(buf))
        if e is None:
            e = self._G[id(buf)] = [self.b.buf(*buf.shape, name="g_" + buf.name), False]
        return e

    def _param_of(self, packed):
        return self.weights.meta[id(packed)]["parts"][0]["param"]

    # ------------------------------------------------------------------ the tape walk
    def _lower_backward(self):
        b = self.b
        self._nograd = {id(self.x_in), id(self.cond), id(self._emb)}
        if self.sig is not self.cond:
            self._nograd.add(id(self.sig))
        if self.std is not None:
            self._nograd.add(id(self.std))
        if not self.input_grad:
            self._nograd.add(id(self._x0))
        fwd = list(b.specs[: self.n_fwd])
        if self.param_grads:
            b.add(L.OP_MEMSET, dict(dst=self.flat.grad, bytes=self.flat.numel * 4, value=0), FC_BWD)
        handlers = {L.OP_CONV: self._bwd_conv, L.OP_UPFIRDN: self._bwd_fir, L.OP_ATTN: self._bwd_attn,
                    L.OP_TO_NCHW: self._bwd_to_nchw, L.OP_TO_NHWC: self._bwd_to_nhwc}
        for kind, f, _, _ in reversed(fwd):
            h = handlers.get(kind)
            if h is not None:
                h(f)
        for kind in list(self._jobs):
            self._flush_finish(kind)

    # ------------------------------------------------------------------ deferred finishing launches
    def _flush_finish(self, kind):
        jobs = self._jobs[kind]
        if jobs:
            self.b.add(kind, dict(count=len(jobs), job=list(jobs)), FC_BWD)
            del jobs[:]
        if kind == L.OP_COLSUM_FINISH:
            self._pending_per.clear()

    def _defer(self, kind, job):
        self._jobs[kind].append(job)
        if len(self._jobs[kind]) == L.FINISH_JOBS:
            self._flush_finish(kind)

    def _colsum(self, fields):
        """One column-sum call: its pass over g now, its reductions over slices and samples with the next finishing launch."""
        if not self.defer_finish:
            self.b.add(L.OP_COLSUM, fields, FC_BWD)
            return
        self.b.add(L.OP_COLSUM, dict(fields, per_sample=None, total=None, total2=None, flags=L.COLSUMF_DEFER), FC_BWD)
        if fields["per_sample"] is not None:
            self._pending_per.add(id(fields["per_sample"]))
        self._defer(L.OP_COLSUM_FINISH, dict(part=fields["scratch"], per_sample=fields["per_sample"], total=fields["total"],
                                             total2=fields["total2"], n=fields["n"], slices=max(1, min(32, fields["hw"] // 64)),
                                             c=fields["c"], ps_ld=fields["ps_ld"], ps_off=fields["ps_off"]))

    def _bwd_to_nchw(self, f):
        e = self._gentry(f["src"])
        self.b.add(L.OP_TO_NHWC, dict(src=self.gout, dst=e[0], n=f["n"], c=f["c"], h=f["h"], w=f["w"], c_pad=f["c_src"],
                                      a=1.0, b=0.0, mode=f["mode"], v=f["v"]), FC_BWD)
        e[1] = True

    def _bwd_to_nhwc(self, f):
        if not self.input_grad:
            return
        e = self._G.get(id(f["dst"]))
        if e is None:
            return
        self.b.add(L.OP_TO_NCHW, dict(src=e[0], dst=self.gx, n=f["n"], c=f["c"], h=f["h"], w=f["w"], c_src=f["c_pad"],
                                      mode=0, v=None, alpha=float(f["a"]), accumulate=0), FC_BWD)

    def _accum(self, target, dp, dp_ld, dp_off, c, n, hw, scale):
        """grad(target) (+)= scale * dp[:, dp_off:dp_off+c]"""
        if not self._needs(target):
            return
        e = self._gentry(target)
        src = dict(E._NOSRC); src.update(c0=c)
        self.b.add(L.OP_PROLOGUE_BWD, dict(src=src, dp=dp, dp_ld=dp_ld, dp_off=dp_off, n=n, hw=hw, sums=None, scale=float(scale),
                                           acc0=int(e[1]), acc1=0, g0=e[0], g1=None), FC_BWD)
        e[1] = True

    def _bwd_prologue(self, src, dP, n, hw):
        b = self.b
        ctot = src["c0"] + src["c1"]
        e0 = self._gentry(src["p0"]) if self._needs(src["p0"]) else None
        e1 = self._gentry(src["p1"]) if self._needs(src["p1"]) else None
        if src["pro_mode"] in (L.PRO_GN, L.PRO_GN_SILU):
            if e0 is None and e1 is None and not self.param_grads:
                return
            # ONE op: reduction, dgamma / dbeta and (with g0 / g1) the gradient of the sources; the library reads dp and x
            # once where a sample's run of groups fits a workgroup's registers (backward.hip, gn_bwd_fused_kernel)
            groups = src["gn_groups"]
            sums = b.buf(n, groups, 2, name="gn_bwd_sums")
            slices = max(1, min(int(math.ceil(256 / n)), hw // 64)) if hw >= 128 else 1
            scratch = b.buf(n * slices * ctot * 2, name="gn_bwd_scratch")
            fields = dict(src=src, dp=dP, n=n, hw=hw, sums=sums,
                          dgamma=self.flat.grad_view(self._param_of(src["gn_gamma"])) if self.param_grads else None,
                          dbeta=self.flat.grad_view(self._param_of(src["gn_beta"])) if self.param_grads else None,
                          scratch=scratch, slices=slices,
                          g0=e0[0] if e0 else None, g1=e1[0] if e1 else None,
                          acc0=int(e0[1]) if e0 else 0, acc1=int(e1[1]) if e1 else 0, scale=1.0, flags=L.gn_bwd_route_flags())
            if self.defer_finish or not self.param_grads:
                # dgamma / dbeta: with the next finishing launch -- or never, when no parameter gradient is wanted (likelihood.py)
                dg, db = fields["dgamma"], fields["dbeta"]
                fields.update(dgamma=None, dbeta=None, flags=fields["flags"] | L.GNBWDF_DEFER_PARAMS)
                b.add(L.OP_GN_BWD_REDUCE, fields, FC_BWD)
                if self.param_grads:
                    self._defer(L.OP_GN_BWD_FINISH, dict(scratch=scratch, dgamma=dg, dbeta=db, rows=self._gn_bwd_rows(fields), c=ctot))
            else:
                b.add(L.OP_GN_BWD_REDUCE, fields, FC_BWD)
        else:
            if e0 is None and e1 is None:
                return
            b.add(L.OP_PROLOGUE_BWD, dict(src=src, dp=dP, dp_ld=ctot, dp_off=0, n=n, hw=hw, sums=None, scale=1.0,
                                          acc0=int(e0[1]) if e0 else 0, acc1=int(e1[1]) if e1 else 0,
                                          g0=e0[0] if e0 else None, g1=e1[0] if e1 else None), FC_BWD)
        if e0:
            e0[1] = True
        if e1:
            e1[1] = True

    @staticmethod
    def _gn_bwd_rows(fields):
        """Rows of the channel sums a deferred GroupNorm backward call leaves in its scratch (shape-only query of the library)."""
        a = L.GnBwdReduceArgs()
        s = fields["src"]
        a.src.c0, a.src.c1, a.src.pro_mode, a.src.gn_groups = s["c0"], s["c1"], s["pro_mode"], s["gn_groups"]
        a.n, a.hw, a.slices, a.flags = fields["n"], fields["hw"], fields["slices"], fields["flags"]
        a.g0 = 0x1000 if fields["g0"] is not None else None
        a.g1 = 0x1000 if fields["g1"] is not None else None
        return int(L.load().ssde_gn_bwd_scratch_rows(C.byref(a)))

    def _bwd_bias(self, f, g, g_ld, n, hw, scale):
        b = self.b
        if not self.param_grads:
            if f["chan_add"] is not None:      # only the temb addend's gradient (it feeds d/dx of nothing, but keeps the tape uniform)
                tbuf, off = f["chan_add"]
                e = self._gentry(tbuf)
                e[1] = True
                self._colsum(dict(g=g, g_ld=g_ld, g_off=0, n=n, hw=hw, c=f["c_out"], scale=float(scale), per_sample=e[0],
                                        ps_ld=f["chan_add_ld"], ps_off=off, total=None, total2=None,
                                        scratch=b.buf(n * (max(1, min(32, hw // 64)) + 1) * f["c_out"], name="colsum_scratch")))
            return
        per, ps_ld, ps_off = None, 0, 0
        if f["chan_add"] is not None:
            tbuf, off = f["chan_add"]
            e = self._gentry(tbuf)
            per, ps_ld, ps_off = e[0], f["chan_add_ld"], off
            e[1] = True
        parts, mode = [], "cat"
        if f["bias"] is not None:
            meta = self.weights.meta[id(f["bias"])]
            parts, mode = meta["parts"], meta["mode"]
        if mode == "sum":
            assert len(parts) == 2
            c = parts[0]["n"]
            scratch = b.buf(n * (max(1, min(32, hw // 64)) + 1) * c, name="colsum_scratch")
            self._colsum(dict(g=g, g_ld=g_ld, g_off=0, n=n, hw=hw, c=c, scale=float(scale), per_sample=per, ps_ld=ps_ld,
                                    ps_off=ps_off, total=self.flat.grad_view(parts[0]["param"]),
                                    total2=self.flat.grad_view(parts[1]["param"]), scratch=scratch))
            return
        if not parts and per is not None:
            self._colsum(dict(g=g, g_ld=g_ld, g_off=0, n=n, hw=hw, c=f["c_out"], scale=float(scale), per_sample=per,
                                    ps_ld=ps_ld, ps_off=ps_off, total=None, total2=None,
                                    scratch=b.buf(n * (max(1, min(32, hw // 64)) + 1) * f["c_out"], name="colsum_scratch")))
            return
        if len(parts) > 1 and per is None and self.flat.contiguous([pt["param"] for pt in parts]):
            # concatenated biases stored back to back (Dense_0 of every block): one launch for all of them
            c = sum(pt["n"] for pt in parts)
            self._colsum(dict(g=g, g_ld=g_ld, g_off=parts[0]["off"], n=n, hw=hw, c=c, scale=float(scale), per_sample=None,
                                    ps_ld=0, ps_off=0, total=self.flat.grad_run([pt["param"] for pt in parts]), total2=None,
                                    scratch=b.buf(n * (max(1, min(32, hw // 64)) + 1) * c, name="colsum_scratch")))
            return
        for i, part in enumerate(parts):
            use_per = per if (i == 0 and len(parts) == 1) else None
            assert per is None or len(parts) == 1
            scratch = b.buf(n * (max(1, min(32, hw // 64)) + 1) * part["n"], name="colsum_scratch")
            self._colsum(dict(g=g, g_ld=g_ld, g_off=part["off"], n=n, hw=hw, c=part["n"], scale=float(scale),
                                    per_sample=use_per, ps_ld=ps_ld, ps_off=ps_off,
                                    total=self.flat.grad_view(part["param"]), total2=None, scratch=scratch))

    def _bwd_branch(self, f, src, wpacked, g, g_ld, scale, ksize):
        b, low, n = self.b, self.low, f["n"]
        ho, wo = f["h_out"], f["w_out"]
        if ksize == 3:
            h_in, w_in, stride, pad = f["h_in"], f["w_in"], f["stride"], f["pad"]
        else:
            h_in, w_in, stride, pad = ho, wo, 1, 0
        meta = self.weights.meta[id(wpacked)]
        ctot = src["c0"] + src["c1"]
        parts = meta["parts"] if self.param_grads else []
        if len(parts) > 1 and not any(pt["transpose"] for pt in parts) and self.flat.contiguous([pt["param"] for pt in parts]):
            # row-concatenated [out_i, in] matrices stored back to back ARE the [sum out_i, in] matrix
            whole = self.flat.grad_run([pt["param"] for pt in parts])
            parts = [dict(param=None, row0=parts[0]["row0"], rows=sum(pt["rows"] for pt in parts), transpose=False, view=whole)]
        for part in parts:
            flops = 2.0 * n * ho * wo * ksize * ksize * meta["cin_store"] * part["rows"]
            fields = dict(src=src, g=g, g_ld=g_ld, g_off=part["row0"], n=n, h_in=h_in, w_in=w_in, h_out=ho, w_out=wo,
                          c_out=part["rows"], ksize=ksize, stride=stride, pad=pad, cin_store=meta["cin_store"],
                          transpose_out=int(part["transpose"]), splits=0, scale=float(scale),
                          dw=part["view"] if part.get("view") is not None else self.flat.grad_view(part["param"]),
                          scratch=None, scratch_floats=0,
                          # the forward launch's by-product (engine.Lowering.conv: only allocated when THIS launch takes it)
                          v_pre=f.get("wino_v") if (ksize == 3 and len(parts) == 1 and f.get("_v_for_wgrad", True)) else None)
            need = self._wgrad_scratch(fields)
            if need > 0:
                fields.update(scratch=b.buf(need, name="wgrad_slabs"), scratch_floats=need)
            b.add(L.OP_WGRAD, fields, FC_WGRAD, flops)
        if not (self._needs(src["p0"]) or self._needs(src["p1"])):
            return
        direct = src["pro_mode"] == L.PRO_NONE and src["p1"] is None
        if direct:
            e = self._gentry(src["p0"])
            dst, resid = e[0], (e[0] if e[1] else None)
            e[1] = True
        else:
            dst, resid = b.buf(n, h_in, w_in, ctot, name="dP"), None
        gsrc = _src(g, g_ld)
        if ksize == 3:
            wino = low.wino_ok(h_in, w_in, ctot, g_ld) if stride == 1 else 0
            pack = {6: E.pack_wino4r_weight, 4: E.pack_wino4_weight, 2: E.pack_wino_weight}.get(wino, E.pack_conv_weight)
            wd = self.weights.derived(wpacked, lambda w: pack(w.permute(1, 0, 2, 3).flip(2, 3)),
                                      {6: "dgrad_wino4r", 4: "dgrad_wino4", 2: "dgrad_wino"}.get(wino, "dgrad"))
            if stride == 1:
                low.conv(dst, h_in, w_in, ctot, main=gsrc, w_main=wd, h_in=ho, w_in=wo, stride=1, pad=1, resid=resid,
                         resid_post=1, scale=scale, wino=wino)
            else:
                # transposed strided conv = zero-insertion (upfirdn, up=2, 1x1 kernel) + stride-1 conv with pad 2, cropped
                assert stride == 2 and pad == 0
                gz, hz, wz = low.upfirdn(gsrc, g_ld, ho, wo, np.ones((1, 1), np.float32), up=2, pad=(0, 0), name="g_zero_ins")
                low.conv(dst, h_in, w_in, ctot, main=_src(gz, g_ld), w_main=wd, h_in=hz, w_in=wz, stride=1, pad=2,
                         resid=resid, resid_post=1, scale=scale)
        else:
            wd = self.weights.derived(wpacked, lambda m: E.pack_matrix(m.t().contiguous()), "dgrad")
            low.conv(dst, ho, wo, ctot, aux=gsrc, w_aux=wd, resid=resid, resid_post=1, scale=scale)
        if not direct:
            self._bwd_prologue(src, dst, n, h_in * w_in)

    @staticmethod
    def _wgrad_scratch(fields):
        """Ask the library how much slab scratch its preferred pixel split of this launch needs (shape-only query)."""
        a = L.WgradArgs()
        for k in ("g_ld", "g_off", "n", "h_in", "w_in", "h_out", "w_out", "c_out", "ksize", "stride", "pad", "cin_store",
                  "transpose_out"):
            setattr(a, k, fields[k])
        s = fields["src"]
        a.src.c0, a.src.c1, a.src.pro_mode, a.src.gn_groups = s["c0"], s["c1"], s["pro_mode"], s["gn_groups"]
        a.flags = fields.get("flags", L.wgrad_route_flags())
        r = int(L.load().ssde_wgrad_scratch_floats(C.byref(a)))
        if r < 0:
            L.check(r, "ssde_wgrad_scratch_floats")
        return r

    def _bwd_conv(self, f):
        e = self._G.get(id(f["dst"]))
        if e is None:
            return
        g, scale, n = e[0], f["out_scale"], f["n"]
        if id(g) in self._pending_per:               # this gradient still waits for deferred per-sample column sums
            self._flush_finish(L.OP_COLSUM_FINISH)
        hw, g_ld = f["h_out"] * f["w_out"], f["dst"].shape[-1]
        if f["resid"] is not None:
            self._accum(f["resid"], g, g_ld, 0, g_ld, n, hw, scale)
        if f["bias"] is not None or f["chan_add"] is not None:
            self._bwd_bias(f, g, g_ld, n, hw, scale)
        if f["ksize"] == 3:
            self._bwd_branch(f, f["main"], f["w_main"], g, g_ld, scale, 3)
        if f["aux"]["p0"] is not None:
            self._bwd_branch(f, f["aux"], f["w_aux"], g, g_ld, scale, 1)

    def _bwd_fir(self, f):
        e = self._G.get(id(f["dst"]))
        src = f["src"]
        if e is None or not self._needs(src["p0"]):
            return
        b, n, c = self.b, f["n"], f["c"]
        up, down, pad0, kh, kw = f["up"], f["down"], f["pad0"], f["kh"], f["kw"]
        # UpFirDn2d.backward (op/upfirdn2d.py:111-116): same op on the gradient with up/down swapped,
        # the flipped kernel and these pads
        gp0 = kw - pad0 - 1
        gp1 = f["w_in"] * up - f["w_out"] * down + pad0 - up + 1
        assert gp0 >= 0 and gp1 >= 0, "negative gradient pads are not lowered"
        k = np.asarray(f["k"][: kh * kw], dtype=np.float32).reshape(kh, kw)[::-1, ::-1]
        k16 = [0.0] * 16
        for i, v in enumerate(k.reshape(-1).tolist()):
            k16[i] = float(v)
        direct = src["pro_mode"] == L.PRO_NONE
        if direct:
            t = self._gentry(src["p0"])
            dst, acc = t[0], int(t[1])
            t[1] = True
        else:
            dst, acc = b.buf(n, f["h_in"], f["w_in"], c, name="dP_fir"), 0
        b.add(L.OP_UPFIRDN, dict(src=_src(e[0], c), n=n, h_in=f["h_out"], w_in=f["w_out"], c=c, h_out=f["h_in"], w_out=f["w_in"],
                                 up=down, down=up, pad0=gp0, pad1=gp1, kh=kh, kw=kw, k=k16, dst=dst, accumulate=acc), E.FC_FIR)
        if not direct:
            self._bwd_prologue(src, dst, n, f["h_in"] * f["w_in"])

    def _bwd_attn(self, f):
        e = self._G.get(id(f["dst"]))
        if e is None:
            return
        gq = self._gentry(f["qkv"])
        stats = self.b.buf(f["n"], f["l"], 4, name="attn_stats")
        self.b.add(L.OP_ATTN_BWD, dict(qkv=f["qkv"], o=f["dst"], d_o=e[0], dqkv=gq[0], stats=stats, n=f["n"], l=f["l"], c=f["c"],
                                       scale=f["scale"]), E.FC_ATTN, 14.0 * f["n"] * f["l"] * f["l"] * f["c"])
        gq[1] = True

    # ------------------------------------------------------------------ execution
    def set_dropout_seed(self, seed):
        if self.drop_seed is not None:
            self.drop_seed.fill_(int(seed) & 0x7FFFFFFF)

    def run_forward(self):
        self.program.run_range(0, self.n_fwd)

    def run_backward(self):
        self.program.run_range(self.n_fwd, self.n_bwd)

    # ------------------------------------------------------------------ gradient buckets (data-parallel overlap)
    def grad_buckets(self, bucket_floats=8 << 20):
        """[(lo, hi, op_end)] from the END of the flat gradient buffer to its start: the slice [lo, hi) is final once the
        backward ops before index `op_end` have run.  The backward walks the network in reverse while the flat buffer
        is laid out in forward order, so the tail of the buffer completes first; `op_end` is found by scanning every
        pointer field of every backward op for addresses inside the flat gradient; a slice is final after the LAST op
        whose write extent (FlatParams.write_extent: its parameter, or the multi-parameter run it was handed) reaches
        at or beyond the slice's start -- an op whose pointer lies below `lo` but whose run crosses it counts too."""
        if getattr(self, "_buckets", None) is not None and self._buckets[0] == bucket_floats:
            return self._buckets[1]
        import ctypes as C
        base, end = self.flat.grad.data_ptr(), self.flat.grad.data_ptr() + self.flat.numel * 4

        def pointers(struct):
            raw = bytes(struct)
            for off in L.pointer_offsets(type(struct)):
                v = int.from_bytes(raw[off:off + 8], "little")
                if v:
                    yield v
        touch = []                                   # (float offset, op index) of every reference into flat.grad
        for i in range(self.n_fwd, self.program.n):
            op = self.program.ops[i]
            if op.kind == L.OP_MEMSET:
                continue                             # the zero-fill at the head of the backward program
            for ptr in pointers(getattr(op.u, L._UNION_FIELD[op.kind])):
                if base <= ptr < end:
                    touch.append(((ptr - base) // 4, i))
        # bucket boundaries at parameter starts; the grouped parameters at the head of the buffer (written as one run by a
        # single kernel) stay in one bucket
        groups = self.model.flat_param_groups() if hasattr(self.model, "flat_param_groups") else []
        grouped_end = 0
        for g in groups:
            for p in g:
                o, n = self.flat.index[id(p)]
                grouped_end = max(grouped_end, o + n)
        starts = sorted(o for o, _ in self.flat.index.values())
        bounds, hi = [], self.flat.numel
        for o in reversed(starts):
            if o < grouped_end:
                break
            if hi - o >= bucket_floats:
                bounds.append((o, hi)); hi = o
        if hi > 0:
            bounds.append((0, hi))
        buckets = []
        for lo, hi_ in bounds:
            last = max([i for off, i in touch if off + self.flat.write_extent(off) > lo], default=self.n_fwd - 1)
            buckets.append((lo, hi_, last + 1))
        self._buckets = (bucket_floats, buckets)
        return buckets

    def run_backward_bucketed(self, on_ready, bucket_floats=8 << 20):
        """Backward program in segments; `on_ready(lo, hi)` is called as soon as the ops that finalise flat.grad[lo:hi]
        are ENQUEUED (stream order makes a collective issued there wait for exactly those ops)."""
        cur = self.n_fwd
        for lo, hi, op_end in self.grad_buckets(bucket_floats):
            if op_end > cur:
                self.program.run_range(cur, op_end - cur)
                cur = op_end
            on_ready(lo, hi)
        if cur < self.program.n:
            self.program.run_range(cur, self.program.n - cur)

    def forward_train(self, x, cond, seed=0):
        if tuple(x.shape) != (self.n, self.channels, self.h, self.w):
            raise ValueError("engine built for %s, got %s" % ((self.n, self.channels, self.h, self.w), tuple(x.shape)))
        self.weights.refresh()
        self.set_dropout_seed(seed)
        self.load_inputs(x.contiguous(), cond)
        self.run_forward()
        return self.output_view()

    def backward(self, grad_out):
        """Runs the backward program for d loss / d out = grad_out [N, C, H, W]; parameter gradients land in
        `self.flat.grad` (zeroed first), d loss / d x in `gx_view()` when built with input_grad."""
        self.gout.tensor[: grad_out.numel()].copy_(grad_out.reshape(-1))
        self.run_backward()

    def gx_view(self):
        return self.gx.tensor[: self.n * self.channels * self.h * self.w].view(self.n, self.channels, self.h, self.w)
