"""Lowers an NCSNpp module to a static program of HIP kernels (libssde_hip.so) and runs it.

The reference walks ~60 nn.Modules per forward and issues ~900 eager kernels
(models/ncsnpp.py:232-381).  Here the walk happens ONCE per
(batch, resolution): `UNetEngine` records a flat op list (include/ssde.h `ssde_op`),
plans activation storage by liveness (buffers are recycled as soon as their last
consumer has been enqueued), packs weights into the layouts the kernels read, and
afterwards a forward is a single C call (`ssde_program_run`) or a hipGraph replay.

Fusions encoded in the program (per ResnetBlockBigGANpp, models/layerspp.py:242-274):
  GroupNorm stats  -> 1 reduction kernel (normalise+affine+SiLU is applied while the
                      consumer stages its LDS tile; the normalised tensor never exists)
  Conv_0           -> conv kernel, epilogue adds bias and Dense_0(SiLU(temb)) (all 40+
                      Dense_0 projections are ONE batched GEMM at the start)
  Conv_1 (+Conv_2) -> conv kernel; the 1x1 skip conv is an extra K-range of the same
                      GEMM; epilogue adds the identity skip and scales by 1/sqrt(2)
  torch.cat        -> never materialised: a source is a pair of tensors
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib as L

_FLT = 4
INV_SQRT2 = float(1.0 / np.sqrt(2.0))

# flops_class tags (echoed by ssde_program_run_timed; bench.py groups by them)
FC_OTHER, FC_CONV3, FC_CONV1, FC_ATTN, FC_GN, FC_FIR = 0, 1, 2, 3, 4, 5


class Buf:
    """A symbolic activation buffer; storage is assigned by ProgramBuilder.finalize()."""
    __slots__ = ("shape", "numel", "name", "persistent", "tensor", "first", "last")

    def __init__(self, shape, name="", persistent=False):
        self.shape = tuple(int(s) for s in shape)
        self.numel = int(np.prod(self.shape))
        self.name = name
        self.persistent = persistent
        self.tensor = None
        self.first = None
        self.last = None


def _walk_refs(value, fn):
    if isinstance(value, Buf):
        fn(value)
    elif isinstance(value, tuple) and len(value) == 2 and isinstance(value[0], Buf):
        fn(value[0])
    elif isinstance(value, dict):
        for v in value.values():
            _walk_refs(v, fn)
    elif isinstance(value, list):                    # job lists of the batched finishing ops
        for v in value:
            _walk_refs(v, fn)


def _ptr(value):
    if value is None:
        return None
    off = 0
    if isinstance(value, tuple):
        value, off = value
    t = value.tensor if isinstance(value, Buf) else value
    assert t is not None, "buffer without storage"
    return t.data_ptr() + off * _FLT


def _fill(struct, fields):
    for k, v in fields.items():
        if k.startswith("_"):
            continue            # lowering-time annotations (e.g. the scratch of a split Winograd conv)
        cur = getattr(struct, k)
        if isinstance(v, dict):
            _fill(cur, v)
        elif isinstance(v, (list, tuple)) and not (len(v) == 2 and isinstance(v[0], (Buf, torch.Tensor))):
            for i, x in enumerate(v):
                if isinstance(x, dict):
                    _fill(cur[i], x)             # an array of structures (the jobs of a batched finishing op)
                else:
                    cur[i] = x
        elif isinstance(v, (Buf, torch.Tensor, tuple)) or v is None:
            setattr(struct, k, _ptr(v))
        else:
            setattr(struct, k, v)


_STRUCT = {L.OP_CONV: L.ConvArgs, L.OP_GN_STATS: L.GnStatsArgs, L.OP_UPFIRDN: L.UpfirdnArgs, L.OP_ATTN: L.AttnArgs,
           L.OP_EMBED: L.EmbedArgs, L.OP_TO_NHWC: L.ToNhwcArgs, L.OP_TO_NCHW: L.ToNchwArgs,
           L.OP_BIAS_ACT: L.BiasActArgs, L.OP_SUMSQ: L.SumsqArgs, L.OP_RANDN: L.RandnArgs,
           L.OP_LANGEVIN: L.LangevinArgs, L.OP_PREDICTOR: L.PredictorArgs, L.OP_FILL: L.FillArgs,
           L.OP_STEP_INC: L.StepIncArgs, L.OP_WGRAD: L.WgradArgs, L.OP_COLSUM: L.ColsumArgs,
           L.OP_GN_BWD_REDUCE: L.GnBwdReduceArgs, L.OP_PROLOGUE_BWD: L.PrologueBwdArgs, L.OP_ATTN_BWD: L.AttnBwdArgs,
           L.OP_PERTURB: L.PerturbArgs, L.OP_DSM_LOSS: L.DsmLossArgs, L.OP_SUMSQ_FLAT: L.SumsqFlatArgs,
           L.OP_ADAM: L.AdamArgs, L.OP_MEMSET: L.MemsetArgs, L.OP_AXPY: L.AxpyArgs, L.OP_GN_FINALIZE: L.GnFinalizeArgs,
           L.OP_COLSUM_FINISH: L.ColsumFinishArgs, L.OP_GN_BWD_FINISH: L.GnBwdFinishArgs}


_ROUTE = {L.OP_CONV: L.conv_route_flags, L.OP_WGRAD: L.wgrad_route_flags, L.OP_GN_BWD_REDUCE: L.gn_bwd_route_flags,
          L.OP_ATTN: L.attn_route_flags}


class ProgramBuilder:
    def __init__(self, device):
        self.device = device
        self.specs = []      # (kind, fields, flops_class, flops)
        self.keep = []       # tensors that must outlive the program

    def buf(self, *shape, name="", persistent=False):
        return Buf(shape, name, persistent)

    def tensor(self, t):
        self.keep.append(t)
        return t

    def add(self, kind, fields, fclass=FC_OTHER, flops=0.0):
        self.specs.append((kind, fields, fclass, float(flops)))

    def extend(self, other):
        self.specs.extend(other.specs)
        self.keep.extend(other.keep)

    def finalize(self):
        """Liveness-planned storage + ctypes op array."""
        bufs = {}
        # FIR pairs are fused into the FIRST spec's launch (dst2): the partner's destination is written at spec i, one
        # spec before its own -- its lifetime must start there, or the allocator could hand it a block that spec i still
        # reads (round-2 advisor finding: safe only by the accident of block sizes)
        fused_fir = _fir_pairs(self.specs)
        for i, (_, fields, _, _) in enumerate(self.specs):
            def mark(b, i=i):
                if b.first is None:
                    b.first = i
                b.last = i
                bufs[id(b)] = b
            _walk_refs(fields, mark)
            if fused_fir.get(i) is not None:
                _walk_refs({"dst2": fused_fir[i]}, mark)
        by_first, by_last = {}, {}
        for b in bufs.values():
            by_first.setdefault(b.first, []).append(b)
            by_last.setdefault(b.last, []).append(b)
        free, blocks = [], []
        for i in range(len(self.specs)):
            for b in by_first.get(i, []):
                if b.tensor is not None:
                    continue
                need = (b.numel + 1023) // 1024 * 1024
                if b.persistent:
                    b.tensor = torch.zeros(need, dtype=torch.float32, device=self.device)
                    blocks.append(b.tensor)
                    continue
                best = None
                for j, t in enumerate(free):
                    if t.numel() >= need and (best is None or t.numel() < free[best].numel()):
                        best = j
                if best is not None and free[best].numel() <= 2 * need:
                    b.tensor = free.pop(best)
                else:
                    b.tensor = torch.zeros(need, dtype=torch.float32, device=self.device)
                    blocks.append(b.tensor)
            for b in by_last.get(i, []):
                if not b.persistent:
                    free.append(b.tensor)
        self.blocks = blocks
        self.arena_bytes = sum(t.numel() for t in blocks) * _FLT
        ops, classes, flops, starts = [], [], [], []
        for si, (kind, fields, fclass, fl) in enumerate(self.specs):
            starts.append(len(ops))
            if si in fused_fir:
                if fused_fir[si] is None:
                    continue                     # the raw-source half of a pair: issued by its partner through dst2
                fields = dict(fields, dst2=fused_fir[si])
            for sub_fields, sub_class, sub_fl in _expand(kind, fields, fclass, fl):
                args = _STRUCT[kind]()
                _fill(args, sub_fields)
                if kind in _ROUTE and "flags" not in sub_fields:
                    args.flags = _ROUTE[kind]()          # A/B switches of the environment, read ONCE here (see _lib.py)
                ops.append(L.make_op(kind, args, sub_class))
                classes.append(sub_class)
                flops.append(sub_fl)
        prog = Program(L.op_array(ops), classes, flops, self)
        prog.spec_start = starts + [len(ops)]      # op index of every lowering spec (a spec may expand to several ops)
        return prog


def _fir_pairs(specs):
    """A residual block resamples act(GroupNorm(x)) and x with the same FIR (layerspp.py:250-258): two adjacent
    OP_UPFIRDN specs on one source, the first with a prologue, the second without.  They run as ONE launch that reads x
    once (ssde_upfirdn_args.dst2); the specs stay separate for the backward lowering.  Returns {spec index: dst2 | None}."""
    out = {}
    if os.environ.get("SSDE_FUSE_FIR", "1") == "0":
        return out
    same = ("n", "h_in", "w_in", "c", "h_out", "w_out", "up", "down", "pad0", "pad1", "kh", "kw", "k")
    for i in range(len(specs) - 1):
        (k0, f0, _, _), (k1, f1, _, _) = specs[i], specs[i + 1]
        if k0 != L.OP_UPFIRDN or k1 != L.OP_UPFIRDN or i in out:
            continue
        s0, s1 = f0["src"], f1["src"]
        if s0["p0"] is not s1["p0"] or s0["p1"] is not s1["p1"] or s0["c0"] != s1["c0"] or s0["c1"] != s1["c1"]:
            continue
        if s0["pro_mode"] == L.PRO_NONE or s1["pro_mode"] != L.PRO_NONE or f0["dst"] is f1["dst"]:
            continue
        if f0.get("accumulate") or f1.get("accumulate") or any(f0[k] != f1[k] for k in same):
            continue
        out[i], out[i + 1] = f1["dst"], None
    return out


def _expand(kind, fields, fclass, fl):
    """One lowering spec -> the C ops that execute it.  A fused (3x3 + 1x1 skip) convolution whose 3x3 part runs on
    the Winograd kernel is issued as two launches: tmp = conv3x3 + bias + temb addend, then the 1x1 GEMM with tmp as
    its residual; the spec stays ONE fused op for the backward lowering."""
    if kind != L.OP_CONV or fields.get("_split_tmp") is None:
        return [(fields, fclass, fl)]
    px = fields["n"] * fields["h_out"] * fields["w_out"]
    fl1 = 2.0 * px * (fields["aux"]["c0"] + fields["aux"]["c1"]) * fields["c_out"]
    a = dict(fields)
    a.update(aux=_NOSRC, w_aux=None, resid=None, resid_post=0, out_scale=1.0, dst=fields["_split_tmp"], gn_part=None)
    b = dict(fields)
    b.update(main=_NOSRC, w_main=None, ksize=0, bias=None, chan_add=None, chan_add_ld=0, resid=fields["_split_tmp"],
             resid_post=0, tile=L.TILE_AUTO, wino_v=None, gn_in_part0=None, gn_in_part1=None, gn_in_slices0=0, gn_in_slices1=0)
    return [(a, fclass, fl - fl1), (b, FC_CONV1, fl1)]


class Program:
    def __init__(self, ops, classes, flops, owner):
        self.ops, self.n = ops, len(ops)
        self.classes, self.flops = classes, flops
        self._owner = owner   # keeps buffers / weights alive
        self._graph = None

    def _launch(self, ops_ptr, count, stream=None):
        lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("libssde_hip programs run on the MI355X only (no CPU fallback)")
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        L.check(lib.ssde_program_run(ops_ptr, count, C.c_void_p(st)), "ssde_program_run")

    def run(self, stream=None):
        self._launch(self.ops, self.n, stream)

    def run_range(self, start, count, stream=None):
        """Run ops [start, start+count) (forward and backward halves of a training program)."""
        assert 0 <= start and start + count <= self.n
        ptr = C.cast(C.byref(self.ops, start * C.sizeof(L.Op)), C.POINTER(L.Op))
        self._launch(ptr, count, stream)

    def run_range_timed(self, start, count):
        lib = L.load()
        ms = (C.c_float * count)()
        ptr = C.cast(C.byref(self.ops, start * C.sizeof(L.Op)), C.POINTER(L.Op))
        st = torch.cuda.current_stream().cuda_stream
        L.check(lib.ssde_program_run_timed(ptr, count, C.c_void_p(st), ms), "ssde_program_run_timed")
        return list(ms)

    def run_timed(self):
        lib = L.load()
        ms = (C.c_float * self.n)()
        st = torch.cuda.current_stream().cuda_stream
        L.check(lib.ssde_program_run_timed(self.ops, self.n, C.c_void_p(st), ms), "ssde_program_run_timed")
        return list(ms)

    def capture(self, stream):
        """Capture into a hipGraph on `stream` (a torch.cuda.Stream, not the default stream)."""
        lib = L.load()
        h = C.c_void_p()
        L.check(lib.ssde_graph_capture(self.ops, self.n, C.c_void_p(stream.cuda_stream), C.byref(h)), "ssde_graph_capture")
        self._graph = h
        return h

    def replay(self, stream):
        L.check(L.load().ssde_graph_launch(self._graph, C.c_void_p(stream.cuda_stream)), "ssde_graph_launch")

    def __del__(self):
        if getattr(self, "_graph", None):
            try:
                L.load().ssde_graph_destroy(self._graph)
            except Exception:
                pass


# --------------------------------------------------------------------------- weights
def pack_conv_weight(w):
    """[Cout, Cin, kh, kw] -> [ceil(Cin/8)][kh*kw][roundup(Cout,64)][8] (layout read by conv_mfma.hip)."""
    cout, cin, kh, kw = w.shape
    cin8, cpad, t = (cin + 7) // 8, (cout + 63) // 64 * 64, kh * kw
    full = torch.zeros(cin8 * 8, t, cpad, dtype=torch.float32, device=w.device)
    full[:cin, :, :cout] = w.detach().to(torch.float32).permute(1, 2, 3, 0).reshape(cin, t, cout)
    return full.reshape(cin8, 8, t, cpad).permute(0, 2, 3, 1).contiguous()


_WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


def pack_wino_weight(w):
    """[Cout, Cin, 3, 3] -> Winograd F(2x2,3x3) weights U = G g G^T arranged as the LDS image conv_wino.hip reads:
    [ceil(Cin/8)][ceil(Cout/64)][16 positions][4 channel pairs][64 couts with bit 4 ^= pair parity][2]."""
    cout, cin = w.shape[0], w.shape[1]
    G = _WINO_G.to(w.device, torch.float32)
    u = torch.einsum("ak,ockl,bl->ocab", G, w.detach().to(torch.float32), G)     # [Cout, Cin, 4, 4]
    c8, nt = (cin + 7) // 8, (cout + 63) // 64
    full = torch.zeros(nt * 64, c8 * 8, 16, dtype=torch.float32, device=w.device)
    full[:cout, :cin] = u.reshape(cout, cin, 16)
    full = full.reshape(nt, 64, c8, 4, 2, 16)                   # [nt, co, c8, pair, e, pos]
    co = torch.arange(64, device=w.device)
    out = torch.empty(c8, nt, 16, 4, 64, 2, dtype=torch.float32, device=w.device)
    for q in range(4):
        src = full[:, :, :, q]                                  # [nt, co, c8, e, pos]
        out[:, :, :, q, co ^ ((q & 1) << 4)] = src.permute(2, 0, 4, 1, 3)   # -> [c8, nt, pos, co, e]
    return out.contiguous()


_WINO4_G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                         [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)


def pack_wino4_weight(w):
    """[Cout, Cin, 3, 3] -> Winograd F(4x4,3x3) weights U = G g G^T (formed in fp64, stored in fp32) arranged as the LDS
    image conv_wino4.hip reads: [ceil(Cin/4)][ceil(Cout/64)][8 waves][9][32 couts][4 channels] -- wave (q, h) of a workgroup
    owns the transform positions q + 4 j (j = 0..8) and the cout half h of the 64-cout tile, and its 4.5 KB are contiguous."""
    cout, cin = w.shape[0], w.shape[1]
    G = _WINO4_G.to(w.device)
    u = torch.einsum("ak,ockl,bl->ocab", G, w.detach().to(torch.float64), G).to(torch.float32)     # [Cout, Cin, 6, 6]
    c4, nt = (cin + 3) // 4, (cout + 63) // 64
    full = torch.zeros(nt * 64, c4 * 4, 36, dtype=torch.float32, device=w.device)
    full[:cout, :cin] = u.reshape(cout, cin, 36)
    # [nt, h, cl, c4, e, j, q] -> [c4, nt, q, h, j, cl, e]   (pos = 4 j + q, cout = 64 nt + 32 h + cl)
    return full.reshape(nt, 2, 32, c4, 4, 9, 4).permute(3, 0, 6, 1, 5, 2, 4).contiguous()


def pack_wino4r_weight(w):
    """[Cout, Cin, 3, 3] -> the F(4x4,3x3) weights U = G g G^T (formed in fp64, stored in fp32) arranged per LANE for
    conv_wino4r.hip, which loads them straight into MFMA operand registers (SSDE_TILE_WINOGRAD4R / SSDE_PACK_WINO4R):
    [ceil(Cin/4)][ceil(Cout/64)][8 waves][4 positions x [64 lanes][4] | [64 lanes][2]] -- wave w owns transform positions
    4 w .. 4 w + 3 with both 32-cout halves of the tile and half w & 1 of position 32 + (w >> 1); lane (lh, li) holds channels
    2 lh, 2 lh + 1 of couts li and 32 + li (four floats per full position), of cout 32 (w & 1) + li for the half position."""
    cout, cin = w.shape[0], w.shape[1]
    G = _WINO4_G.to(w.device)
    u = torch.einsum("ak,ockl,bl->ocab", G, w.detach().to(torch.float64), G).to(torch.float32)     # [Cout, Cin, 6, 6]
    c4, nt = (cin + 3) // 4, (cout + 63) // 64
    full = torch.zeros(nt * 64, c4 * 4, 36, dtype=torch.float32, device=w.device)
    full[:cout, :cin] = u.reshape(cout, cin, 36)
    v = full.reshape(nt, 2, 32, c4, 2, 2, 36)                    # [nt, half, li, c4, lh, e2, pos]
    # full positions: [c4, nt, wave, i, lh, li, half, e2]  (pos = 4 wave + i)
    fp = v[..., :32].reshape(nt, 2, 32, c4, 2, 2, 8, 4).permute(3, 0, 6, 7, 4, 2, 1, 5).reshape(c4, nt, 8, 4 * 64 * 4)
    # half positions 32 + k: wave = 2 k + half -> [c4, nt, k, half, lh, li, e2]
    hp = v[..., 32:].permute(3, 0, 6, 1, 4, 2, 5).reshape(c4, nt, 8, 64 * 2)
    return torch.cat([fp, hp], dim=3).contiguous()


def pack_matrix(w):
    """[Cout, Cin] (nn.Linear / 1x1 conv orientation) -> [ceil(Cin/8)][roundup(Cout,64)][8]."""
    return pack_conv_weight(w.reshape(w.shape[0], w.shape[1], 1, 1))


class WeightStore:
    """Packed copies of module parameters with stable device addresses.

    `refresh()` re-packs an entry in place when any of its source parameters changed
    (`Tensor._version`, bumped by optimizer steps / load_state_dict), so programs and
    captured graphs keep valid pointers.  Every entry also records its provenance
    (`meta[id(packed)]`: the logical weight, and which rows of it belong to which
    parameter) -- the backward lowering uses it to build input-gradient weights and to
    aim weight-gradient kernels at the parameters' own `.grad` storage."""

    def __init__(self, device):
        self.device = device
        self.entries = []   # [packed, sources, fn, stamp]
        self.meta = {}

    def add(self, sources, fn, meta=None, recipe=None):
        """recipe: descriptors for the device-side re-pack (ssde_pack_weights); fn is the same packing in torch, used for
        the first fill, for CPU dry lowering, and as the cross-check of the device kernels in the tests."""
        lazy = recipe is not None and self._on_device() and os.environ.get("SSDE_TORCH_PACK", "0") != "1"
        with torch.no_grad():
            if lazy:
                # on the GPU the FIRST fill is the device re-pack too (ssde_pack_weights, four launches for the whole store on the
                # first refresh()): no torch einsum -- hence no rocBLAS / Tensile kernel -- anywhere on the product path.  The
                # torch packer only tells the shape (evaluated on meta tensors); zero padding comes from the zero fill.
                try:
                    shape = fn(*[torch.empty(s.shape, dtype=s.dtype, device="meta") for s in sources]).shape
                except Exception:                                 # noqa: BLE001  (an op without a meta kernel: shape from CPU zeros)
                    shape = fn(*[torch.zeros(s.shape, dtype=s.dtype) for s in sources]).shape
                packed = torch.zeros(tuple(shape), dtype=torch.float32, device=self.device)
            else:
                packed = fn(*[s.detach() for s in sources]).to(self.device).contiguous()
        self.entries.append([packed, list(sources), fn, None if lazy else self._stamp(sources), recipe])
        if meta is not None:
            self.meta[id(packed)] = meta
        self._tables = None
        return packed

    def _on_device(self):
        return self.device.type == "cuda" if isinstance(self.device, torch.device) else str(self.device).startswith("cuda")

    # -- typed registrations -------------------------------------------------------------------
    def conv3(self, param, cin_pad=None, cout_pad=None, wino=False):
        """[Cout, Cin, 3, 3] conv weight, optionally zero-padded to cin_pad / cout_pad channels; wino (see
        Lowering.wino_ok): 2 / True = packed for the Winograd F(2x2,3x3) kernel (G g G^T, conv_wino.hip), 4 = for the fused
        F(4x4,3x3) kernel (conv_wino4.hip), 6 = per lane for the
        register-fed matrix kernel (conv_wino4r.hip), 0 = for the direct one."""
        def logical(w):
            w = w.to(torch.float32)
            if cin_pad and w.shape[1] < cin_pad:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin_pad - w.shape[1]))
            if cout_pad and w.shape[0] < cout_pad:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_pad - w.shape[0]))
            return w
        cout_l, cin_l = max(param.shape[0], cout_pad or 0), max(param.shape[1], cin_pad or 0)
        meta = dict(kind="conv3", sources=[param], logical=logical, dims=(cout_l, cin_l),
                    parts=[dict(param=param, row0=0, rows=param.shape[0], transpose=False)], cin_store=param.shape[1])
        pack = pack_wino4r_weight if wino == 6 else pack_wino4_weight if wino == 4 else pack_wino_weight if wino else pack_conv_weight
        kind = L.PACK_WINO4R if wino == 6 else L.PACK_WINO4 if wino == 4 else L.PACK_WINO3 if wino else L.PACK_CONV3
        recipe = [dict(kind=kind, src=param, cout=param.shape[0], cin=param.shape[1], cout_l=cout_l, cin_l=cin_l, flags=0, n="dst")]
        return self.add([param], lambda w: pack(logical(w)), meta, recipe)

    def matrix(self, parts, cin_pad=None):
        """Rows-concatenated [Cout_i, Cin] matrices; parts = [(param, transpose)], transpose for NIN's [in, out]."""
        params = [p for p, _ in parts]

        def logical(*ws):
            ms = []
            for w, (_, tr) in zip(ws, parts):
                w = w.to(torch.float32)
                w = w.t() if tr else w.reshape(w.shape[0], -1)
                ms.append(w)
            m = torch.cat(ms, dim=0) if len(ms) > 1 else ms[0]
            if cin_pad and m.shape[1] < cin_pad:
                m = torch.nn.functional.pad(m, (0, cin_pad - m.shape[1]))
            return m
        rows, desc = 0, []
        for p_, tr in parts:
            n_rows = p_.shape[1] if tr else p_.shape[0]
            cin = p_.shape[0] if tr else int(np.prod(p_.shape[1:]))
            desc.append(dict(param=p_, row0=rows, rows=n_rows, transpose=tr))
            rows += n_rows
        meta = dict(kind="matrix", sources=params, logical=logical, parts=desc, cin_store=cin,
                    dims=(rows, max(cin, cin_pad or 0)))
        recipe = [dict(kind=L.PACK_MATRIX, src=d["param"], cout=d["param"].shape[0], cin=int(np.prod(d["param"].shape[1:])),
                       cout_l=rows, cin_l=0, flags=int(d["transpose"]), r_off=d["row0"], c_off=0, n="src") for d in desc]
        return self.add(params, lambda *ws: pack_matrix(logical(*ws)), meta, recipe)

    def vector(self, params, mode="cat", pad_to=None):
        """Per-channel vectors: biases (concatenated, or summed when two convolutions share one epilogue), GroupNorm affine."""
        def fn(*vs):
            vs = [v.to(torch.float32) for v in vs]
            x = torch.cat(vs) if mode == "cat" else sum(vs[1:], vs[0])
            if pad_to and x.numel() < pad_to:
                x = torch.nn.functional.pad(x, (0, pad_to - x.numel()))
            return x.clone()
        off, desc = 0, []
        for p_ in params:
            desc.append(dict(param=p_, off=0 if mode == "sum" else off, n=p_.numel()))
            off += p_.numel()
        if mode == "sum":
            assert len(params) == 2
            recipe = [dict(kind=L.PACK_VECTOR, src=params[0], src2=params[1], r_off=0, n="src")]
        else:
            recipe = [dict(kind=L.PACK_VECTOR, src=d["param"], r_off=d["off"], n="src") for d in desc]
        return self.add(params, fn, dict(kind="vector", mode=mode, sources=list(params), parts=desc), recipe)

    def derived(self, packed, fn, tag):
        """The input-gradient packing of an existing entry's logical weight: fn(logical) -> packed, with tag
        'dgrad' (direct conv / matrix), 'dgrad_wino', 'dgrad_wino4' or 'dgrad_wino4r'."""
        meta = self.meta[id(packed)]
        key = (id(packed), tag)
        if key not in self.meta:
            cout_l, cin_l = meta["dims"]
            if meta["kind"] == "conv3":
                p_ = meta["sources"][0]
                kind = {"dgrad_wino": L.PACK_WINO3, "dgrad_wino4": L.PACK_WINO4, "dgrad_wino4r": L.PACK_WINO4R}.get(tag, L.PACK_CONV3)
                recipe = [dict(kind=kind, src=p_, cout=p_.shape[0], cin=p_.shape[1],
                               cout_l=cin_l, cin_l=cout_l, flags=1, n="dst")]
            else:
                # D = M^T: part rows become column ranges; a transposed (NIN) part is read straight, a plain one swapped
                recipe = [dict(kind=L.PACK_MATRIX, src=d["param"], cout=d["param"].shape[0], cin=int(np.prod(d["param"].shape[1:])),
                               cout_l=cin_l, cin_l=0, flags=int(not d["transpose"]), r_off=0, c_off=d["row0"], n="src")
                          for d in meta["parts"]]
            self.meta[key] = self.add(meta["sources"], lambda *ws: fn(meta["logical"](*ws)), None, recipe)
        return self.meta[key]

    # -- device-side re-pack -------------------------------------------------------------------------
    def _build_tables(self):
        descs = {}
        for e in self.entries:
            packed, recipe = e[0], e[4]
            if recipe is None:
                continue
            for r in recipe:
                d = L.PackDesc()
                src = r["src"]
                d.src, d.dst, d.kind = src.data_ptr(), packed.data_ptr(), r["kind"]
                d.src2 = r["src2"].data_ptr() if r.get("src2") is not None else None
                d.cout, d.cin = r.get("cout", 0), r.get("cin", 0)
                d.cout_l, d.cin_l, d.flags = r.get("cout_l", 0), r.get("cin_l", 0), r.get("flags", 0)
                d.r_off, d.c_off = r.get("r_off", 0), r.get("c_off", 0)
                d.n = packed.numel() if r["n"] == "dst" else src.numel()
                descs.setdefault(r["kind"], []).append(d)
        tables = []
        for kind, ds in sorted(descs.items()):
            arr = (L.PackDesc * len(ds))(*ds)
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            args = L.PackArgs()
            args.table, args.count, args.kind, args.max_n = raw.data_ptr(), len(ds), kind, max(int(x.n) for x in ds)
            tables.append((args, raw))
        stamp = tuple(s.data_ptr() for e in self.entries if e[4] is not None for s in e[1])
        self._tables = (tables, stamp)

    def device_refresh(self):
        """Re-pack every entry that has a recipe with four launches (one per kind); returns False when nothing has one."""
        from . import hipops
        if getattr(self, "_tables", None) is None or \
                self._tables[1] != tuple(s.data_ptr() for e in self.entries if e[4] is not None for s in e[1]):
            self._build_tables()
        lib = L.load()
        for args, _ in self._tables[0]:
            L.check(lib.ssde_pack_weights(C.byref(args), hipops._stream()), "ssde_pack_weights")
        return bool(self._tables[0])

    @staticmethod
    def _stamp(sources):
        # `_ssde_flat.generation` counts whole-buffer writes that bypass the per-tensor version counters
        # (backward.FlatParams: fused optimizer step, EMA swap)
        return tuple((s.data_ptr(), s._version, getattr(getattr(s, "_ssde_flat", None), "generation", 0)) for s in sources)

    def mark_fresh(self):
        """The caller has just re-packed every entry that has a device recipe itself (the re-pack launches inside a captured
        training step): record the sources' stamps so that the next refresh() does not do it again; entries without a recipe
        are re-packed here."""
        for e in self.entries:
            st = self._stamp(e[1])
            if e[4] is None and st != e[3]:
                with torch.no_grad():
                    e[0].copy_(e[2](*[s.detach() for s in e[1]]).to(self.device))
            e[3] = st

    def refresh(self, force=False, on_device=None):
        """Bring packed copies up to date.  force=True (after the fused optimizer wrote the flat parameter buffer behind
        torch's back) re-packs everything: with the device kernels when the library can run here, else in torch."""
        if on_device is None:
            on_device = self._on_device()
        stamps = [self._stamp(e[1]) for e in self.entries]
        stale = [force or st != e[3] for e, st in zip(self.entries, stamps)]
        # any stale entry that has a device recipe (first use, optimizer step, checkpoint load, EMA swap, a user's in-place
        # edit): four device launches re-pack the whole store -- cheaper than one torch packer, and no rocBLAS kernel
        done_on_device = on_device and any(old and e[4] is not None for e, old in zip(self.entries, stale)) and self.device_refresh()
        for e, st, old in zip(self.entries, stamps, stale):
            if old and not (done_on_device and e[4] is not None):
                with torch.no_grad():
                    e[0].copy_(e[2](*[s.detach() for s in e[1]]).to(self.device))
            e[3] = st


# --------------------------------------------------------------------------- lowering helpers
def _src(t, c, t2=None, c2=0, pro=L.PRO_NONE, gn=None, drop=None):
    d = dict(p0=t, p1=t2, c0=c, c1=c2, pro_mode=pro, gn_groups=0, gn_mean=None, gn_rstd=None, gn_gamma=None, gn_beta=None,
             drop_thresh=0, drop_scale=1.0, drop_seed=None, drop_salt=0)
    if gn is not None:
        d.update(gn_groups=gn["groups"], gn_mean=gn["mean"], gn_rstd=gn["rstd"], gn_gamma=gn["gamma"], gn_beta=gn["beta"])
        d["_gn"] = gn        # (lowering-time only: the statistics may still be partials, Lowering._materialize_stats)
    if drop is not None:
        p, seed_t, salt = drop
        d.update(drop_thresh=min(int(round(p * 2.0 ** 32)), 2 ** 32 - 1), drop_scale=1.0 / (1.0 - p), drop_seed=seed_t,
                 drop_salt=salt & 0xFFFFFFFF)
    return d


_NOSRC = dict(p0=None, p1=None, c0=0, c1=0, pro_mode=0, gn_groups=0, gn_mean=None, gn_rstd=None, gn_gamma=None, gn_beta=None,
              drop_thresh=0, drop_scale=1.0, drop_seed=None, drop_salt=0)


def fir_taps(k, gain=1.0):
    """_setup_kernel (models/up_or_down_sampling.py:181-188): normalised outer product."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k = k / np.sum(k)
    return (k * gain).astype(np.float32)


class Lowering:
    """Shared op emitters (used by the U-Net engine and by per-op wrappers in tests/op)."""

    def __init__(self, builder, weights, batch):
        self.b, self.w, self.n = builder, weights, batch
        self.parts = {}      # id(activation Buf) -> (partials Buf, slices per image, channels): written by its producer's epilogue
        # compute units of the device the program will run on: the kernel choice below asks "does this launch fill the
        # chip?" (256 on the MI355X; also the value used for CPU dry lowering).  SSDE_NUM_CUS overrides (tests)
        dev = getattr(builder, "device", None)
        self.cus = 256
        if os.environ.get("SSDE_NUM_CUS"):
            self.cus = int(os.environ["SSDE_NUM_CUS"])
        elif dev is not None and torch.device(dev).type == "cuda" and torch.cuda.is_available():
            self.cus = int(torch.cuda.get_device_properties(torch.device(dev)).multi_processor_count)

    def _gn_slices(self, fields):
        """Slices per image of the GroupNorm partials the LAST launch of this conv spec would write (0: not available)."""
        if os.environ.get("SSDE_GN_FUSE", "1") == "0":
            return 0
        sub = _expand(L.OP_CONV, fields, 0, 0.0)[-1][0]
        a = L.ConvArgs()
        dummy = 0x1000                                            # the planner only tests pointers for presence
        for src_name in ("main", "aux"):
            src, dst = sub[src_name], getattr(a, src_name)
            dst.p0 = dummy if src["p0"] is not None else None
            dst.p1 = dummy if src["p1"] is not None else None
            dst.c0, dst.c1, dst.pro_mode, dst.gn_groups = src["c0"], src["c1"], src["pro_mode"], src["gn_groups"]
            if src["gn_groups"]:
                dst.gn_mean = dst.gn_rstd = dst.gn_gamma = dst.gn_beta = dummy
            if src["drop_thresh"]:
                dst.drop_thresh, dst.drop_seed = src["drop_thresh"], dummy
        a.w_main = dummy if sub["w_main"] is not None else None
        a.w_aux = dummy if sub["w_aux"] is not None else None
        for k in ("n", "h_in", "w_in", "h_out", "w_out", "c_out", "ksize", "stride", "pad", "tile"):
            setattr(a, k, sub[k])
        a.dst = dummy
        a.flags = sub.get("flags", L.conv_route_flags())
        return int(L.load().ssde_conv_gn_slices(C.byref(a)))

    # -- GroupNorm statistics of a (possibly concatenated) NHWC source
    def gn_stats(self, t, c, hw, gn_module, t2=None, c2=0):
        ctot = c + c2
        groups = gn_module.num_groups
        mean = self.b.buf(self.n, groups, name="gn_mean")
        rstd = self.b.buf(self.n, groups, name="gn_rstd")
        p0, p1 = self.parts.get(id(t)), (self.parts.get(id(t2)) if t2 is not None else None)
        if p0 is not None and p0[2] == c and (t2 is None or (p1 is not None and p1[2] == c2)) and (ctot // groups) % 4 == 0:
            # the producers' epilogues already reduced this tensor: merge their partials (no pass over the activations).
            # WHO merges is decided by the first consumer (_materialize_stats): the transform pass of a two-kernel F(4x4,3x3)
            # convolution does it for itself under its pixel loads (ssde_conv_args.gn_in_part0, ABI 10); everybody else gets
            # the ssde_gn_finalize launch in front
            fin = dict(part0=p0[0], part1=p1[0] if p1 else None, c0=c, c1=c2, slices0=p0[1], slices1=p1[1] if p1 else 0,
                       n=self.n, groups=groups, eps=float(gn_module.eps), mean=mean, rstd=rstd)
            gamma = self.w.vector([gn_module.weight])
            beta = self.w.vector([gn_module.bias])
            assert ctot == gn_module.num_channels
            gn = dict(groups=groups, mean=mean, rstd=rstd, gamma=gamma, beta=beta, pending=fin)
            # (Measured level on the sampler -- profiles/r6_gn_merge_in_transform_pass_ab.txt: the pass sits at its 256-register
            #  limit and pays for the merge what the 59 launches cost -- so the consumer-side merge is opt-in and the default
            #  stays the finalize launch, now a 16-lane-team kernel: SSDE_GN_MERGE_IN_CONSUMER=1)
            if os.environ.get("SSDE_GN_MERGE_IN_CONSUMER", "0") != "1":
                self._materialize_stats(gn)
            return gn
        slices = max(1, min(int(math.ceil(256 / self.n)), hw // 64)) if hw >= 128 else 1
        scratch = self.b.buf(self.n * slices * groups * 2, name="gn_scratch") if slices > 1 else None
        self.b.add(L.OP_GN_STATS, dict(p0=t, p1=t2, c0=c, c1=c2, n=self.n, hw=hw, groups=groups, eps=float(gn_module.eps),
                                       mean=mean, rstd=rstd, scratch=scratch, slices=slices), FC_GN)
        gamma = self.w.vector([gn_module.weight])
        beta = self.w.vector([gn_module.bias])
        assert ctot == gn_module.num_channels
        return dict(groups=groups, mean=mean, rstd=rstd, gamma=gamma, beta=beta)

    def _materialize_stats(self, gn, fields=None):
        """The statistics of `gn` are about to be read: if they are still the producers' partials, either hand the merge to the
        consuming launch itself (fields: the spec of a two-kernel F(4x4,3x3) convolution whose main source they normalise) or
        emit the finalize launch now."""
        fin = gn.get("pending") if gn else None
        if fin is None:
            return
        gn["pending"] = None
        if fields is not None:
            fields.update(gn_in_part0=fin["part0"], gn_in_part1=fin["part1"], gn_in_slices0=fin["slices0"],
                          gn_in_slices1=fin["slices1"], gn_in_eps=fin["eps"])
        else:
            self.b.add(L.OP_GN_FINALIZE, fin, FC_GN)

    def conv(self, dst, h_out, w_out, c_out, main=None, w_main=None, h_in=0, w_in=0, stride=1, pad=1,
             aux=None, w_aux=None, bias=None, chan_add=None, chan_add_ld=0, resid=None, scale=1.0, tile=L.TILE_AUTO,
             resid_post=0, wino=False, stats=False):
        """wino (2 / True, 4 or 6): w_main is Winograd-packed (see wino_ok); a fused 1x1 source then runs as a second launch.
        stats=True: dst feeds a GroupNorm later -- when the launch plan allows it (ssde_conv_gn_slices) the epilogue
        also writes the tensor's partial statistics and gn_stats() turns into a finalize of a few thousand floats."""
        split_tmp = None
        if wino:
            assert main is not None and stride == 1 and pad == 1 and (h_in, w_in) == (h_out, w_out)
            tile = L.TILE_WINOGRAD4R if wino == 6 else L.TILE_WINOGRAD4 if wino == 4 else L.TILE_WINOGRAD
            if aux is not None:
                split_tmp = self.b.buf(self.n, h_out, w_out, c_out, name="wino_tmp")
        px = self.n * h_out * w_out
        flops = 0.0
        if main is not None:
            flops += 2.0 * px * 9 * (main["c0"] + main["c1"]) * c_out
        if aux is not None:
            flops += 2.0 * px * (aux["c0"] + aux["c1"]) * c_out
        fields = dict(
            main=main if main is not None else _NOSRC, aux=aux if aux is not None else _NOSRC,
            w_main=w_main, w_aux=w_aux, n=self.n, h_in=h_in, w_in=w_in, h_out=h_out, w_out=w_out, c_out=c_out,
            ksize=3 if main is not None else 0, stride=stride, pad=pad, tile=tile, bias=bias, chan_add=chan_add,
            chan_add_ld=chan_add_ld, resid_post=resid_post, resid=resid, out_scale=float(scale), dst=dst, gn_part=None,
            wino_v=None, _split_tmp=split_tmp)
        if wino in (4, 6):
            # the transformed input B^T pro(x) B in HBM (2.25x the input): the two-kernel form (wino4_xform.hip + conv_wino4r.hip) cannot do without
            # it; in a training forward the one-kernel form leaves it behind as a by-product when the layer's weight gradient
            # takes the F(4x4,3x3) route (ssde_conv_args.wino_v -> ssde_wgrad_args.v_pre, backward.TrainEngine._bwd_branch:
            # alive until that weight gradient has been enqueued)
            ctot = main["c0"] + main["c1"]
            v_floats = 36 * self.n * (h_out // 4) * (w_out // 4) * ctot
            takes = getattr(self, "emit_wino_v", False) and v_floats * 4 < 2 ** 32 and self._wgrad_takes_wino4(fields)
            if tile == L.TILE_WINOGRAD4R or takes:
                fields["wino_v"] = self.b.buf(v_floats, name="wino_v")
                fields["_v_for_wgrad"] = bool(takes)
        # GroupNorm statistics that are still partials (gn_stats): the transform pass of the two-kernel form merges those of
        # its main source itself; any other reader gets the finalize launch in front
        if aux is not None:
            self._materialize_stats(aux.get("_gn"))
        if main is not None:
            in_pass = tile == L.TILE_WINOGRAD4R and h_in * w_in >= 64
            self._materialize_stats(main.get("_gn"), fields if in_pass else None)
        if stats and isinstance(dst, Buf):
            slices = self._gn_slices(fields)
            if slices > 0:
                part = self.b.buf(self.n, slices, c_out // 4, 3, name="gn_part")
                fields["gn_part"] = part
                self.parts[id(dst)] = (part, slices, c_out)
        self.b.add(L.OP_CONV, fields, FC_CONV3 if main is not None else FC_CONV1, flops)

    def _wino4_two_kernels(self, h, w, c_out, c_in):
        """F(4x4,3x3) as a transform pass (wino4_xform.hip) + the register-fed matrix kernel (conv_wino4r.hip) instead of the one
        fused kernel?  The pass costs one more read of x and 2.25x of it written and read; the matrix kernel saves the prologue
        and transform VALU work that serialises with fp32 MFMAs in EVERY 64-cout workgroup of a pixel tile, and runs at 0.6-0.9
        of the matrix peak against the fused kernel's 0.40-0.45.  Measured at batch 256 (profiles/
        r5_wino4r_v4_one_v_load_per_workgroup.txt, fused -> pair): 256->256 @16x16 0.28 -> 0.22 ms, 512->256 @16x16 0.50 -> 0.40,
        256->128 @32x32 0.61 -> 0.55, 128->128 @32x32 0.350 -> 0.326, 384->128 @32x32 0.77 -> 0.85 (three times the input for
        two cout tiles: the one shape where the pair loses in isolation).  In the networks (profiles/r5_two_kernel_rule_ab.txt,
        CIFAR sampler, images/s on one box): four cout tiles up 4.68-4.70, + input at most twice the output 4.71-4.73, everywhere
        4.75 -- so: wherever F(4x4,3x3) runs and the kernel's shape limits allow.  In a training program the pass's output is
        also the F(4x4,3x3) weight gradient's input (v_pre).
        SSDE_WINO4_TWO: 0 = never, 1 = from four cout tiles up or input <= 2 x output, 3 = round 4's rule (four cout tiles)."""
        mode = os.environ.get("SSDE_WINO4_TWO", "2")
        if mode == "0" or 36 * self.n * (h // 4) * (w // 4) * c_in * 4 >= 2 ** 32 or c_in % 8 != 0:
            return False
        if mode == "3":
            return c_out >= 256
        if mode == "1":
            return c_out >= 256 or c_in <= 2 * c_out or bool(getattr(self, "emit_wino_v", False))
        return True

    def _wgrad_takes_wino4(self, f):
        """Would ssde_conv_wgrad run the weight gradient of this forward conv on the F(4x4,3x3) path?  (shape-only query with
        the arguments backward.TrainEngine._bwd_branch will pass: g = d dst, one part covering the whole weight)"""
        if os.environ.get("SSDE_WINO_V_FROM_FORWARD", "1") == "0":
            return False
        meta = self.w.meta.get(id(f["w_main"]))
        if meta is None or len(meta["parts"]) != 1 or meta["parts"][0]["transpose"]:
            return False
        a = L.WgradArgs()
        s = f["main"]
        a.src.c0, a.src.c1, a.src.pro_mode, a.src.gn_groups = s["c0"], s["c1"], s["pro_mode"], s["gn_groups"]
        a.g_ld, a.g_off = f["c_out"], meta["parts"][0]["row0"]
        a.n, a.h_in, a.w_in, a.h_out, a.w_out = f["n"], f["h_in"], f["w_in"], f["h_out"], f["w_out"]
        a.c_out, a.ksize, a.stride, a.pad = meta["parts"][0]["rows"], 3, f["stride"], f["pad"]
        a.cin_store, a.transpose_out = meta["cin_store"], 0
        a.flags = L.wgrad_route_flags()
        return bool(L.load().ssde_wgrad_wants_winograd4(C.byref(a)))

    def wino_ok(self, h, w, c_out, c_in):
        """Which 3x3 / stride 1 kernel a layer gets: 0 = direct, 2 = Winograd F(2x2,3x3), 4 = F(4x4,3x3) in one fused kernel
        (conv_wino4.hip), 6 = F(4x4,3x3) as a transform pass + the register-fed matrix kernel (conv_wino4r.hip; _wino4_two_kernels
        decides between 4 and 6).
        Winograd pays when the matrix pipe is the bound: enough channels to fill the 64-cout tile, and enough workgroups
        to cover the 256 CUs (F(2x2,3x3): 64 tiles x 64 couts per workgroup -- measured x1.2-1.5 over the direct kernel
        from 8x8 up at batch 256, x0.5 at 4x4 where only 64 workgroups exist).  F(4x4,3x3) does 1.78x less matrix work
        again and is 15-23 % faster than F(2x2,3x3) from 16x16 maps up (profiles/r2_wino4_v3_interleaved.txt); its
        workgroups cover 32 tiles = 512 pixels, so 8x8 maps at batch 256 give 128 of them: there the register-fed matrix kernel
        splits its reduction over two workgroups per tile (four at batch 128) and wins from 256 input channels up (round 5, see
        below); with fewer channels, and on 4x4 maps, F(2x2,3x3) / the direct kernel stay.
        Rounding: ~5x coarser than the direct form, 2.5e-6 .. 1.3e-5 on the whole network against the 1e-4 the parity
        tests allow (tools/experiments/wino43_error_budget.py).
        SSDE_WINOGRAD: 0 = direct (bitwise fmaf-chain) kernel everywhere, 1 = this heuristic (default), 2 = F(2x2,3x3)
        wherever it is legal, 3 = the heuristic without F(4x4,3x3), 4 = F(4x4,3x3) wherever it is legal."""
        mode = os.environ.get("SSDE_WINOGRAD", "1")
        if mode == "0":
            return 0
        legal2 = h % 2 == 0 and w % 2 == 0 and h >= 8 and w >= 8 and c_out >= 32 and c_in >= 8 and c_in % 8 == 0
        legal4 = h % 4 == 0 and w % 4 == 0 and h >= 8 and w >= 8 and c_out >= 32 and c_in >= 8 and c_in % 4 == 0
        four = lambda: 6 if self._wino4_two_kernels(h, w, c_out, c_in) else 4  # noqa: E731
        if mode == "4" and legal4:
            return four()
        if mode == "2" or mode == "4":
            return 2 if legal2 else 0
        n_tiles = -(-c_out // 64)
        # (one workgroup per CU: F(4x4,3x3) workgroups own a whole CU's LDS and registers)
        if mode != "3" and legal4 and h >= 16 and w >= 16 and -(-(self.n * h * w) // 512) * n_tiles >= self.cus:
            return four()
        # Fewer tiles than that (8x8 maps at batch 256: 128 tiles of 8 images x 64 couts): the kernel splits its reduction
        # over 2 or 4 workgroups per tile (conv_wino4.hip, ssde_conv_wino4_splits -- the same rule).  Measured
        # (profiles/r3_wino4_split_reduction_ab.txt): 17-40 % over the unsplit kernel, but only level with F(2x2,3x3) on
        # 8x8 maps at batch 256 and 128 (0.123 / 0.184 ms against 0.110 / 0.190 ms; the PC iteration 58.1 against 57.0 ms),
        # so the heuristic takes it only on request: SSDE_W4_SPLIT_MIN_WGS=<workgroups after the split, e.g. 192>
        if mode != "3" and legal4 and "SSDE_W4_SPLIT_MIN_WGS" in os.environ:
            wgs4 = -(-(-(-(self.n * h * w) // 512)) // 8) * 8 * n_tiles
            splits = 1
            if c_out % 4 == 0 and os.environ.get("SSDE_CONV_KSPLIT", "1") != "0":
                splits = 4 if wgs4 <= self.cus // 4 and c_in >= 256 else 2 if wgs4 <= self.cus // 2 and c_in >= 128 else 1
            if splits > 1 and wgs4 * splits >= int(os.environ["SSDE_W4_SPLIT_MIN_WGS"]):
                return 4
        # The register-fed matrix kernel splits its reduction too (conv_wino4r.hip, ssde_conv_wino4r_splits -- the same rule
        # here): its channel stage is ~2450 cycles against the fused kernel's ~4600, so where the shares fill exactly one round of
        # workgroups, transform pass + split matrix kernel beat F(2x2,3x3).  TWO shares -- the 8x8 maps at batch 256: 128 tiles ->
        # 256 workgroups: 256->256 0.077 against 0.109 ms, 512->256 0.112 against 0.193 (profiles/r5_wino4r_split_8x8.txt); with 128
        # input channels (16 stages per share) the hand-over costs what the split saves and F(2x2,3x3) stays.
        # SSDE_W4R_SPLIT=0 keeps round 4's choice.
        if mode != "3" and legal4 and c_in % 16 == 0 and c_in >= 256 and c_out % 4 == 0 \
                and os.environ.get("SSDE_W4R_SPLIT", "1") != "0" and os.environ.get("SSDE_CONV_KSPLIT", "1") != "0":
            wgs4 = -(-(-(-(self.n * h * w) // 512)) // 8) * 8 * n_tiles
            if self.cus // 2 < wgs4 * 2 <= self.cus and self._wino4_two_kernels(h, w, c_out, c_in):
                return 6
            # (four shares filling one round -- the 8x8 maps at batch 128, the training step: 256->256 0.102 -> 0.063 ms, 512->256 0.184 -> 0.082,
            #  step 0.0548 -> 0.0530 s, profiles/r5_wino4r_split_8x8_batch128.txt, r5_wino4r_split4_train_ab.txt; SSDE_W4R_SPLIT4=0 switches it off)
            if os.environ.get("SSDE_W4R_SPLIT4", "1") == "1" and c_in % 32 == 0 and self.cus // 2 < wgs4 * 4 <= self.cus \
                    and self._wino4_two_kernels(h, w, c_out, c_in):
                return 6
        # tools/heuristic_sweep.py (profiles/r2_heuristic_sweep.txt): F(4x4,3x3) wins from one of its workgroups per CU (256),
        # F(2x2,3x3) over the direct kernel from half a workgroup per CU (128; by 2-6 %; at 64 the direct kernel is 1.5x
        # faster).  tools/batch_sweep.py: the sampler at batch 16 / 64 / 256 under this rule (profiles/r4_batch_sweep.txt)
        return 2 if legal2 and -(-(self.n * h * w) // 256) * n_tiles >= self.cus // 2 else 0

    def upfirdn(self, src, n_ch, h_in, w_in, taps, up=1, down=1, pad=(0, 0), name="fir"):
        kh, kw = taps.shape
        h_out = (h_in * up + pad[0] + pad[1] - kh) // down + 1
        w_out = (w_in * up + pad[0] + pad[1] - kw) // down + 1
        dst = self.b.buf(self.n, h_out, w_out, n_ch, name=name)
        self._materialize_stats(src.get("_gn"))
        k16 = [0.0] * 16
        for i, v in enumerate(taps.reshape(-1).tolist()):
            k16[i] = float(v)
        self.b.add(L.OP_UPFIRDN, dict(src=src, n=self.n, h_in=h_in, w_in=w_in, c=n_ch, h_out=h_out, w_out=w_out,
                                      up=up, down=down, pad0=pad[0], pad1=pad[1], kh=kh, kw=kw, k=k16, dst=dst), FC_FIR)
        return dst, h_out, w_out


# --------------------------------------------------------------------------- the U-Net engine
class UNetEngine:
    """Static program for NCSNpp.forward at a fixed (batch, H, W)."""

    def __init__(self, model, batch, height, width, device, vp_score=False, train=False, input_grad=False,
                 finalize=True):
        L.load()
        # train=True: Dropout_0 of every residual block is live (layerspp.py:265) and a backward program is
        # appended by score_sde_pytorch_amd.backward.TrainEngine; input_grad adds d out / d x to it.
        self.train, self.input_grad = train, input_grad
        # device 'cpu' is accepted for DRY lowering only (plan validation, FLOP census in the CPU tests);
        # running a program needs the MI355X.
        self.model, self.n, self.h, self.w, self.device = model, batch, height, width, device
        cfg = model.config
        self.cfg = cfg
        self.b = ProgramBuilder(device)
        self.weights = WeightStore(device)
        self.low = Lowering(self.b, self.weights, batch)
        # a training program with parameter gradients (backward.TrainEngine sets param_grads before lowering): the forward
        # F(4x4,3x3) launches leave their transformed input behind for the weight gradients
        self.low.emit_wino_v = bool(getattr(self, "param_grads", False))
        self.channels = model.channels
        # static I/O (addresses are baked into the program / graph)
        self.x_in = self.b.buf(batch, self.channels, height, width, name="x_in", persistent=True)
        self.cond = self.b.buf(batch, name="cond", persistent=True)
        # discrete-label models index the sigma table for scale_by_sigma (ncsnpp.py:245,377-379)
        self.sig = self.b.buf(batch, name="sigma", persistent=True) \
            if (model.embedding_type == "positional" and cfg.model.scale_by_sigma) else self.cond
        self.out = self.b.buf(batch, self.channels, height, width, name="out", persistent=True)
        # vp_score: emit score = -h / std[n] (models/utils.py:159) instead of the raw network output
        self.vp_score = vp_score
        self.std = self.b.buf(batch, name="std", persistent=True) if vp_score else None
        if vp_score and cfg.model.scale_by_sigma:
            raise NotImplementedError("scale_by_sigma together with a VP score head is not lowered")
        self.drop_seed = None
        if train and float(cfg.model.dropout) > 0:
            self.drop_seed = torch.zeros(1, dtype=torch.int32, device=device)
            self.b.tensor(self.drop_seed)
        self._n_res = 0
        self._lower()
        if finalize:
            self.program = self.b.finalize()

    @property
    def cond_only_ops(self):
        """How many leading ops of `program` read nothing but the noise level (embedding, temb MLP, all Dense_0 projections):
        a caller that evaluates the network again at the SAME noise level may skip them -- their outputs are persistent."""
        return int(self.program.spec_start[getattr(self, "_cond_specs", 0)])

    # ------------------------------------------------------------------ lowering
    def _lower(self):
        model, b, low, n = self.model, self.b, self.low, self.n
        mods = list(model.all_modules)
        nf = model.nf
        idx = 0
        skip_scale = INV_SQRT2 if model.skip_rescale else 1.0
        fir, fk = model.fir, model.fir_kernel

        # ---- time embedding (ncsnpp.py:236-257)
        if model.embedding_type == "fourier":
            emb_dim = 2 * nf
            table = self.weights.add([mods[idx].W], lambda w: w.to(torch.float32).clone()); idx += 1
            kind = 0
        else:
            emb_dim = nf
            half = nf // 2
            e = math.log(10000) / (half - 1)
            freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -e)   # layers.py:519-521
            table = b.tensor(freqs.to(self.device))
            kind = 1
        # (the conditioning chain -- embedding, the two Linear layers, every Dense_0 -- depends on the noise level only: its
        #  buffers keep their own storage, so that a program which evaluates the network several times at ONE noise level -- a
        #  PC iteration: corrector and predictor -- can run the chain once: cond_only_ops, pc_engine._assemble)
        emb = b.buf(n, emb_dim, name="emb", persistent=True)
        self._emb = emb
        b.add(L.OP_EMBED, dict(cond=self.cond, w=table, dst=emb, n=n, dim=emb_dim, kind=kind))
        temb = None
        if model.conditional:
            lin0, lin1 = mods[idx], mods[idx + 1]; idx += 2
            t0 = b.buf(n, 4 * nf, name="temb0", persistent=True)
            low.conv(t0, 1, 1, 4 * nf, aux=_src(emb, emb_dim), w_aux=self.weights.matrix([(lin0.weight, False)]),
                     bias=self.weights.vector([lin0.bias]))
            temb = b.buf(n, 4 * nf, name="temb", persistent=True)
            low.conv(temb, 1, 1, 4 * nf, aux=_src(t0, 4 * nf, pro=L.PRO_SILU),
                     w_aux=self.weights.matrix([(lin1.weight, False)]),
                     bias=self.weights.vector([lin1.bias]))
            # every Dense_0(act(temb)) of every residual block as ONE GEMM (layerspp.py:263)
            dense = [m.Dense_0 for m in mods if getattr(m, "kind", "") == "res"]
            self._tproj_off, off = {}, 0
            for d in dense:
                self._tproj_off[id(d)] = off
                off += d.out_features
            self._tproj_ld = off
            wd = self.weights.matrix([(d.weight, False) for d in dense])
            bd = self.weights.vector([d.bias for d in dense])
            self._tproj = b.buf(n, off, name="tproj", persistent=True)
            low.conv(self._tproj, 1, 1, off, aux=_src(temb, 4 * nf, pro=L.PRO_SILU), w_aux=wd, bias=bd)
        self._cond_specs = len(b.specs)            # the leading specs that read nothing but the noise level

        # ---- input boundary: NCHW -> NHWC (C padded to 4), 2x-1 for un-centred data (ncsnpp.py:259-261)
        H, W = self.h, self.w
        cpad = 4
        x0 = b.buf(n, H, W, cpad, name="x_nhwc")
        self._x0 = x0
        a, sh = (1.0, 0.0) if self.cfg.data.centered else (2.0, -1.0)
        b.add(L.OP_TO_NHWC, dict(src=self.x_in, dst=x0, n=n, c=self.channels, h=H, w=W, c_pad=cpad, a=a, b=sh))
        pyr, pyr_c = (x0, cpad) if model.progressive_input != "none" else (None, 0)

        conv_in = mods[idx]; idx += 1
        h = b.buf(n, H, W, nf, name="h0")
        low.conv(h, H, W, nf, main=_src(x0, cpad), w_main=self._w3(conv_in, cin_pad=cpad), h_in=H, w_in=W,
                 bias=self._bias(conv_in), stats=True)
        hs = [(h, nf, H, W)]
        cur_c = nf

        def is_attn(res):
            return res in model.attn_resolutions

        # ---- encoder (ncsnpp.py:270-303)
        for lvl in range(model.num_resolutions):
            for _ in range(model.num_res_blocks):
                t, c, hh, ww = hs[-1]
                h, cur_c = self._res(mods[idx], t, c, hh, ww); idx += 1
                if is_attn(ww):
                    h = self._attn(mods[idx], h, cur_c, hh, ww); idx += 1
                hs.append((h, cur_c, hh, ww))
            if lvl != model.num_resolutions - 1:
                t, c, hh, ww = hs[-1]
                h, cur_c = self._res(mods[idx], t, c, hh, ww); idx += 1
                hh, ww = hh // 2, ww // 2
                if model.progressive_input == "input_skip":
                    taps = fir_taps(fk) if fir else fir_taps([1, 1])
                    pd = (1, 1) if fir else (0, 0)
                    pyr, _, _ = low.upfirdn(_src(pyr, pyr_c), pyr_c, hh * 2, ww * 2, taps, down=2, pad=pd, name="pyr_down")
                    comb = mods[idx]; idx += 1
                    if comb.method != "sum":
                        raise NotImplementedError("progressive_combine='cat' is not lowered (no shipped config uses it)")
                    hn = b.buf(n, hh, ww, cur_c, name="combine")
                    low.conv(hn, hh, ww, cur_c, aux=_src(pyr, pyr_c), w_aux=self._w1(comb.Conv_0, cin_pad=pyr_c),
                             bias=self._bias(comb.Conv_0), resid=h, scale=1.0, stats=True)
                    h = hn
                elif model.progressive_input == "residual":
                    down = mods[idx]; idx += 1
                    if not fir:
                        raise NotImplementedError("progressive_input='residual' without FIR is not lowered")
                    # conv_downsample_2d: FIR with pad (2,2) then stride-2 VALID conv (up_or_down_sampling.py:144-178)
                    pf, ph, pw = low.upfirdn(_src(pyr, pyr_c), pyr_c, hh * 2, ww * 2, fir_taps(fk), pad=(2, 2), name="pyr_fir")
                    hn = b.buf(n, hh, ww, cur_c, name="pyr")
                    low.conv(hn, hh, ww, cur_c, main=_src(pf, pyr_c), w_main=self._w3(down.Conv2d_0, cin_pad=pyr_c),
                             h_in=ph, w_in=pw, stride=2, pad=0, bias=self._bias(down.Conv2d_0), resid=h, scale=skip_scale,
                             stats=True)
                    pyr, pyr_c, h = hn, cur_c, hn
                hs.append((h, cur_c, hh, ww))

        # ---- bottleneck (ncsnpp.py:305-311)
        t, c, hh, ww = hs[-1]
        h, cur_c = self._res(mods[idx], t, c, hh, ww); idx += 1
        h = self._attn(mods[idx], h, cur_c, hh, ww); idx += 1
        h, cur_c = self._res(mods[idx], h, cur_c, hh, ww); idx += 1

        # ---- decoder (ncsnpp.py:316-364)
        pyramid = None
        for lvl in reversed(range(model.num_resolutions)):
            for _ in range(model.num_res_blocks + 1):
                st, sc, sh_, sw_ = hs.pop()
                assert (sh_, sw_) == (hh, ww)
                h, cur_c = self._res(mods[idx], h, cur_c, hh, ww, t2=st, c2=sc); idx += 1
            if is_attn(ww):
                h = self._attn(mods[idx], h, cur_c, hh, ww); idx += 1
            if model.progressive == "output_skip":
                gn_m, conv_m = mods[idx], mods[idx + 1]; idx += 2
                up_pyr = None
                if pyramid is not None:
                    taps = fir_taps(fk, gain=4.0) if fir else fir_taps([1, 1], gain=4.0)
                    pd = (2, 1) if fir else (1, 0)
                    up_pyr, _, _ = low.upfirdn(_src(pyramid, 4), 4, hh // 2, ww // 2, taps, up=2, pad=pd, name="pyr_up")
                gn = low.gn_stats(h, cur_c, hh * ww, gn_m)
                pn = b.buf(n, hh, ww, 4, name="pyramid")
                low.conv(pn, hh, ww, 4, main=_src(h, cur_c, pro=L.PRO_GN_SILU, gn=gn), w_main=self._w3(conv_m, cout_pad=4),
                         h_in=hh, w_in=ww, bias=self._bias(conv_m, pad_to=4), resid=up_pyr)
                pyramid = pn
            if lvl != 0:
                h, cur_c = self._res(mods[idx], h, cur_c, hh, ww); idx += 1
                hh, ww = hh * 2, ww * 2
        assert not hs

        # ---- head (ncsnpp.py:368-379)
        if model.progressive == "output_skip":
            o = pyramid
        else:
            gn_m, conv_m = mods[idx], mods[idx + 1]; idx += 2
            gn = low.gn_stats(h, cur_c, hh * ww, gn_m)
            o = b.buf(n, hh, ww, 4, name="head")
            low.conv(o, hh, ww, 4, main=_src(h, cur_c, pro=L.PRO_GN_SILU, gn=gn), w_main=self._w3(conv_m, cout_pad=4),
                     h_in=hh, w_in=ww, bias=self._bias(conv_m, pad_to=4))
        assert idx == len(mods), (idx, len(mods))
        mode = 2 if self.vp_score else (1 if self.cfg.model.scale_by_sigma else 0)
        vec = self.std if mode == 2 else (self.sig if mode == 1 else None)
        b.add(L.OP_TO_NCHW, dict(src=o, dst=self.out, n=n, c=self.channels, h=hh, w=ww, c_src=4, mode=mode, v=vec))

    # -- packed parameter helpers
    def _w3(self, m, cin_pad=None, cout_pad=None, wino=False):
        return self.weights.conv3(m.weight, cin_pad, cout_pad, wino=wino)

    def _w1(self, m, cin_pad=None):
        return self.weights.matrix([(m.weight, False)], cin_pad=cin_pad)

    def _bias(self, m, pad_to=None):
        return self.weights.vector([m.bias], pad_to=pad_to)

    # -- ResnetBlockBigGANpp (layerspp.py:242-274)
    def _res(self, m, t, c, hh, ww, t2=None, c2=0):
        b, low, n = self.b, self.low, self.n
        cin, cout = c + c2, m.out_ch
        assert cin == m.in_ch, (cin, m.in_ch)
        scale = INV_SQRT2 if m.skip_rescale else 1.0
        gn0 = low.gn_stats(t, c, hh * ww, m.GroupNorm_0, t2, c2)
        chan_add, ld = None, 0
        if hasattr(m, "Dense_0"):
            chan_add, ld = (self._tproj, self._tproj_off[id(m.Dense_0)]), self._tproj_ld
        if m.up or m.down:
            assert t2 is None
            if m.up:
                taps = fir_taps(m.fir_kernel, gain=4.0) if m.fir else fir_taps([1, 1], gain=4.0)
                kw = dict(up=2, pad=(2, 1) if m.fir else (1, 0))
            else:
                taps = fir_taps(m.fir_kernel) if m.fir else fir_taps([1, 1])
                kw = dict(down=2, pad=(1, 1) if m.fir else (0, 0))
            hr, ho, wo = low.upfirdn(_src(t, c, pro=L.PRO_GN_SILU, gn=gn0), c, hh, ww, taps, name="res_h_rs", **kw)
            xr, _, _ = low.upfirdn(_src(t, c), c, hh, ww, taps, name="res_x_rs", **kw)
            main0, skip_t, skip_c, skip_t2, skip_c2 = _src(hr, c), xr, c, None, 0
            hh, ww = ho, wo
        else:
            main0 = _src(t, c, t2, c2, pro=L.PRO_GN_SILU, gn=gn0)
            skip_t, skip_c, skip_t2, skip_c2 = t, c, t2, c2
        h1 = b.buf(n, hh, ww, cout, name="res_h1")
        wino0 = low.wino_ok(hh, ww, cout, main0["c0"] + main0["c1"])
        wino1 = low.wino_ok(hh, ww, cout, cout)
        low.conv(h1, hh, ww, cout, main=main0, w_main=self._w3(m.Conv_0, wino=wino0), h_in=hh, w_in=ww, bias=self._bias(m.Conv_0),
                 chan_add=chan_add, chan_add_ld=ld, wino=wino0, stats=True)
        gn1 = low.gn_stats(h1, cout, hh * ww, m.GroupNorm_1)
        out = b.buf(n, hh, ww, cout, name="res_out")
        self._n_res += 1
        drop = (float(m.dropout), self.drop_seed, 0x9E3779B1 * self._n_res) if self.drop_seed is not None else None
        main1 = _src(h1, cout, pro=L.PRO_GN_SILU, gn=gn1, drop=drop)
        if hasattr(m, "Conv_2"):
            bsum = self.weights.vector([m.Conv_1.bias, m.Conv_2.bias], mode="sum")
            low.conv(out, hh, ww, cout, main=main1, w_main=self._w3(m.Conv_1, wino=wino1), h_in=hh, w_in=ww,
                     aux=_src(skip_t, skip_c, skip_t2, skip_c2), w_aux=self._w1(m.Conv_2), bias=bsum, scale=scale, wino=wino1,
                     stats=True)
        else:
            assert skip_t2 is None
            low.conv(out, hh, ww, cout, main=main1, w_main=self._w3(m.Conv_1, wino=wino1), h_in=hh, w_in=ww,
                     bias=self._bias(m.Conv_1), resid=skip_t, scale=scale, wino=wino1, stats=True)
        return out, cout

    # -- AttnBlockpp (layerspp.py:75-91)
    def _attn(self, m, t, c, hh, ww):
        b, low, n = self.b, self.low, self.n
        gn = low.gn_stats(t, c, hh * ww, m.GroupNorm_0)
        wqkv = self.weights.matrix([(m.NIN_0.W, True), (m.NIN_1.W, True), (m.NIN_2.W, True)])
        bqkv = self.weights.vector([m.NIN_0.b, m.NIN_1.b, m.NIN_2.b])
        qkv = b.buf(n, hh, ww, 3 * c, name="qkv")
        low.conv(qkv, hh, ww, 3 * c, aux=_src(t, c, pro=L.PRO_GN, gn=gn), w_aux=wqkv, bias=bqkv)
        ao = b.buf(n, hh, ww, c, name="attn_o")
        L_ = hh * ww
        b.add(L.OP_ATTN, dict(qkv=qkv, dst=ao, n=n, l=L_, c=c, scale=float(int(c) ** (-0.5))), FC_ATTN,
              4.0 * n * L_ * L_ * c)
        out = b.buf(n, hh, ww, c, name="attn_out")
        low.conv(out, hh, ww, c, aux=_src(ao, c), w_aux=self.weights.matrix([(m.NIN_3.W, True)]),
                 bias=self.weights.vector([m.NIN_3.b]), resid=t,
                 scale=INV_SQRT2 if m.skip_rescale else 1.0, stats=True)
        return out

    # ------------------------------------------------------------------ execution
    def load_inputs(self, x, cond):
        self.x_in.tensor[: x.numel()].copy_(x.reshape(-1))
        self.cond.tensor[: self.n].copy_(cond.reshape(-1).to(torch.float32))
        if self.sig is not self.cond and self.sig.tensor is not None:
            self.sig.tensor[: self.n].copy_(self.model.sigmas.to(torch.float32)[cond.long()])

    def output_view(self):
        return self.out.tensor[: self.n * self.channels * self.h * self.w].view(self.n, self.channels, self.h, self.w)

    def forward(self, x, cond):
        if tuple(x.shape) != (self.n, self.channels, self.h, self.w):
            raise ValueError("engine built for %s, got %s" % ((self.n, self.channels, self.h, self.w), tuple(x.shape)))
        self.weights.refresh()
        self.load_inputs(x.contiguous(), cond)
        self.program.run()
        return self.output_view().clone()

    def flops_per_forward(self):
        return sum(self.program.flops)

    def validate_plans(self):
        """Host-side check of every conv launch plan (tile choice, halo fit, LDS size); returns LDS bytes per conv."""
        lib = L.load()
        out = []
        for i in range(self.program.n):
            op = self.program.ops[i]
            if op.kind == L.OP_CONV:
                r = lib.ssde_conv_lds_bytes(C.byref(op.u.conv))
                if r < 0:
                    raise L.SsdeError("op %d: %s" % (i, lib.ssde_last_error().decode()))
                out.append(r)
        return out
