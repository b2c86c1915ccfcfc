"""Serialise a lowered program as a position-independent PLAN BLOB for hosts without Python (include/ssde.h, "plans";
loader and runner: csrc/plan.hip -> ssde_plan_load, ssde_unet_forward, ssde_pc_reset / ssde_pc_run / ssde_pc_state).

The lowering of NCSNpp.forward (reference models/ncsnpp.py:232-381) and of the predictor-corrector loop body
(sampling.py:403-407) exists once, in engine.py / pc_engine.py.  This module walks what that lowering produced:

  * every tensor the program can point into (activation arena, static I/O, packed kernel-layout weights, the
    reference-layout parameters, embedding / step tables, sampler state, the device-side re-pack descriptor tables)
    becomes a REGION (one per torch storage; constant regions carry their bytes);
  * every pointer field of every `ssde_op` -- found through the ctypes mirror of include/ssde.h -- is zeroed and
    recorded as a relocation (region, byte offset); the same for the src / src2 / dst pointers inside the re-pack tables;
  * parameters are listed with their state_dict names so that a host can overwrite them from a checkpoint and call
    ssde_plan_refresh_weights (four device launches re-pack every kernel-layout copy).

Export needs no GPU: a program lowered on 'cpu' (dry lowering) serialises the same way.
"""
import bisect
import ctypes as C
import struct

import torch

from . import _lib as L

MAGIC = b"SSDEPLN1"
REGION_ZERO, REGION_CONST = 0, 1
RELOC_OP, RELOC_REFRESH_OP, RELOC_REGION = 0, 1, 2
PLAN_UNET, PLAN_PC, PLAN_TRAIN = 0, 1, 2
IO_X, IO_COND, IO_SIGMA, IO_STD, IO_OUT, IO_XMEAN, IO_STEP, IO_SEED = range(8)
(IO_BATCH, IO_Z, IO_A, IO_S, IO_G2, IO_LOSS, IO_HYPER, IO_DROP_SEED, IO_GOUT, IO_GX, IO_GRAD, IO_PARAMS) = range(8, 20)
IO_SLOTS = 24


class PlanHeader(C.Structure):
    _fields_ = [("magic", C.c_char * 8), ("abi_version", C.c_int32), ("sizeof_op", C.c_int32),
                ("n_regions", C.c_int32), ("n_ops", C.c_int32), ("n_refresh_ops", C.c_int32), ("n_relocs", C.c_int32),
                ("n_params", C.c_int32), ("kind", C.c_int32),
                ("batch", C.c_int32), ("channels", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("nfe_per_iteration", C.c_int32), ("sde_steps", C.c_int32), ("io", C.c_int32 * IO_SLOTS), ("seg", C.c_int32 * 4),
                ("n_flat", C.c_int64), ("data_bytes", C.c_int64)]


class PlanRegion(C.Structure):
    _fields_ = [("bytes", C.c_int64), ("data_offset", C.c_int64), ("kind", C.c_int32), ("_pad0", C.c_int32), ("name", C.c_char * 32)]


class PlanReloc(C.Structure):
    _fields_ = [("target_kind", C.c_int32), ("target", C.c_int32), ("byte_offset", C.c_int64), ("region", C.c_int32),
                ("_pad0", C.c_int32), ("offset", C.c_int64)]


class PlanParam(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("region", C.c_int32), ("_pad0", C.c_int32), ("offset", C.c_int64), ("numel", C.c_int64)]


_pointer_fields = L.pointer_offsets      # (byte offset, ) of every pointer field: nested structures and arrays of them included


_OP_PTRS = {}


def _op_pointer_offsets(kind):
    if kind not in _OP_PTRS:
        member = L._UNION_FIELD[kind]
        cls = dict(L._OpUnion._fields_)[member]
        _OP_PTRS[kind] = _pointer_fields(cls, L.Op.u.offset)
    return _OP_PTRS[kind]


class _Regions:
    """Torch storages the program points into; lookup of an address -> (region id, byte offset)."""

    def __init__(self):
        self.by_ptr, self.list = {}, []      # storage data_ptr -> id ; [dict(ptr, bytes, kind, name, storage_tensor)]

    def add(self, t, kind, name):
        if t is None:
            return None
        st = t.untyped_storage()
        key = st.data_ptr()
        if key in self.by_ptr:
            r = self.list[self.by_ptr[key]]
            r["kind"] = max(r["kind"], kind)
            return self.by_ptr[key]
        self.by_ptr[key] = len(self.list)
        self.list.append(dict(ptr=key, bytes=st.nbytes(), kind=kind, name=name, keep=t))
        return len(self.list) - 1

    def freeze(self):
        self._starts = sorted((r["ptr"], i) for i, r in enumerate(self.list))
        self._keys = [s for s, _ in self._starts]

    def find(self, addr):
        i = bisect.bisect_right(self._keys, addr) - 1
        if i >= 0:
            rid = self._starts[i][1]
            r = self.list[rid]
            if r["ptr"] <= addr <= r["ptr"] + r["bytes"]:
                return rid, addr - r["ptr"]
        raise ValueError("plan export: pointer 0x%x lies in no known tensor" % addr)

    def data(self, rid):
        r = self.list[rid]
        t = r["keep"]
        flat = torch.empty(0, dtype=torch.uint8).set_(t.untyped_storage().cpu() if t.is_cuda else t.untyped_storage())
        return bytes(flat.numpy().tobytes())


def _collect_unet(regions, eng):
    for i, blk in enumerate(eng.b.blocks):
        regions.add(blk, REGION_ZERO, "arena%d" % i)
    for i, t in enumerate(eng.b.keep):
        regions.add(t, REGION_CONST, "table%d" % i)
    for i, e in enumerate(eng.weights.entries):
        regions.add(e[0], REGION_CONST, "packed%d" % i)
        for s in e[1]:
            regions.add(s.detach(), REGION_CONST, "param")
    if getattr(eng.weights, "_tables", None) is None:
        eng.weights._build_tables()
    for i, (_, raw) in enumerate(eng.weights._tables[0]):
        regions.add(raw, REGION_CONST, "packtab%d" % i)


def _emit(kind, regions, ops, n_ops, eng_unet, model, io_tensors, batch, shape, nfe, sde_steps, seg=(0, 0, 0, 0), n_flat=0):
    regions.freeze()
    relocs = []
    op_bytes = bytearray()
    for i in range(n_ops):
        raw = bytearray(bytes(ops[i]))
        for off in _op_pointer_offsets(int(ops[i].kind)):
            (addr,) = struct.unpack_from("<Q", raw, off)
            if addr:
                rid, roff = regions.find(addr)
                relocs.append((RELOC_OP, i, off, rid, roff))
                struct.pack_into("<Q", raw, off, 0)
        op_bytes += raw
    # ---- weight refresh program: one ssde_pack_weights launch per descriptor table
    refresh = bytearray()
    tables = eng_unet.weights._tables[0]
    desc_ptrs = _pointer_fields(L.PackDesc)
    table_patches = {}
    for j, (args, raw_t) in enumerate(tables):
        op = L.make_op(L.OP_PACK, args)
        rawop = bytearray(bytes(op))
        for off in _op_pointer_offsets(L.OP_PACK):
            (addr,) = struct.unpack_from("<Q", rawop, off)
            if addr:
                rid, roff = regions.find(addr)
                relocs.append((RELOC_REFRESH_OP, j, off, rid, roff))
                struct.pack_into("<Q", rawop, off, 0)
        refresh += rawop
        # pointers inside the table
        tid, toff0 = regions.find(raw_t.data_ptr())
        tb = bytearray(regions.data(tid))
        for d in range(int(args.count)):
            for off in desc_ptrs:
                pos = toff0 + d * C.sizeof(L.PackDesc) + off
                (addr,) = struct.unpack_from("<Q", tb, pos)
                if addr:
                    rid, roff = regions.find(addr)
                    relocs.append((RELOC_REGION, tid, pos, rid, roff))
                    struct.pack_into("<Q", tb, pos, 0)
        table_patches[tid] = bytes(tb)
    # ---- parameters by state_dict name
    params = []
    for name, p in model.named_parameters():
        try:
            rid, roff = regions.find(p.data_ptr())
        except ValueError:
            continue                                    # a parameter no kernel reads (none today)
        params.append((name, rid, roff, p.numel()))
    # ---- data section
    data = bytearray()
    reg_structs = []
    for rid, r in enumerate(regions.list):
        rs = PlanRegion()
        rs.bytes, rs.kind, rs.name = r["bytes"], r["kind"], r["name"].encode()[:31]
        rs.data_offset = -1
        if r["kind"] == REGION_CONST:
            while len(data) % 16:
                data.append(0)
            rs.data_offset = len(data)
            data += table_patches.get(rid) or regions.data(rid)
        reg_structs.append(rs)
    hdr = PlanHeader()
    hdr.magic, hdr.abi_version, hdr.sizeof_op = MAGIC, L.ABI_VERSION, C.sizeof(L.Op)
    hdr.n_regions, hdr.n_ops, hdr.n_refresh_ops, hdr.n_relocs, hdr.n_params = len(reg_structs), n_ops, len(tables), len(relocs), len(params)
    hdr.kind, hdr.batch = kind, batch
    hdr.channels, hdr.height, hdr.width = shape
    hdr.nfe_per_iteration, hdr.sde_steps = nfe, sde_steps
    for k in range(4):
        hdr.seg[k] = int(seg[k])
    hdr.n_flat = int(n_flat)
    for slot in range(IO_SLOTS):
        t = io_tensors.get(slot)
        hdr.io[slot] = -1
        if t is not None:
            rid, roff = regions.find(t.data_ptr())
            assert roff == 0, "plan I/O tensors must start their storage"
            hdr.io[slot] = rid
    hdr.data_bytes = len(data)
    out = bytearray(bytes(hdr))
    for rs in reg_structs:
        out += bytes(rs)
    out += op_bytes + refresh
    for tk, tgt, boff, rid, roff in relocs:
        q = PlanReloc()
        q.target_kind, q.target, q.byte_offset, q.region, q.offset = tk, tgt, boff, rid, roff
        out += bytes(q)
    for name, rid, roff, numel in params:
        q = PlanParam()
        q.name, q.region, q.offset, q.numel = name.encode()[:95], rid, roff, numel
        out += bytes(q)
    out += data
    return bytes(out)


def export_unet_plan(eng):
    """engine.UNetEngine -> blob for ssde_plan_load / ssde_unet_forward."""
    eng.weights.refresh()
    regions = _Regions()
    _collect_unet(regions, eng)
    io = {IO_X: eng.x_in.tensor, IO_COND: eng.cond.tensor, IO_OUT: eng.out.tensor}
    if eng.sig is not eng.cond:
        io[IO_SIGMA] = eng.sig.tensor
    if eng.std is not None:
        io[IO_STD] = eng.std.tensor
    return _emit(PLAN_UNET, regions, eng.program.ops, eng.program.n, eng, eng.model, io, eng.n, (eng.channels, eng.h, eng.w), 1, 0)


def export_pc_plan(sampler, with_rng=True):
    """pc_engine.FusedPCSampler -> blob for ssde_pc_reset / ssde_pc_run / ssde_pc_state (one program = one PC iteration)."""
    eng = sampler.unet
    eng.weights.refresh()
    prog = sampler.step_program(with_rng=with_rng)
    regions = _Regions()
    _collect_unet(regions, eng)
    for name in ("x_mean", "z_c", "z_p", "gss", "zss", "step", "seed_word", "proj_data", "proj_mask", "z_pc", "z_pp"):
        t = getattr(sampler, name, None)
        if t is not None:
            regions.add(t, REGION_ZERO, name)
    for k, t in sampler.tabs.items():
        regions.add(t, REGION_CONST, "tab_" + k)
    io = {IO_X: eng.x_in.tensor, IO_COND: eng.cond.tensor, IO_OUT: eng.out.tensor, IO_XMEAN: sampler.x_mean,
          IO_STEP: sampler.step, IO_SEED: sampler.seed_word}
    if eng.std is not None:
        io[IO_STD] = eng.std.tensor
    B, Cc, H, W = sampler.shape
    return _emit(PLAN_PC, regions, prog.ops, prog.n, eng, sampler.model, io, B, (Cc, H, W), sampler.nfe_per_step(), sampler.sde.N)


def export_train_plan(fs, optimizer=None, ema=None):
    """losses.FusedTrainStep -> blob for ssde_train_step / ssde_train_forward / ssde_unet_backward.

    One op array: [perturb | forward | loss head | backward | clip + Adam + EMA]; the segment starts go into the
    header.  The flat parameter buffer (the parameters are views into it), the flat gradient, Adam's moments and the EMA
    shadow are regions of their own; moments and shadow carry their CURRENT contents, so a plan exported mid-run resumes.
    Without `optimizer` the plan stops after the backward segment (hosts that bring their own optimizer read the flat
    gradient through ssde_unet_backward)."""
    eng = fs.eng
    eng.weights.refresh()
    regions = _Regions()
    _collect_unet(regions, eng)
    regions.add(fs.flat.data, REGION_CONST, "flat_params")
    regions.add(fs.flat.grad, REGION_ZERO, "flat_grad")
    for name in ("z", "batch", "a", "s", "g2", "losses", "loss", "hyper", "gnorm", "partial"):
        regions.add(getattr(fs, name), REGION_ZERO, name)
    head_ops = lambda prog: [prog.ops[i] for i in range(prog.n)]                 # noqa: E731
    ops = head_ops(fs._head[0])
    seg0 = len(ops)
    ops += [eng.program.ops[i] for i in range(eng.n_fwd)]
    seg1 = len(ops)
    ops += head_ops(fs._head[1])
    seg2 = len(ops)
    ops += [eng.program.ops[i] for i in range(eng.n_fwd, eng.program.n)]
    seg3 = 0
    if optimizer is not None:
        m, v = optimizer.flatten_like(fs.flat)
        regions.add(m, REGION_CONST, "adam_m")
        regions.add(v, REGION_CONST, "adam_v")
        if ema is not None:
            regions.add(ema.flatten_like(fs.flat), REGION_CONST, "ema")
        seg3 = len(ops)
        ops += head_ops(fs._optimizer_program(optimizer, ema))
    io = {IO_X: eng.x_in.tensor, IO_COND: eng.cond.tensor, IO_OUT: eng.out.tensor, IO_BATCH: fs.batch, IO_Z: fs.z, IO_A: fs.a,
          IO_S: fs.s, IO_LOSS: fs.loss, IO_HYPER: fs.hyper, IO_GOUT: eng.gout.tensor, IO_GRAD: fs.flat.grad, IO_PARAMS: fs.flat.data}
    if fs.spec["likelihood_weighting"]:
        io[IO_G2] = fs.g2
    if eng.sig is not eng.cond:
        io[IO_SIGMA] = eng.sig.tensor
    if eng.std is not None:
        io[IO_STD] = eng.std.tensor
    if eng.drop_seed is not None:
        io[IO_DROP_SEED] = eng.drop_seed
    if eng.gx is not None:
        io[IO_GX] = eng.gx.tensor
    arr = L.op_array(ops)
    return _emit(PLAN_TRAIN, regions, arr, len(ops), eng, fs.model, io, eng.n, (eng.channels, eng.h, eng.w), 1, 0,
                 seg=(seg0, seg1, seg2, seg3), n_flat=fs.flat.numel)


# ---------------------------------------------------------------------------------------------------------------------
# ctypes binding of the plan entry points (what a C host calls; used by the tests and by tools)
def bind(lib):
    lib.ssde_plan_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.ssde_plan_load_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.ssde_plan_destroy.argtypes = [C.c_void_p]
    lib.ssde_plan_info.argtypes = [C.c_void_p, C.POINTER(PlanHeader)]
    lib.ssde_plan_param.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]
    lib.ssde_plan_refresh_weights.argtypes = [C.c_void_p, C.c_void_p]
    lib.ssde_unet_forward.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_void_p]
    lib.ssde_pc_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.ssde_pc_run.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.ssde_pc_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ssde_train_step.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.POINTER(C.c_float), C.c_uint32, C.c_void_p, C.c_void_p]
    lib.ssde_train_forward.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p, C.c_void_p]
    lib.ssde_unet_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ssde_plan_copy_io.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    return lib


class LoadedPlan:
    """A plan loaded through the C API (device memory owned by the library)."""

    def __init__(self, blob):
        self.lib = bind(L.load())
        self.handle = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        L.check(self.lib.ssde_plan_load(C.cast(buf, C.c_void_p), len(blob), C.byref(self.handle)), "ssde_plan_load")
        self.header = PlanHeader()
        L.check(self.lib.ssde_plan_info(self.handle, C.byref(self.header)))

    def param(self, name):
        dev, numel = C.c_void_p(), C.c_int64()
        L.check(self.lib.ssde_plan_param(self.handle, name.encode(), -1, C.byref(dev), C.byref(numel), None), "ssde_plan_param")
        return dev.value, numel.value

    def refresh_weights(self, stream=None):
        L.check(self.lib.ssde_plan_refresh_weights(self.handle, C.c_void_p(stream or 0)), "ssde_plan_refresh_weights")

    def unet_forward(self, x, cond, sigma=None, std=None, stream=None):
        out = torch.empty_like(x)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        L.check(self.lib.ssde_unet_forward(self.handle, p(x), p(cond), p(sigma), p(std), p(out), C.c_void_p(stream or 0)), "ssde_unet_forward")
        return out

    def train_step(self, batch, z, a, s, labels, hyper9, seed, g2=None, stream=None):
        """one optimisation step through the C entry point; returns the loss as a 1-element tensor on the inputs' device"""
        loss = torch.zeros(1, dtype=torch.float32, device=batch.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        hy = (C.c_float * 9)(*[float(v) for v in hyper9])
        L.check(self.lib.ssde_train_step(self.handle, p(batch), p(z), p(a), p(s), p(labels), p(g2), hy, int(seed) & 0xFFFFFFFF, p(loss),
                                         C.c_void_p(stream or 0)), "ssde_train_step")
        return loss

    def train_forward(self, x, cond, seed=0, sigma=None, std=None, stream=None):
        out = torch.empty_like(x)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        L.check(self.lib.ssde_train_forward(self.handle, p(x), p(cond), p(sigma), p(std), int(seed) & 0xFFFFFFFF, p(out),
                                            C.c_void_p(stream or 0)), "ssde_train_forward")
        return out

    def unet_backward(self, dout, want_dx=False, stream=None):
        dx = torch.empty_like(dout) if want_dx else None
        dparams = torch.empty(int(self.header.n_flat), dtype=torch.float32, device=dout.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        L.check(self.lib.ssde_unet_backward(self.handle, p(dout), p(dx), p(dparams), C.c_void_p(stream or 0)), "ssde_unet_backward")
        return dx, dparams

    def read_io(self, slot, like):
        """the first like.numel() elements of an I/O region, as a tensor shaped / typed / placed like `like`"""
        out = torch.empty_like(like)
        L.check(self.lib.ssde_plan_copy_io(self.handle, int(slot), C.c_void_p(out.data_ptr()), out.numel() * out.element_size(), 0,
                                           C.c_void_p(0)), "ssde_plan_copy_io")
        return out

    def close(self):
        if self.handle:
            self.lib.ssde_plan_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
