"""Compile-time facts the hand-written kernels rely on, checked in the gfx950 ISA hipcc emits (no GPU needed):
  * the LDS-DMA macro writes M0 from inline asm without a clobber (hipcc rejects "m0" as a clobber): nothing else in the
    Winograd kernels may read or write M0;
  * the main loops of the Winograd kernels must not touch scratch (a spill inside the matrix stream is a memory round trip
    per stage); the epilogues may keep a handful of spills."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _isa(src, tmp_path, defines=()):
    out = os.path.join(str(tmp_path), src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "score_sde_pytorch_amd", "csrc"), *defines,
           os.path.join(ROOT, "score_sde_pytorch_amd", "csrc", src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _kernels(isa, name):
    """(symbol, body) of every instantiation of a kernel"""
    found = []
    for m in re.finditer(r"^(_ZN\S*%s\S*):" % name, isa, re.M):
        body = isa[m.start():]
        found.append((m.group(1), body[:body.index(".Lfunc_end")]))
    assert found, name
    return found


def _main_loop(body):
    """the innermost backward-branch region that holds the kernel's MFMAs"""
    lines = body.split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    total = sum("v_mfma" in l for l in lines)
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i) < i:
            lo = labels[m.group(1)]
            n = sum("v_mfma" in x for x in lines[lo:i])
            if n >= 8 and (best is None or i - lo < best[1] - best[0]):
                best = (lo, i)
    assert best is not None
    return lines[best[0]:best[1] + 1]


@pytest.mark.parametrize("src,kernel", [("conv_wino4.hip", "conv_wino4_kernel"), ("conv_wino.hip", "conv_wino_kernel"),
                                        ("conv_wino4r.hip", "conv_wino4r_kernel")])
def test_winograd_kernels_own_m0_and_keep_scratch_out_of_the_loop(src, kernel, tmp_path):
    isa = _isa(src, tmp_path)
    for sym, body in _kernels(isa, kernel):
        m0_users = [l.strip() for l in body.split("\n") if re.search(r"\bm0\b", l.split(";")[0]) and "s_mov_b32 m0" not in l]
        assert not m0_users, (sym, m0_users[:4])
        loop = _main_loop(body)
        assert sum("v_mfma" in l for l in loop) >= 8
        scratch = [l.strip() for l in loop if re.match(r"\s+scratch_", l)]
        assert not scratch, (sym, scratch[:4])
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", isa)]
    # (outside the loop: set-up and the two-round epilogue; the training-forward instantiations that also store the transformed
    #  input -- kEmitV -- carry two more live values through the epilogue)
    assert max(spills) <= 24, spills


def _regs(text):
    """VGPR numbers an operand string mentions: v7, v[4:7]"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def test_asm_halo_loads_are_not_read_before_their_counted_wait(tmp_path):
    """conv_wino4.hip issues its halo loads from inline asm (SSDE_GLOAD16) and counts them with its own s_waitcnt
    (SSDE_WAIT_VMCNT_FOR): hipcc does not know the destination registers are in flight, so a copy or a spill of them
    between the load and the wait would move stale data.  Walk every path from each such load to the first
    `s_waitcnt vmcnt(n)`, n <= 3 (the stage head / the fill waits), and require that no instruction on the way names
    the destination registers."""
    isa = _isa("conv_wino4.hip", tmp_path)
    for sym, body in _kernels(isa, "conv_wino4_kernel"):
        lines = [l.split(";")[0].rstrip() for l in body.split("\n")]
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
        loads = [i for i, l in enumerate(lines) if re.match(r"\s+global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[", l)]
        assert len(loads) >= 4, (sym, len(loads))
        for i in loads:
            dst = _regs(re.match(r"\s+global_load_dwordx4 (v\[\d+:\d+\])", lines[i]).group(1))
            seen, todo = set(), [i + 1]
            while todo:
                j = todo.pop()
                while j < len(lines) and j not in seen:
                    seen.add(j)
                    l = lines[j].strip()
                    m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
                    if m and int(m.group(1)) <= 3:
                        break
                    if l and not l.startswith(".") and not l.endswith(":"):
                        ops = l.split(None, 1)[1] if " " in l else ""
                        assert not (_regs(ops) & dst), (sym, lines[i].strip(), l)
                        b = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
                        if b:
                            todo.append(labels[b.group(1)])
                            if l.startswith("s_branch"):
                                break
                        assert not l.startswith("s_endpgm"), (sym, lines[i].strip(), "reached the end without a wait")
                    j += 1


def test_register_fed_winograd_matrix_loop_counts_its_own_loads(tmp_path):
    """conv_wino4r.hip: the steady-state loop body is TWO stages (register sets 0 and 1), each 18 MFMAs fed by 10 global loads per
    wave (five dwordx2 runs of V, four dwordx4 + one dwordx2 of the per-lane weight image) issued from inline asm into the very
    registers the MFMAs of stage + 2 read, with the hand-counted wait vmcnt(18) before every position (the comment above the stage
    body derives it); no LDS access, no barrier, no wait hipcc added.  hipcc does not know the destination registers are in flight:
    replay the loop against a model of the in-order VMEM queue and require that no instruction reads a register whose load has not
    been waited for (two trips: the second starts from the queue the first left behind, as every trip after the prologue does).
    The two peeled last stages issue nothing: 18 / 16 / 14 / 12 / 10, then 8 / 6 / 4 / 2 / 0."""
    isa = _isa("conv_wino4r.hip", tmp_path)
    found = _kernels(isa, "conv_wino4r_kernel")
    assert len(found) == 4                        # unsplit (one tile per workgroup / persistent), the reduction split over 2 / 4 workgroups per tile
    for sym, body in found:
        _check_register_fed_loop(sym, body)
        # (the split instantiations keep a few scalars in VGPR lanes: none of that inside the matrix loop)
        assert not any(re.match(r"\s+v_(readlane|writelane)", l) for l in _main_loop(body)), sym


def _check_register_fed_loop(sym, body):
    loop = _main_loop(body)
    code = [l.split(";")[0].strip() for l in loop]
    code = [l for l in code if l and not l.startswith(".") and not l.endswith(":")]
    assert sum("v_mfma_f32_32x32x2" in l for l in code) == 36
    assert sum(l.startswith("global_load_dwordx2") for l in code) == 12 and sum(l.startswith("global_load_dwordx4") for l in code) == 8
    assert not any(l.startswith(("ds_", "s_barrier", "scratch_", "buffer_")) or "global_load_lds" in l for l in code), sym
    waits = [int(m.group(1)) for l in code for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", l)] if m]
    assert waits == [18] * 10, waits
    assert not any(re.match(r"s_waitcnt\s+(?!vmcnt)", l) for l in code), [l for l in code if l.startswith("s_waitcnt")]

    def replay(instrs, queue, check):
        for l in instrs:
            m = re.match(r"global_load_dwordx[24] (v\[\d+:\d+\]), (v\d+), s\[", l)
            if m:
                if check and queue:
                    assert not (_regs(m.group(2)) & set().union(*queue)), (sym, l)
                queue.append(_regs(m.group(1)))
                continue
            w = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
            if w:
                del queue[:max(0, len(queue) - int(w.group(1)))]
                continue
            if check and queue and " " in l and not l.startswith(("s_cbranch", "s_branch")):
                busy = set().union(*queue)
                assert not (_regs(l.split(None, 1)[1]) & busy), (sym, l, sorted(_regs(l.split(None, 1)[1]) & busy))
        return queue
    q = replay(code, [], False)
    assert len(q) == 20, len(q)                   # two whole rounds of loads are in flight at the loop boundary
    q = replay(code, q, True)
    assert len(q) == 20
    # the two peeled last stages: the 36 MFMAs after the loop with no load between them
    lines = [l.split(";")[0].strip() for l in body.split("\n")]
    end = max(i for i, l in enumerate(lines) if l == loop[-1].split(";")[0].strip())
    tail, n = [], 0
    for l in lines[end + 1:]:
        if l and not l.startswith(".") and not l.endswith(":"):
            tail.append(l)
            n += "v_mfma_f32_32x32x2" in l
            if n == 36:
                break
    assert n == 36 and not any(l.startswith("global_load") for l in tail)
    assert [int(m.group(1)) for l in tail for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", l)] if m] == [18, 16, 14, 12, 10, 8, 6, 4, 2, 0]
    assert replay(tail, q, True) == []


def test_persistent_register_fed_kernel_leaves_its_prefetched_operands_alone(tmp_path):
    """conv_wino4r_kernel<1, true> (one workgroup per CU walking several tiles) issues the first two stages' loads of the NEXT tile
    from the epilogue of the current one, into the very registers the matrix loop reads, 20 asm loads hipcc knows nothing about.
    Between their issue and the loop's first counted wait (s_waitcnt vmcnt(18)) lie the second output round's store phase, the loop
    back-edge and the next tile's set-up: walk every path from each of those loads (and from the 20 loads that prime the first tile)
    to that wait and require that no instruction names a destination register -- no copy at the back-edge, no spill, no reuse as a
    temporary of the store phase."""
    isa = _isa("conv_wino4r.hip", tmp_path)
    body = [b for sym, b in _kernels(isa, "conv_wino4r_kernel") if "ILi1ELb1E" in sym]
    assert len(body) == 1
    lines = [l.split(";")[0].rstrip() for l in body[0].split("\n")]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    mfma = [i for i, l in enumerate(lines) if "v_mfma" in l]
    asm_load = re.compile(r"\s+global_load_dwordx[24] (v\[\d+:\d+\]), v\d+, s\[")
    loads = [i for i, l in enumerate(lines) if asm_load.match(l)]
    outside = [i for i in loads if i < mfma[0] or i > mfma[-1]]
    assert len(outside) == 40 and sum(i > mfma[-1] for i in outside) == 20, (len(loads), len(outside))
    for i in outside:
        dst = _regs(asm_load.match(lines[i]).group(1))
        seen, todo, reached = set(), [i + 1], False
        while todo:
            j = todo.pop()
            while j < len(lines) and j not in seen:
                seen.add(j)
                l = lines[j].strip()
                if re.match(r"s_waitcnt vmcnt\(18\)", l):
                    reached = True
                    break
                if l and not l.startswith(".") and not l.endswith(":"):
                    if asm_load.match(lines[j]):
                        # another load of the round: its own destination -- or the priming load of the same register under
                        # `if (!primed)`, which a prefetched tile never executes: a redefinition ends this path either way
                        if _regs(asm_load.match(lines[j]).group(1)) & dst:
                            reached = True
                            break
                    else:
                        ops = l.split(None, 1)[1] if " " in l else ""
                        assert not (_regs(ops) & dst), (lines[i].strip(), j, l)
                    b = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
                    if b:
                        todo.append(labels[b.group(1)])
                        if l.startswith("s_branch"):
                            break
                    if l.startswith("s_endpgm"):
                        break
                j += 1
        assert reached, lines[i].strip()


def test_attention_x6_no_instruction_touches_a_register_whose_asm_load_is_in_flight(tmp_path):
    """attn_x6_kernel's streamed K / V loads are inline asm the compiler does not track (the kernel counts them: SSDE_X6_WAIT).
    A compiler-inserted copy of a ring register between a request and its wait would copy stale data: replay the vmcnt queue over the
    producers' straight-line code (tools/isa_inflight_check.py) for every instantiation -- and make sure the check can fail."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_inflight_check", os.path.join(ROOT, "tools", "isa_inflight_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    isa = _isa("attention.hip", tmp_path, defines=("-fno-gpu-rdc", "-munsafe-fp-atomics", "-fPIC"))
    kernels = _kernels(isa, "attn_x6_kernel")
    assert len(kernels) == 4
    for sym, body in kernels:
        lines = body[:body.index("s_endpgm")].split("\n")
        n_asm, n_wait, bad = chk.check_kernel(sym, lines)
        assert n_asm >= 80 and n_wait >= 20 and not bad, (sym, n_asm, n_wait, bad[:4])
        # the same code with its first counted wait removed must be flagged (the parked element is read while in flight)
        ks = [i for i, l in enumerate(lines) if "s_waitcnt vmcnt(12)" in l and "ASMSTART" in lines[i - 1]]
        k = ks[len(ks) // 2]                               # (a wait in the steady state: the first one follows the Q loads' own wait)
        broken = lines[:k] + lines[k + 1:]
        assert chk.check_kernel(sym, broken)[2], sym
