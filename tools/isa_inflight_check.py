#!/usr/bin/env python
"""Walks the emitted ISA of attn_x6_kernel (attention.hip) and checks that no instruction touches a VGPR whose asm-issued load is
still in flight.  The streamed K / V loads of that kernel are inline-asm statements hipcc does not track; the kernel counts them
itself (SSDE_X6_WAIT).  hipcc is free to COPY a value between two statements -- and a copy of a register whose load has not
landed copies stale data.  The producers' code is straight-line, so the vmcnt queue can be replayed: every global load enters the
queue, every `s_waitcnt vmcnt(n)` retires all but the n youngest, and any other instruction naming a register of a queued asm load
is a violation.  Blocks that hold MFMAs (the consumers' arm: those waves never issue the loads) are skipped.

usage: python tools/isa_inflight_check.py [attention.s]   (without an argument: compiles csrc/attention.hip to a temporary .s)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(txt):
    r = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', txt):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'(?<![\w\[:])v(\d+)\b', txt):
        r.add(int(m.group(1)))
    return r


def check_kernel(name, lines):
    # basic blocks by label
    blocks, cur = [], []
    for ln in lines:
        if re.match(r'^\.LBB\d+_\d+:', ln):
            blocks.append(cur)
            cur = []
        cur.append(ln)
    blocks.append(cur)
    queue, violations, n_asm, n_wait = [], [], 0, 0          # queue: (is_asm, regs)
    for blk in blocks:
        if any('v_mfma' in ln for ln in blk):
            continue
        prev = ''
        for ln in blk:
            t = ln.strip()
            if not t or t.startswith(';') or t.startswith('.'):
                prev = t if t.startswith(';;#ASM') else prev
                continue
            if 'global_load' in t or 'buffer_load' in t:
                is_asm = 's_nop 4' in prev
                m = re.search(r'v\[\d+:\d+\]|(?<![\w\[:])v\d+\b', t)
                queue.append((is_asm, regs_of(m.group(0)) if m else set()))
                n_asm += is_asm
            elif t.startswith('s_waitcnt') and 'vmcnt' in t:
                n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
                n_wait += 1
                while len(queue) > n:
                    queue.pop(0)
            elif 'global_store' in t or 'buffer_store' in t:
                queue.append((False, set()))                 # (stores count in vmcnt on gfx9)
            else:
                busy = set().union(*[r for a, r in queue if a]) if queue else set()
                if busy & regs_of(t) and not t.startswith('s_'):
                    violations.append(t)
            prev = t
    return n_asm, n_wait, violations


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "attention.s")
            src = os.path.join(ROOT, "score_sde_pytorch_amd", "csrc", "attention.hip")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-munsafe-fp-atomics",
                            "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True, cwd=os.path.dirname(src))
            text = open(out).read()
    bad = 0
    for m in re.finditer(r'^(_ZN\S*attn_x6_kernel[^:\s]*):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M):
        n_asm, n_wait, v = check_kernel(m.group(1), m.group(2).split('\n'))
        print("%s: %d asm loads, %d vmcnt waits, %d instructions touching a register in flight" % (m.group(1), n_asm, n_wait, len(v)))
        for t in v[:8]:
            print("    ", t)
        bad += len(v)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
