#!/usr/bin/env python
"""Summarise the rocprofv3 outputs of tools/profile_gpu.sh: per-kernel time (kernel-trace --stats) and per-kernel PMC
averages (separate passes).  HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB
and, on gfx950, FETCH_SIZE counts exactly half of the bytes of wide coalesced reads -> traffic = (2*FETCH + WRITE)*1024."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("conv_wino4r_kernel", "wino4_xform_vq_kernel", "conv_small_cout_kernel", "conv_wino4x_kernel", "gn_bwd_fused_kernel", "gemm1x1_bf16x6_kernel", "gemm1x1_pipe_kernel",
                "wgrad4_gemm_kernel", "wino4_xform_z_kernel", "wino4_xform_v_kernel", "wgrad1x1_gemm_kernel", "gn_part_finalize_kernel", "gn_bwd_finalize_kernel",
                "conv_wino4_kernel", "conv_wino_kernel", "conv_mfma_kernel", "gemm1x1_kernel", "wgrad_wino_kernel", "wgrad_kernel", "wgrad_reduce_kernel", "attn_kernel", "attn_bwd_q_kernel",
                "attn_bwd_kv_kernel", "gn_stats_kernel", "gn_bwd_reduce_kernel", "prologue_bwd_kernel", "colsum_kernel",
                "upfirdn_kernel", "adam_kernel", "pack_wino3_kernel", "pack_conv3_kernel", "randn_kernel", "langevin_kernel",
                "predictor_kernel", "sumsq_kernel"):
        if key in name:
            if key == "conv_mfma_kernel":
                return key + name[name.find("<"):name.find(">") + 1]
            return key
    return name[:60]


def find(dirname, pattern):
    fs = glob.glob(os.path.join(dirname, "**", pattern), recursive=True)
    return fs[0] if fs else None


def kernel_stats(out):
    f = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    rows = []
    if f:
        for r in csv.DictReader(open(f)):
            rows.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "total_us": float(r["TotalDurationNs"]) / 1e3,
                         "avg_us": float(r["AverageNs"]) / 1e3, "percent": float(r["Percentage"])})
    return rows


def pmc(out, counter_dir):
    f = find(os.path.join(out, counter_dir), "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    if f:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: {c: {"sum": v[0], "dispatches": v[1], "avg": v[0] / max(v[1], 1)} for c, v in cs.items()} for k, cs in acc.items()}


def csrc_sha():
    """sha256 over the kernel sources the counters were collected on: bench.py compares it with the sources it runs and
    marks the PMC figures `pmc_stale` when a kernel changed after the profile was taken"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "score_sde_pytorch_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def main(out, dst):
    res = {"csrc_sha256": csrc_sha(), "kernel_stats": kernel_stats(out)}
    fetch, write, mfma = pmc(out, "pmc_FETCH_SIZE"), pmc(out, "pmc_WRITE_SIZE"), pmc(out, "pmc_MFMA")
    traffic = {}
    for k in set(fetch) | set(write):
        fe = fetch.get(k, {}).get("FETCH_SIZE", {}).get("avg", 0.0)
        wr = write.get(k, {}).get("WRITE_SIZE", {}).get("avg", 0.0)
        traffic[k] = {"fetch_kib_avg": fe, "write_kib_avg": wr, "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0,
                      "dispatches": fetch.get(k, {}).get("FETCH_SIZE", {}).get("dispatches", 0)}
    res["hbm_traffic"] = traffic
    res["mfma"] = {k: {c: v["avg"] for c, v in cs.items()} for k, cs in mfma.items()}
    json.dump(res, open(dst, "w"), indent=1)
    top = sorted(res["kernel_stats"], key=lambda r: -r["total_us"])[:12]
    for r in top:
        t = traffic.get(r["kernel"], {})
        print("%-46s calls %6d avg %9.1f us  %5.1f%%  hbm/launch %8.1f MB" % (r["kernel"], r["calls"], r["avg_us"], r["percent"],
                                                                             t.get("hbm_bytes_per_launch", 0) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
