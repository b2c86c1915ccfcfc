#!/bin/bash
# Runs ON THE GPU BOX: HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, --kernel-trace only) over the
# TRAINING step, which tools/profile_gpu.sh leaves out of its PMC passes.  -> gpurun_out/train_pmc_<tag>.json
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/train_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --train-steps 2 --train-warmup 1"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o bench --output-format csv -- $BENCH > $OUT/pmc_$C.log 2>&1
  echo "pmc $C rc=$?"
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for key in ("wgrad_wino_kernel", "wgrad_wino_reduce_kernel", "wgrad_kernel", "wgrad_reduce_kernel", "conv_wino_kernel", "gemm1x1_kernel",
                        "prologue_bwd_kernel", "gn_bwd_reduce_kernel", "attn_bwd_kv_kernel", "attn_bwd_q_kernel", "adam_kernel", "colsum_kernel",
                        "conv_wino4_kernel", "wino4_xform_v_kernel", "wino4_xform_z_kernel", "wgrad4_gemm_kernel", "wgrad4_sum_splits_kernel",
                        "wgrad1x1_gemm_kernel", "gn_bwd_finalize_kernel", "conv_mfma_kernel", "conv_wino4g_kernel", "wino4_xform_vq_kernel", "gn_bwd_fused_kernel"):
                if key + "<" in k or key + "(" in k:
                    a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, cs in acc.items():
    fe = cs["FETCH_SIZE"][0] / max(cs["FETCH_SIZE"][1], 1); wr = cs["WRITE_SIZE"][0] / max(cs["WRITE_SIZE"][1], 1)
    out[k] = {"fetch_kib_avg": fe, "write_kib_avg": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "dispatches": cs["FETCH_SIZE"][1]}
json.dump(out, open("$ROOT/gpurun_out/train_pmc_$TAG.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"]):
    print("%-28s launches %4d  hbm/launch %8.1f MB" % (k, v["dispatches"], v["hbm_bytes_per_launch"] / 1e6))
PY
find $OUT -type f -size +8M -delete
