#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
(for W in 576 768; do echo "== F(4x4,3x3) SSDE_WGRAD4_WGS=$W"; SSDE_WGRAD_WINOGRAD=44 SSDE_WGRAD4_WGS=$W timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2"; done
echo "== F(2x2,3x3)"; SSDE_WGRAD_WINOGRAD=2 timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2") | tee $OUT/r3k_wgrad_bench.txt
cd /tmp && export TMPDIR=/tmp
SSDE_WGRAD_WINOGRAD=44 rocprofv3 --kernel-trace -d $OUT/prof_wg4b -o wg --output-format csv -- python $ROOT/tools/wgrad_bench.py 128 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/prof_wg4b/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("wgrad4","wino4_xform"))]
# group consecutive launches of one conv_wgrad call (xform_v, xform_z, gemm, sum, reduce)
names=lambda n: "xv" if "xform_v" in n else "xz" if "xform_z" in n else "gemm" if "gemm" in n else "sum" if "sum_splits" in n else "red"
calls=[]; cur={}
for r in rows:
    k=names(r["Kernel_Name"]); d=(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3
    if k=="xv" and cur: calls.append(cur); cur={}
    cur[k]=d; cur.setdefault("grid",{})[k]=r["Grid_Size_X"]
calls.append(cur)
seen=set()
for c in calls:
    key=tuple(sorted(c["grid"].items()))
    if key in seen: continue
    seen.add(key)
    print({k:round(v,1) for k,v in c.items() if k!="grid"}, c["grid"].get("gemm"))
PY
