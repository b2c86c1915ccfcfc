cd $GRAFT_REPO_ROOT
O=gpurun_out/persist_ab.txt
echo "== persist 0" > $O; SSDE_WINO_PERSIST=0 python tools/conv_bench.py 256 >> $O 2>&1
echo "== persist 1" >> $O; SSDE_WINO_PERSIST=1 python tools/conv_bench.py 256 >> $O 2>&1
echo "== tests persist 1" >> $O
SSDE_WINO_PERSIST=1 timeout 600 python -m pytest tests/test_bench_sizes_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -5 >> $O
echo "== bench persist 0" >> $O; SSDE_WINO_PERSIST=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-train --no-roofline 2>/dev/null | tail -1 >> $O
echo "== bench persist 1" >> $O; SSDE_WINO_PERSIST=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-train --no-roofline 2>/dev/null | tail -1 >> $O
cat $O
