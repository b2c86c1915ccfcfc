#!/bin/bash
# Local helper: (re)build libssde_hip.so, check that it loads against the ctypes mirror, then hand a script to gpurun.
# usage: tools/gpu_call.sh <timeout_s> <script on the GPU box>   (round-3 scripts: tools/gpu_calls/)
set -e
cd /root/repo
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"
