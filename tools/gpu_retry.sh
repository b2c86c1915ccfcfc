#!/bin/bash
# usage: tools/gpu_retry.sh <timeout_s> <script> <logfile>: tools/gpu_call.sh, retried every 90 s while the pod has no free GPU slot (exit code 3)
cd /root/repo
for i in $(seq 1 40); do
  tools/gpu_call.sh "$1" "$2" > "$3" 2>&1
  rc=$?
  if grep -q "nothing was charged" "$3"; then sleep 90; continue; fi
  exit $rc
done
