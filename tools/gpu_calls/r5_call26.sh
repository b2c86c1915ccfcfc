#!/bin/bash
# round 5, call 26: the rocprofv3 passes of tools/profile_gpu.sh again on the final kernel sources (comment-only edits since call 20
# changed the source hash the PMC summary is tied to)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 bash tools/profile_gpu.sh r5final2 2>&1 | tail -16
