#!/bin/bash
# round 5, call 27: the sampler / ODE / likelihood GPU tests after the change of ode.rhs_cache_get (host code only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python -m pytest tests/test_sampler_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
