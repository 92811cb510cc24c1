#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_densify.py tests/test_trainer_dp_gpu.py tests/test_optim.py -m gpu -q -x > gpurun_out/pytest_densify.log 2>&1; echo "pytest exit $?"; tail -25 gpurun_out/pytest_densify.log
