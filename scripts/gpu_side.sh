#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_trainer_dp_gpu.py tests/test_rccl.py -m gpu -q 2>&1 | tail -3
