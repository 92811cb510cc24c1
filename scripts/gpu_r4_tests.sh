#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8
