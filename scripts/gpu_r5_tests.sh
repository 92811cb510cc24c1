#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS} 2>&1 | tail -12
