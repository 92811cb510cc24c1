#!/bin/bash
# A/B: the weight-gradient reduction as one launch at the end of the backward pass, or split in two (DGM_MLP_REDUCE_SPLIT=<layer>: the
# layers down to <layer> reduced right behind that layer's paired launch, while their 130-160 MB of partial tiles may still sit in the MALL)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -2
DGM_MLP_REDUCE_SPLIT=4 timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -2
for s in -1 4 -1 4 3 5; do
  DGM_MLP_REDUCE_SPLIT=$s timeout 300 python bench.py --steps 200 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=$s', round(d['value'],2), 'it/s', round(d['ms_per_step'],4))"
done
