#!/bin/bash
# where the host's ~1.6 ms per step go: cProfile over the cfg1 bench (host-bound), top entries by own time and by cumulative time
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--workload', 'cfg1', '--steps', '400', '--warmup', '20', '--no-extras', '--no-cpu-baseline']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
for key in ('tottime', 'cumulative'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45); print(s.getvalue()[:9000])
" > gpurun_out/r06_hostprof_cfg1.txt 2>&1
grep -n "ncalls" -A45 gpurun_out/r06_hostprof_cfg1.txt | head -60 | cut -c1-170
