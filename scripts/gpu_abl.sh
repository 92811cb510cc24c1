cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for a in 0 1 2; do
  cd /tmp && DGM_MLP_ABL=$a rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$a -o x -- python $GRAFT_REPO_ROOT/tools/mlp_bench.py 100000 5 hip > /dev/null 2>&1
  echo "ABL=$a"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/abl$a -name "*kernel_stats.csv") 1 4 | grep "gemm_kernel<0" | cut -c1-120
done
