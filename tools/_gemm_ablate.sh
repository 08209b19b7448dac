cd /tmp && export TMPDIR=/tmp
T=$(date +%s)
for A in 31 32 95; do
  PTTA_GEMM_ABLATE=$A rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/x$T-$A -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "ablate=$A $(grep 'gemm_x3_wide_kernel<1, 0>' /tmp/x$T-$A/*/*_kernel_stats.csv | cut -d, -f1-4 | sed 's/.*void //' | tr '\n' ' ')"
done
