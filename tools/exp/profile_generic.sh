cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r06; O=gpurun_out/$R; mkdir -p $O
cat > /tmp/run_other.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
which = sys.argv[1]
print(bench.costdcnet_workload(4) if which == 'costdcnet' else bench.nlspn_workload(2, 3, dtype='mixed' if which == 'nlspn_mixed' else 'fp32', with_mixed=False))
PY
for W in costdcnet nlspn nlspn_mixed; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/other_$W -o x -- python3 /tmp/run_other.py $W > $O/other_$W.log 2>&1
  python3 tools/prof_top.py $O/other_$W 30 > $O/${R}_${W}_top_kernels.txt 2>&1
  rm -rf $O/other_$W
done
ls $O
