#!/bin/bash
# per-kernel durations of the tree's build (A) against proxytta/libptta_hip.alt.so (B) under rocprofv3 --kernel-trace:
#   bash tools/exp_kernel_ab.sh 'regex of kernel names'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=tta-depth-completion_amd/proxytta
cp $L/libptta_hip.so /tmp/libA.so; cp $L/libptta_hip.alt.so /tmp/libB.so
for V in A B; do
  cp /tmp/lib$V.so $L/libptta_hip.so
  rm -rf gpurun_out/kab
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kab -o x -- python3 bench.py --steps 20 --warmup 10 --single-block --no-nlspn --no-cpu-baseline --no-self-check > /dev/null 2> gpurun_out/kab.log
  echo "== lib $V"; python3 - gpurun_out/kab/x_kernel_stats.csv "$1" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        print('  %-70s calls %5s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cp /tmp/libA.so $L/libptta_hip.so; rm -rf gpurun_out/kab
