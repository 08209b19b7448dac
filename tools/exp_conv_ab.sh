#!/bin/bash
# solo duration of the stride-1 32->32 convolution at 352x1216 (tools/bench_conv32.py) for the tree's build (A) and proxytta/libptta_hip.alt.so (B)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=tta-depth-completion_amd/proxytta
cp $L/libptta_hip.so /tmp/libA.so; cp $L/libptta_hip.alt.so /tmp/libB.so
for V in A B A B; do
  cp /tmp/lib$V.so $L/libptta_hip.so
  for DT in fp32 narrow; do
    rm -rf gpurun_out/cab
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cab -o x -- python3 tools/bench_conv32.py 1 $DT > /dev/null 2> gpurun_out/cab.log
    python3 - gpurun_out/cab/x_kernel_stats.csv "lib $V $DT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv32_s1_x3' in r['Name']: print(sys.argv[2], r['Name'][:60], 'calls', r['Calls'], 'avg us %.2f' % (float(r['AverageNs']) / 1e3), 'min %.2f' % (float(r['MinNs']) / 1e3))
PY
  done
done
cp /tmp/libA.so $L/libptta_hip.so; rm -rf gpurun_out/cab
