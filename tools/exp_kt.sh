#!/bin/bash
# kernel-level A/B of one ptta_set_option key: bash tools/exp_kt.sh KEY 'regex of kernel names'   (per kernel name and grid size: calls, mean us)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
KEY=${1:-fuse_head_bwd}; RE=${2:-first_kernel|conv_in_lds|conv32_s1_x3_kernel<false}
for v in 0 1; do
export PTTA_BENCH_OPTIONS=$KEY=$v
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt$v -o x -- python3 bench.py --dtype ${DTYPE:-mixed} --steps 20 --warmup 10 --single-block --no-self-check --no-nlspn --no-cpu-baseline > /dev/null 2> gpurun_out/kt$v.log
echo "== $KEY=$v"; python3 - gpurun_out/kt$v/x_kernel_trace.csv "$RE" <<'PY'
import csv,sys,re,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if re.search(sys.argv[2], n):
        d[(n[:100], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    print(k[0], 'grid', k[1], 'calls', len(v), 'mean us', round(sum(v)/len(v),1))
PY
done
