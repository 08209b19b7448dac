#!/bin/bash
# one replayed step of each dtype as an ordered launch sequence + top kernels (run through gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/mx; rm -rf $O; mkdir -p $O
for DT in ${DTYPES:-mixed}; do
  python3 bench.py --dtype $DT --steps 30 --warmup 10 --single-block --no-nlspn --no-cpu-baseline --no-self-check > $O/bench_$DT.json 2> $O/bench_$DT.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$DT -o x -- python3 bench.py --dtype $DT --steps 20 --warmup 10 --single-block --no-nlspn --no-cpu-baseline --no-self-check > /dev/null 2> $O/trace_$DT.log
  python3 tools/trace_sequence.py $O/trace_$DT/x_kernel_trace.csv 25 > $O/seq_$DT.txt
  python3 tools/prof_top.py $O/trace_$DT 40 > $O/top_$DT.txt
  rm -rf $O/trace_$DT
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/mx/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms_per_step', d['ms_per_step'], 'plain', d['config'].get('ms_per_step_without_frame_pipelining'), 'step frac', d['step_roofline']['hbm_frac_per_gpu'])
        for k,v in d['roofline_by_class'].items():
            if isinstance(v,dict): print('   %-16s launches %5.1f us %7.1f bytes %6.1f MB frac %s' % (k, v['launches_per_step'], v['us_per_step'], v['alg_bytes_per_step']/1e6, v['hbm_frac']))
    except Exception as e: print(f, 'ERR', e)
PY
