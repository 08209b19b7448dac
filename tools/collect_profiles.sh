#!/bin/bash
# Round-6 profile artifacts of the FINAL tree (run through gpurun; outputs under gpurun_out/r05/, copy into profiles/).  For each mode
# (mixed = the headline / bench default, fp32):
#   1. rocprofv3 --kernel-trace --stats of the bench command (as the driver runs it: direct launches, three streams) + the bench JSON of that run
#   2. ONE step of the timed region as its ordered launch sequence (from the same trace) and one step of the one-stream profiling leg, grouped by kernel
#   3. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM traffic per launch of the dominant kernel class; the counters calibrated on the
#      same kernels over a map of known size), SQ counters
# then 4. what a small convolution launch costs inside a replayed graph; 5. NLSPN and CostDCNet: top kernels
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r06; O=gpurun_out/$R; rm -rf $O; mkdir -p $O
CLASS='conv32_s1_(x3_kernel<(float|unsigned short), true|first_kernel<(float|unsigned short), [23])'      # bench.py's dominant kernel: ReLU-on-load stride-1 convolutions on large maps, forward forms
SHORT="--steps 4 --warmup 2 --single-block --no-nlspn --no-cpu-baseline --no-self-check"
# counter calibration on a known byte count (one 352x1216 map in, one out)
for DT in fp32 narrow; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_f_$DT -o x -- python3 tools/bench_conv32.py 1 $DT > $O/cal_$DT.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_w_$DT -o x -- python3 tools/bench_conv32.py 1 $DT >> $O/cal_$DT.log 2>&1
done
cat $O/cal_f_fp32/x_counter_collection.csv > $O/cal_fetch.csv; tail -n +2 $O/cal_f_narrow/x_counter_collection.csv >> $O/cal_fetch.csv
cat $O/cal_w_fp32/x_counter_collection.csv > $O/cal_write.csv; tail -n +2 $O/cal_w_narrow/x_counter_collection.csv >> $O/cal_write.csv
python3 tools/traffic_from_pmc.py --calibrate $O/cal_fetch.csv $O/cal_write.csv $((352*1216*32*4)) $O/${R}_pmc_calibration.json > /dev/null
for DT in mixed fp32; do
  BENCH="bench.py --dtype $DT --steps 20 --warmup 10 --no-nlspn --no-cpu-baseline --no-self-check"      # (no side runs of the other mode / launch form: the stats are this mode's launches only)
  # 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_graph -o x -- python3 $BENCH > $O/${R}_${DT}_bench_under_rocprofv3_kernel_trace.json 2> $O/trace_graph_$DT.log
  python3 tools/prof_top.py $O/trace_graph 50 > $O/${R}_${DT}_kernel_stats_top.txt
  cp $O/trace_graph/x_kernel_stats.csv $O/${R}_${DT}_kernel_stats.csv
  python3 - "$O/trace_graph/x_kernel_stats.csv" "$CLASS" >> $O/${R}_${DT}_kernel_stats_top.txt <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r['Name'])]
n = sum(int(r['Calls']) for r in rows); t = sum(float(r['TotalDurationNs']) for r in rows)
print('\nroofline class (%s): %d launches, %.2f ms -> %.2f us per launch' % (sys.argv[2], n, t / 1e6, t / 1e3 / max(n, 1)))
PY
  # 1b: the same command with every launch on ONE stream (no frame pipelining, no second stream) -- the condition bench.py's `roofline` leg
  # measures its kernels in; in the three-stream trace above the class's launches overlap other chains and read ~20 % longer
  PTTA_PIPELINE=0 PTTA_BENCH_OPTIONS=aux_stream=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_one -o x -- python3 $BENCH > /dev/null 2> $O/trace_one_$DT.log
  { echo "# bench.py --dtype $DT --steps 20 --warmup 10, PTTA_PIPELINE=0, option aux_stream=0: every launch on one stream (rocprofv3 --kernel-trace --stats)"; python3 tools/prof_top.py $O/trace_one 12; } > $O/${R}_${DT}_kernel_stats_one_stream.txt
  python3 - "$O/trace_one/x_kernel_stats.csv" "$CLASS" >> $O/${R}_${DT}_kernel_stats_one_stream.txt <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r['Name'])]
n = sum(int(r['Calls']) for r in rows); t = sum(float(r['TotalDurationNs']) for r in rows)
print('\nroofline class (%s): %d launches, %.2f ms -> %.2f us per launch' % (sys.argv[2], n, t / 1e6, t / 1e3 / max(n, 1)))
PY
  rm -rf $O/trace_one
  # 2
  python3 tools/trace_sequence.py $O/trace_graph/x_kernel_trace.csv 40 > $O/${R}_${DT}_step_sequence_graph.txt
  python3 tools/trace_step.py $O/trace_graph/x_kernel_trace.csv 140 > $O/${R}_${DT}_step_trace_summary.txt
  # 3
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o x -- python3 bench.py --dtype $DT $SHORT > /dev/null 2> $O/pmc_fetch_$DT.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o x -- python3 bench.py --dtype $DT $SHORT > /dev/null 2> $O/pmc_write_$DT.log
  python3 tools/traffic_from_pmc.py $O/pmc_fetch/x_counter_collection.csv $O/pmc_write/x_counter_collection.csv "$CLASS" $O/traffic_$DT.json $O/${R}_pmc_calibration.json > /dev/null
  python3 tools/pmc_summary.py "$O/pmc_fetch/x_counter_collection.csv" "$O/pmc_write/x_counter_collection.csv" > $O/${R}_${DT}_pmc_fetch_write_per_kernel.txt
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $O/pmc_sq -o x -- python3 bench.py --dtype $DT $SHORT > /dev/null 2> $O/pmc_sq_$DT.log
  python3 tools/pmc_summary.py "$O/pmc_sq/x_counter_collection.csv" > $O/${R}_${DT}_pmc_sq.txt
  rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/trace_graph
done
# 4
{ echo "== one stride-1 32->32 convolution launch inside a replayed graph (tools/bench_chain.py)"; python3 tools/bench_chain.py 2>&1 | grep -v amdgpu; } > $O/${R}_small_conv_in_graph.txt 2>&1
# 5
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
done
# the big raw CSVs do not travel back (64 MiB cap): keep summaries only
rm -rf $O/other_costdcnet $O/other_nlspn $O/other_nlspn_mixed $O/cal_f_* $O/cal_w_* $O/cal_fetch.csv $O/cal_write.csv
ls -la $O
