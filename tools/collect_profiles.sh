#!/bin/bash
# Round-3 profile artifacts of the FINAL tree (run through gpurun; outputs under gpurun_out/r03/, copy into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the bench command (graph replay, as the driver runs it) + bench JSON
#   2. the same command kernel by kernel (PTTA_GRAPH=0): per-kernel summary of ONE step
#   3. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (traffic per launch of the dominant kernel), SQ counters
#   4. NLSPN and CostDCNet: top kernels + NLSPN SQ counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
BENCH="bench.py --steps 50 --warmup 10 --no-nlspn --no-cpu-baseline"
# 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_graph -o x -- python3 $BENCH > $O/bench_under_rocprof.json 2> $O/trace_graph.log
python3 tools/prof_top.py $O/trace_graph 40 > $O/r03_fp32_kernel_stats_top.txt
cp $O/trace_graph/x_kernel_stats.csv $O/r03_fp32_kernel_stats.csv
# 2
PTTA_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_step -o x -- python3 $BENCH > /dev/null 2> $O/trace_step.log
python3 tools/trace_step.py $O/trace_step/x_kernel_trace.csv 120 > $O/r03_fp32_step_trace_summary.txt
python3 tools/trace_sequence.py $O/trace_graph/x_kernel_trace.csv > $O/r03_fp32_step_sequence_graph.txt
# 3
PTTA_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o x -- python3 bench.py --steps 8 --warmup 2 --no-nlspn --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.log
PTTA_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o x -- python3 bench.py --steps 8 --warmup 2 --no-nlspn --no-cpu-baseline > /dev/null 2> $O/pmc_write.log
python3 tools/traffic_from_pmc.py $O/pmc_fetch/x_counter_collection.csv $O/pmc_write/x_counter_collection.csv "conv32_s1_(x3|small)_kernel<true" $O/traffic_fp32.json > /dev/null
python3 tools/pmc_summary.py "$O/pmc_fetch/x_counter_collection.csv" "$O/pmc_write/x_counter_collection.csv" > $O/r03_fp32_pmc_fetch_write_per_kernel.txt
PTTA_GRAPH=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $O/pmc_sq -o x -- python3 bench.py --steps 4 --warmup 2 --no-nlspn --no-cpu-baseline > /dev/null 2> $O/pmc_sq.log
python3 tools/pmc_summary.py "$O/pmc_sq/x_counter_collection.csv" > $O/r03_msgchn_pmc_sq.txt
# 4
cat > /tmp/run_other.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tta-depth-completion_amd'))
import bench
which = sys.argv[1]
print(bench.costdcnet_workload(4) if which == 'costdcnet' else bench.nlspn_workload(2, 3))
PY
for W in costdcnet nlspn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/other_$W -o x -- python3 /tmp/run_other.py $W > $O/other_$W.log 2>&1
  python3 tools/prof_top.py $O/other_$W 30 > $O/r03_${W}_top_kernels.txt 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $O/pmc_nlspn -o x -- python3 /tmp/run_other.py nlspn > /dev/null 2> $O/pmc_nlspn.log
python3 tools/pmc_summary.py "$O/pmc_nlspn/x_counter_collection.csv" > $O/r03_nlspn_pmc_sq.txt
# the big raw CSVs do not travel back (64 MiB cap): keep summaries only
rm -rf $O/trace_step $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_nlspn $O/other_costdcnet $O/other_nlspn $O/trace_graph
ls -la $O
