#!/bin/bash
# Precision budget of the mixed mode, measured with the REAL kernels against the CPU oracle (run through gpurun; ~3 min):
# every combination of the three narrow classes (proxy chain, data gradients, heads) on the 352x1216 / 256x320 frames, the ten-step
# sequence and both cosine-gate fixtures' configuration, plus the step time of each combination.  Output: profiles/r05_precision_budget.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/budget; rm -rf $O; mkdir -p $O
OUT=$O/r05_precision_budget.txt
{
echo "# Precision budget, MSG_CHN 1layer, HIP kernels vs the PyTorch-CPU fp32 oracle (tools/accuracy_report.py; bar: depth 1e-3, target <= 3e-4)"
echo "# classes: proxy = the no_grad zero-image pass (narrow maps, 1 MFMA); backward = every data gradient; heads = proj / pred GEMMs + embeddings"
echo "# columns per step: depth_train depth_eval (rel. MAE) | emb ref (rel. MAE) | grad_w rel. MAE, max-normalised max error, sign flips | param_w | max rel. error of loss_info"
} > $OUT
run() {  # label, dtype, keep, size, steps, extra
  echo "== $1 | size $4 steps $5 $6" >> $OUT
  python3 tools/accuracy_report.py --dtype $2 --keep "$3" --size $4 --steps $5 $6 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    r = json.loads(l)
    li = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(r['loss_info'], r['loss_info_ref']))
    print('  s%d  %.2e %.2e | %.1e %.1e | %.2e %.2e %.4f | %.2e | %.1e   (L_cos %.3f, depth move %.1e)' % (r['step'], r['depth_train'], r['depth_eval'], r['emb'], r['ref'], r['grad_w'], r['grad_w_relmax'], r['sign_flips'], r['param_w'], li, r['loss_info_ref'][3], r['depth_move']))
" >> $OUT
}
for CFG in "fp32:fp32:" "heads-only:mixed:proxy,backward" "backward-only:mixed:proxy,heads" "proxy-only:mixed:backward,heads" "proxy+heads:mixed:backward" "ALL-NARROW(shipped):mixed:"; do
  L=${CFG%%:*}; R=${CFG#*:}; D=${R%%:*}; K=${R#*:}
  echo "" >> $OUT; echo "######## $L  (dtype $D, kept at fp32/bf16x3: ${K:-none})" >> $OUT
  run "$L" $D "$K" 352x1216 3 ""
  run "$L" $D "$K" 256x320 3 ""
  run "$L" $D "$K" 64x96 10 "--frame0 100"
  run "$L gate below (head_bias 5, w_cos 300)" $D "$K" 64x96 3 "--w-cos 300 --head-bias 5.0"
  run "$L gate above (head_bias 3.5, w_cos 300)" $D "$K" 64x96 3 "--w-cos 300 --head-bias 3.5"
  T=$(PTTA_BENCH_KEEP="$K" python3 bench.py --dtype $D --steps 40 --warmup 10 --single-block --no-nlspn --no-cpu-baseline --no-self-check 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step pipelined, %.3f call by call' % (d['ms_per_step'], d['config']['ms_per_step_without_frame_pipelining']))")
  echo "== $L | step time at 352x1216 (one block of 40 steps, this box): $T" >> $OUT
done
cat $OUT
