for B in 512 768 1024 2048; do
  PTTA_S1_BLOCKS=$B python bench.py --steps 50 --warmup 10 --no-nlspn --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('blocks', $B, 'ms', round(d['ms_per_step'],4), 'class1 us', round(d['roofline']['avg_launch_us'],2))"
done
