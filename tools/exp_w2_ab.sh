for o in bwd_w2=1 bwd_w2=0; do PTTA_BENCH_OPTIONS=$o python bench.py --steps 50 --warmup 10 --no-self-check --no-nlspn --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$o', round(d['ms_per_step'],4), ' '.join('%s %.0f/%d' % (k, v['us_per_step'], v['launches_per_step']) for k,v in d['roofline_by_class'].items() if isinstance(v,dict)))"; done
