import sys, time
sys.path.insert(0, 'tta-depth-completion_amd'); sys.path.insert(0, '.')
import numpy as np, torch
if 'one' in sys.argv: torch.set_num_threads(1)
from proxytta import synth
from proxytta.staging import FrameStager
H, W = 352, 1216
if 'engine' in sys.argv:
    from tests.util import make_engine
    HP = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, w_sparse_depth=1.0, w_smoothness=2.0, w_cos=0.1, max_input_depth=80.0)
    eng, _, _ = make_engine(1, H, W, 'fp32', HP)
frames = [synth.synthetic_frame(i, H, W, 1) for i in range(4)]
if 'tensor' in sys.argv: frames = [[torch.from_numpy(a) for a in f] for f in frames]
st = FrameStager(1, H, W, copy_stream=torch.cuda.current_stream() if 'same' in sys.argv else (torch.cuda.Stream(priority=-1) if 'prio' in sys.argv else None))
st.submit(*frames[0])
rows = []
T0 = time.perf_counter()
for i in range(12):
    t0 = time.perf_counter(); st.submit(*frames[(i + 1) % 4]); t1 = time.perf_counter()
    st.acquire(); t2 = time.perf_counter()
    if 'engine' in sys.argv and 'step' in sys.argv: eng.step(*st.dev[(st.tail) % st.slots])
    t3 = time.perf_counter()
    st.release(); t4 = time.perf_counter()
    rows.append('%.2f/%.2f/%.2f/%.2f' % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)))
torch.cuda.synchronize()
print(sys.argv[1:], 'total %.2f ms/iter' % (1e3 * (time.perf_counter() - T0) / 12), ' submit/acquire/step/release:', ' '.join(rows), flush=True)
