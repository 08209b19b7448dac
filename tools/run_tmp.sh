#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tta-depth-completion_amd')
import torch, numpy as np
from tests.test_gpu_edge_cases import *
from tests.test_gpu_edge_cases import _oracle
for shape in [(1, 256, 320), (1, 480, 640), (3, 48, 80)]:
    n,h,w=shape
    eng, sd, adapted = make_engine(n, h, w, 'fp32', HP)
    o=_oracle()
    image, sparse = synth.synthetic_frame(5, h, w, n, density=1500.0 / (h * w) if h >= 256 else 0.05, dmin=0.2, dmax=8.0)
    r = o.step(torch.from_numpy(image), torch.from_numpy(sparse))
    eng.step(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda())
    ref_eval = o.forward_eval(torch.from_numpy(image), torch.from_numpy(sparse))
    d_eval = eng.forward_eval(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda())
    print(shape, 'post-update eval rel MAE %.2e' % rel_mae(d_eval, ref_eval))
    eng.close()
PY
timeout 2400 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_nlspn.py -q 2>&1 | tail -12
