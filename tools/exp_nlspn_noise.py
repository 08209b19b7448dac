"""Run-to-run spread of one NLSPN TTA step on two identical engines (the propagation gradient uses float atomics):
largest rel. MAE over the adapted tensors, fused step vs fused step."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tta-depth-completion_amd'))
import torch
from test_gpu_nlspn import make_nlspn, nlspn_frame, rel_mae
n, h, w = 1, 32, 64
raw, image1, sparse = [torch.from_numpy(x).cuda() for x in nlspn_frame(0, h, w, n)]
for rep in range(5):
    e1, sd1, ad1 = make_nlspn(n, h, w)
    e2, sd2, ad2 = make_nlspn(n, h, w)
    e1.step(image1, sparse, loss_image=raw); e2.step(image1, sparse, loss_image=raw)
    torch.cuda.synchronize()
    worst = max((float(rel_mae(ad2[k][0], ad1[k][0])), k) for k in ad1)
    print('rep', rep, 'worst rel_mae %.3e' % worst[0], worst[1])
