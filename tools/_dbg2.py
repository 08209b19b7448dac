import sys, os, faulthandler
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tta-depth-completion_amd')
import torch, numpy as np
from proxytta import synth
from tests.util import make_engine
eng, sd, adapted = make_engine(1, 32, 48, 'fp32', dict(max_input_depth=80.0), meta='2layers')
print('engine ok', flush=True)
eng.set_graph(False)
image, sparse = synth.synthetic_frame(0, 32, 48, 1)
d, e, r = eng.forward_train(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda())
torch.cuda.synchronize(); print('fwd ok', float(d.mean()), flush=True)
info, depth = eng.step(torch.from_numpy(image).cuda(), torch.from_numpy(sparse).cuda(), want_depth=True)
torch.cuda.synchronize(); print('step ok', info.cpu().tolist(), flush=True)
