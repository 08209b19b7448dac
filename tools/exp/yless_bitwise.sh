#!/bin/bash
# libptta_hip.so (BatchNorm backward masks from x) against libptta_hip.alt.so (built with -DGBN_NO_YLESS: masks from the saved output): bitwise
cd $GRAFT_REPO_ROOT
L=tta-depth-completion_amd/proxytta
cp $L/libptta_hip.so /tmp/libA.so
python3 tools/exp/yless_dump.py /tmp/a.npz 2>&1 | tail -1
python3 tools/exp/yless_dump.py /tmp/a2.npz 2>&1 | tail -1
cp $L/libptta_hip.alt.so $L/libptta_hip.so
python3 tools/exp/yless_dump.py /tmp/b.npz 2>&1 | tail -1
python3 tools/exp/yless_dump.py /tmp/b2.npz 2>&1 | tail -1
cp /tmp/libA.so $L/libptta_hip.so
python3 - <<'PY'
import numpy as np
def cmp(x, y, label):
    a, b = np.load(x), np.load(y)
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))]
    print('%s: %d arrays compared, %d differ bitwise (nlspn %d, costdcnet %d)' % (label, len(a.files), len(bad), sum(k.startswith('nlspn') for k in bad), sum(k.startswith('costdc') for k in bad)))
    for k in bad[:2]:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print('   %-44s max abs %.3e  rel-to-mean %.3e  differing elements %d / %d' % (k, d.max(), d.mean() / max(np.abs(b[k]).mean(), 1e-30), int((d > 0).sum()), d.size))
cmp('/tmp/a.npz', '/tmp/a2.npz', 'A vs A (run to run)')
cmp('/tmp/b.npz', '/tmp/b2.npz', 'B vs B (run to run)')
cmp('/tmp/a.npz', '/tmp/b.npz', 'A vs B')
PY
