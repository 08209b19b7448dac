#!/bin/bash
# libptta_hip.so against libptta_hip.alt.so, MSG_CHN: every dumped array bit for bit (tools/exp/msgchn_dump.py)
cd $GRAFT_REPO_ROOT
L=tta-depth-completion_amd/proxytta
cp $L/libptta_hip.so /tmp/libA.so
python3 tools/exp/msgchn_dump.py /tmp/a.npz 2>&1 | tail -1
cp $L/libptta_hip.alt.so $L/libptta_hip.so
python3 tools/exp/msgchn_dump.py /tmp/b.npz 2>&1 | tail -1
cp /tmp/libA.so $L/libptta_hip.so
python3 - <<'PY'
import numpy as np
a, b = np.load('/tmp/a.npz'), np.load('/tmp/b.npz')
bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))]
print('%d arrays compared, %d differ bitwise' % (len(a.files), len(bad)), bad[:6])
PY
