#!/bin/bash
# in-kernel s_memtime stamps of the MSG_CHN stride-1 conv (TIMING variant): PTTA_S1_STAMPS = 1 + flags (1 plain, 2 bilinear skip, 3 mask, 5 add),
# PTTA_S1_STAMPS_BLOCKS selects the launch size (default 512 = full resolution; 418 / 220 / 110 = the 1/2, 1/4, 1/8 maps)
cd $GRAFT_REPO_ROOT
for F in "$@"; do
  echo "== flags $((F-1))"
  PTTA_GRAPH=0 PTTA_S1_STAMPS=$F python3 bench.py --steps 1 --warmup 1 --no-nlspn --no-cpu-baseline 2>&1 | grep "^blk" | tail -8 | cut -c1-220
done
