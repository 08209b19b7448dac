cd $GRAFT_REPO_ROOT
for A in 0 1 2 3 4 7 8 10 15 16; do echo "ABL=$A"; PTTA_SMALL_ABL=$A python tools/bench_chain.py 2>&1 | grep -E "1/16|1/8|1/4 x2" | awk '{print "   ", $1, $2, $3, $4}'; done
