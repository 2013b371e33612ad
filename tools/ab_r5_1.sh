cd $GRAFT_REPO_ROOT
for n in 2000 16000 125000; do python tools/rank_quick.py $n 2>&1 | grep -v "^\[ugs\]" | cut -c1-60; done
for n in 2000 16000 125000; do RQ_SHAPE=aa python tools/rank_quick.py $n 2>&1 | grep -v "^\[ugs\]" | cut -c1-60; done
