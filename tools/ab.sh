#!/bin/bash
# A/B of two builds of libtwingan_hip.so on the SAME box: tools/ab/prev.so (saved before a kernel change) vs the
# in-tree build.  usage: tools/ab.sh [steps]   -> alternating bench runs, images/sec of each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
STEPS=${1:-20}
for i in 1 2 3; do
  for which in prev cur; do
    if [ $which = prev ]; then export TG_LIB_PATH=$PWD/tools/ab/prev.so; else unset TG_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-roofline --steps $STEPS --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'])"
  done
done
