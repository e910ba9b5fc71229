#!/bin/bash
# HBM traffic of the dominant conv layer shapes from PMC counters (run on the GPU box via gpurun).
# Separate passes per counter (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md), counters only
# (no --kernel-trace / --stats in the same run).  Output: gpurun_out/pmc/<case>_<op>_<counter>/... csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
cd /tmp
for spec in "G256a fwd 64" "G256a wgrad 64" "G256a dgrad 64" "E256a fwd 32" "E256a wgrad 48" "E128a wgrad 16" "G32a fwd 64" "E32b fwd 32"; do
  set -- $spec
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=$REPO/gpurun_out/pmc/$1_$2_n$3_$ctr
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $out -o pmc -- $REPO/tools/kbench.bin $1 --op $2 --batch $3 --nocheck --iters 3 > $out.log 2>&1
  done
done
cd $REPO
python tools/pmc_parse.py gpurun_out/pmc > gpurun_out/pmc/summary.json
cat gpurun_out/pmc/summary.json
