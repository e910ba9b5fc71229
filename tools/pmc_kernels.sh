#!/bin/bash
# HBM traffic and MFMA utilisation of the conv kernel families from PMC counters (run on the GPU box via gpurun).
# One counter set per pass, counters only (no --kernel-trace / --stats in the same run; MI355X_MICROARCH.md: TCC has 4
# slots -- FETCH_SIZE costs 3, WRITE_SIZE 2 -- SQ 8, GRBM 2).  Output: gpurun_out/pmc2/<case>_<op>_n<batch>_<set>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
SPECS=${1:-"E256a fwd 64;E256a dgrad 64;E256a wgrad 64;E256b fwd 48;E256b wgrad 64;E128a fwd 64;E128a wgrad 64;E128b fwd 48;E64a fwd 64;E64a wgrad 64;E64b fwd 48;E64b wgrad 64;E32a fwd 64;E32a wgrad 64;E32b fwd 48;E32b wgrad 64;E16 fwd 64;E16 dgrad 64;E16 wgrad 64;E8 fwd 64;G64a fwd 64;G64a wgrad 64;G32a fwd 64;G16a fwd 64"}
IFS=';' read -ra LIST <<< "$SPECS"
for spec in "${LIST[@]}"; do
  set -- $spec
  for cs in FETCH "FETCH_SIZE" WRITE "WRITE_SIZE" MFMA "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
    if [ -z "$tag" ]; then tag=$cs; continue; fi
    out=$OUT/$1_$2_n$3_$tag
    timeout 200 rocprofv3 --pmc $cs --output-format csv -d $out -o pmc -- $REPO/tools/kbench.bin $1 --op $2 --batch $3 --nocheck --iters 3 > $out.log 2>&1
    tag=
  done
done
cd $REPO
python tools/pmc_parse.py gpurun_out/pmc2 > gpurun_out/pmc2/summary.json
python -c "
import json; d=json.load(open('gpurun_out/pmc2/summary.json'))
for e in d['kernels']: print('%-7s %-5s n%-3d %-42s traffic/alg %s  mfma_util %s' % (e['case'], e['op'], e['n'], e['kernel'][:42], e.get('traffic_over_algorithmic'), e.get('mfma_util')))
"
