#!/bin/bash
# Closing evidence pass of a round (one gpurun call): tools/gpu_close.sh <rNN>
#   1. PMC passes (HBM traffic, MFMA utilisation) of the conv families + the normalisation backward at bench shapes, merged
#      into profiles/<rNN>_pmc.json ON THE BOX so that the bench line below cites this round's counters
#   2. the full GPU suite + smoke
#   3. bench lines WITH roofline of configs 3 (headline; per-shape table), 1, 2, 4, 0
#   4. rocprofv3 --kernel-trace --stats of the headline bench, kernel trace of replayed steps (gaps, per-step table), batch scan
# Everything lands under gpurun_out/<rNN>z/ (copy what is to be judged into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r06}; OUT=gpurun_out/${R}z; mkdir -p $OUT; export TMPDIR=/tmp
make -C twingan_amd/csrc kbench > /dev/null 2>&1
SPECS="E256a fwd 64;E256a dgrad 64;E256a wgrad 64;E256b fwd 48;E256b dgrad 48;E256b wgrad 64;E128a fwd 64;E128a wgrad 64;E128b fwd 48;E128b wgrad 64;E64a fwd 64;E64a wgrad 64;E64b fwd 48;E64b wgrad 64;E32a fwd 64;E32a wgrad 64;E32b fwd 48;E16 fwd 64;E16 dgrad 64;E16 wgrad 64;E8 fwd 64;E8 wgrad 64;G64a fwd 64;G64a wgrad 64;G32a fwd 64;G32a wgrad 64;G16a fwd 64;G256a fwd 64;G128a fwd 64"
bash tools/pmc_kernels.sh "$SPECS" > $OUT/pmc_kernels.txt 2>&1
bash tools/pmc_norm.sh > $OUT/pmc_norm.txt 2>&1
python - "$R" <<'PY'
import json, sys
r = sys.argv[1]
a = json.load(open('gpurun_out/pmc2/summary.json'))
b = json.load(open('gpurun_out/pmc_norm/summary.json'))
a['note'] += ('; tg_norm_act_bwd rows: tools/pmc_norm.sh (the launch pair norm_act_bwd1_kernel + norm_act_bwd2_part_kernel, FETCH_SIZE '
              'and WRITE_SIZE in separate passes); every row measured by tools/gpu_close.sh on the final build of the round')
a['kernels'] += b['kernels']
for p in ('profiles/%s_pmc.json' % r, 'gpurun_out/%sz/%s_pmc.json' % (r, r)):
  json.dump(a, open(p, 'w'), indent=1)
print(len(a['kernels']), 'PMC rows')
PY
bash tools/gpu_pass.sh ${R}z full smoke bench:3 prof:3 trace:3 scan bench:1 bench:2 bench:4 bench:0 > $OUT/pass.txt 2>&1
tail -40 $OUT/pass.txt
