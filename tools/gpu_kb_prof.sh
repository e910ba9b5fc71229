#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace, not API-call events) of kbench cases:
#   tools/gpu_kb_prof.sh "<cases>" <op> <batch> [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; OUT=$REPO/gpurun_out/kbprof/${4:-n$3}_$2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for c in $1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -o kb -- $REPO/tools/kbench.bin $c --op $2 --batch $3 --iters 10 --nocheck > $OUT/$c.log 2>&1
  find $OUT/$c -name "*kernel_trace.csv" -delete
done
python3 - "$OUT" "$2" "$3" <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/*/kb_kernel_stats.csv')):
  print(sys.argv[2], 'n' + sys.argv[3], f.split('/')[-2], ' | '.join('%s %.1f us' % (r['Name'].split('::')[-1][:40], float(r['AverageNs']) / 1e3) for r in list(csv.DictReader(open(f)))[:2]))
PY
