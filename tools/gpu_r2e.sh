#!/bin/bash
# conv statistics epilogue in the model: tests, bench A/B, per-shape table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2e; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_golden.py tests/test_gpu_ops.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
TG_DUMP_SHAPES=$OUT/shapes_eager_step.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2> $OUT/bench.err; echo "exit $?" >> $OUT/bench.log
TG_CONV_STATS=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $OUT/bench_off.log 2> $OUT/bench_off.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $OUT/bench_on.log 2> $OUT/bench_on.err
python - <<'PY'
import json
for f in ('bench','bench_off','bench_on'):
    try:
        l=[x for x in open('gpurun_out/r2e/%s.log'%f) if x.startswith('{')][-1]; d=json.loads(l); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
