#!/bin/bash
# flash attention: op + model tests, config-4 bench (flash vs composed)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "flash or attention or softmax or bgemm" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "attention or sn or fp16" 2>&1 | tail -3
timeout 200 python bench.py --config 4 --steps 10 --warmup 3 2>gpurun_out/c4_flash.err > gpurun_out/c4_flash.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c4_flash.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
for f in d['roofline']['families'][:8]: print(f)
PY
