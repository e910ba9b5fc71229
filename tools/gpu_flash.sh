#!/bin/bash
# flash attention: op tests, model tests with attention, config-4 bench A/B
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "flash or attention or softmax or bgemm" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "attention or sn or config4 or fp16" 2>&1 | tail -8
timeout 200 python bench.py --config 4 --steps 10 --warmup 3 2>gpurun_out/c4_flash.err | tee gpurun_out/c4_flash.json
TG_FLASH_ATTENTION=0 timeout 200 python bench.py --config 4 --steps 10 --warmup 3 2>gpurun_out/c4_noflash.err | tee gpurun_out/c4_noflash.json
tail -3 gpurun_out/c4_flash.err
