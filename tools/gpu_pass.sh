#!/bin/bash
# One parameterised GPU pass (replaces the per-round gpu_r2*.sh / gpu_r3*.sh scripts).  Meant for
#   gpurun --timeout N -- 'bash tools/gpu_pass.sh <tag> <step> [<step> ...]'
# Everything is written under gpurun_out/<tag>/.  Steps, run in the order given:
#   t:<expr>          pytest -m gpu -k '<expr>' over tests/ (use '+' for spaces: t:upcat+or+mbstd)
#   f:<file>[:<expr>] pytest -m gpu of one test file (optionally -k '<expr>')
#   full              the whole GPU suite, -x (what the driver runs)
#   record            re-record tests/golden/bench_dispatch_kernels.json from this build (copied to the out dir only)
#   smoke             __graft_entry__.smoke()
#   bench[:C]         default bench line of config C (3): roofline + cpu_baseline + per-shape table (shapes_cC.json)
#   quick[:C]         bench line without roofline / cpu baseline
#   prof[:C]          rocprofv3 --kernel-trace --stats of a short bench run -> prof_cC/ (kernel trace deleted)
#   ab:<VAR=v,...>    interleaved same-box A/B, ROUNDS (2) rounds: default environment vs the given variables (config 3)
#   abc:<C>:<VAR=v,...>  the same on config C
#   lib:<name>        interleaved A/B of the in-tree library vs tools/ab/<name>.so (same ABI)
#   old:<name>        interleaved A/B of this tree vs the complete older tree tools/ab/<name>_tree (git archive + its built library); OLD_ARGS='--config 4' for another configuration
#   trace[:C]         rocprofv3 kernel trace of replayed steps: gaps, overlap, per-kernel table of ONE replayed step
#   py:<VAR=v,..|->:<script+args>  run a python tool (under the given environment) -> py.log
#   scan              batch scan b = 4 8 16 24 32 (ms per step)
#   eager             bench --no-graph (3 steps)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD; TAG=${1:-pass}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROUNDS=${ROUNDS:-2}
nproc > $OUT/box.txt; rocminfo 2>/dev/null | grep -E "gfx9" | head -2 >> $OUT/box.txt
line() { python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])
except Exception as e:
  print('$1', 'FAILED', e)"; }
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  case $kind in
    t)
      expr=${arg//+/ }
      timeout 1500 python -m pytest tests -m gpu -q -k "$expr" --tb=short -p no:cacheprovider > $OUT/pytest_${arg//[^A-Za-z0-9]/_}.log 2>&1
      echo "pytest exit $?" >> $OUT/pytest_${arg//[^A-Za-z0-9]/_}.log; tail -4 $OUT/pytest_${arg//[^A-Za-z0-9]/_}.log ;;
    f)
      file=${arg%%:*}; expr=""; [ "$file" != "$arg" ] && expr=${arg#*:}; expr=${expr//+/ }
      name=$(basename $file .py)_${expr//[^A-Za-z0-9]/_}
      if [ -n "$expr" ]; then timeout 1500 python -m pytest $file -m gpu -q -k "$expr" --tb=short -p no:cacheprovider > $OUT/pytest_$name.log 2>&1
      else timeout 1500 python -m pytest $file -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_$name.log 2>&1; fi
      echo "pytest exit $?" >> $OUT/pytest_$name.log; tail -4 $OUT/pytest_$name.log ;;
    full)
      timeout 1800 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
      echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log ;;
    record)
      TG_RECORD_KERNELS=$REPO/$OUT/bench_dispatch_kernels.json timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_record.log 2>&1
      tail -2 $OUT/pytest_record.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -1 $OUT/smoke.log ;;
    bench)
      c=${arg:-3}
      TG_DUMP_SHAPES=$REPO/$OUT/shapes_c$c.json timeout 600 python bench.py --config $c > $OUT/bench_c$c.log 2> $OUT/bench_c$c.err; echo "exit $?" >> $OUT/bench_c$c.log
      head -c 400 $OUT/bench_c$c.log; echo ;;
    quick)
      c=${arg:-3}
      timeout 300 python bench.py --config $c --no-roofline --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 2> $OUT/quick_c$c.err | tee $OUT/quick_c$c.log | line quick_c$c ;;
    eager)
      timeout 300 python bench.py --no-graph --steps 3 --warmup 1 --no-roofline --no-cpu-baseline 2> $OUT/eager.err | tee $OUT/eager.log | line eager ;;
    prof)
      c=${arg:-3}
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_c$c -o bench -- python $REPO/bench.py --config $c --steps 4 --warmup 1 --no-roofline --no-cpu-baseline > $REPO/$OUT/prof_c$c.log 2>&1)
      find $OUT/prof_c$c -name "*kernel_trace.csv" -delete; find $OUT/prof_c$c -name "*kernel_stats*" | head -2 ;;
    ab|abc)
      c=3; vars=$arg
      if [ $kind = abc ]; then c=${arg%%:*}; vars=${arg#*:}; fi
      for i in $(seq 1 $ROUNDS); do
        for which in default variant; do
          if [ $which = variant ]; then envs=$(echo $vars | tr ',' ' '); else envs=""; fi
          env $envs timeout 300 python bench.py --config $c --no-cpu-baseline --no-roofline --steps ${STEPS:-20} --warmup 3 2> $OUT/ab_${which}_$i.err | line "$which[$vars]" | tee -a $OUT/ab.log
        done
      done ;;
    lib)
      for i in $(seq 1 $ROUNDS); do
        for which in cur $arg; do
          if [ $which = cur ]; then unset TG_LIB_PATH; else export TG_LIB_PATH=$REPO/tools/ab/$which.so; fi
          timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps ${STEPS:-20} --warmup 3 2> $OUT/lib_${which}_$i.err | line "lib:$which" | tee -a $OUT/ab.log
        done
      done; unset TG_LIB_PATH ;;
    old)      # interleaved A/B of this tree against a complete older tree under tools/ab/<name>_tree (its own library + python)
      for i in $(seq 1 $ROUNDS); do
        for which in cur $arg; do
          if [ $which = cur ]; then dir=$REPO; else dir=$REPO/tools/ab/${which}_tree; fi
          (cd $dir && timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps ${STEPS:-20} --warmup 3 ${OLD_ARGS} 2> $REPO/$OUT/old_${which}_$i.err) | line "tree:$which ${OLD_ARGS}" | tee -a $OUT/ab.log
        done
      done ;;
    trace)    # kernel trace of replayed steps -> gaps / overlap of the last one + its per-kernel table (step_kernels_cC.json)
      c=${arg:-3}
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_c$c -o bench -- python $REPO/bench.py --config $c --steps 3 --warmup 2 --no-roofline --no-cpu-baseline > $REPO/$OUT/trace_c$c.log 2>&1)
      f=$(find $OUT/trace_c$c -name "*kernel_trace.csv" | head -1)
      python tools/trace_gaps.py "$f" $OUT/step_kernels_c$c.json > $OUT/gaps_c$c.txt 2>&1; head -12 $OUT/gaps_c$c.txt
      rm -rf $OUT/trace_c$c ;;
    py)       # py:<VAR=v,...|->:<script and args, '+' for spaces>   -> appended to py.log
      vars=${arg%%:*}; cmd=${arg#*:}; cmd=${cmd//+/ }
      envs=""; [ "$vars" != "-" ] && envs=$(echo $vars | tr ',' ' ')
      env $envs timeout 600 python $cmd 2>> $OUT/py.err | tee -a $OUT/py.log ;;
    scan)
      for b in 4 8 16 24 32; do
        timeout 300 python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 20 --warmup 3 2> $OUT/scan_$b.err | line "batch $b" | tee -a $OUT/batch_scan.txt
      done ;;
    *) echo "unknown step $step" ;;
  esac
done
