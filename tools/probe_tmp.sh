C="E16 E32a G32b E64a"
for v in base cur; do
  if [ $v = base ]; then export LD_PRELOAD=$PWD/tools/ab/base.so; else unset LD_PRELOAD; fi
  echo "== $v"; bash tools/gpu_kb_prof.sh "$C" fwd 16 ${v}16 2>&1 | grep -v amdgpu.ids; bash tools/gpu_kb_prof.sh "$C" fwd 32 ${v}32 2>&1 | grep -v amdgpu.ids
done
unset LD_PRELOAD
ROUNDS=3 bash tools/gpu_pass.sh r05w lib:base
