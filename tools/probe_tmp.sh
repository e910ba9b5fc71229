C="E8 G8a"
for v in base imgprobe1 imgprobe2 imgprobe4 imgprobe8 imgprobe15; do
  if [ $v = base ]; then unset LD_PRELOAD; else export LD_PRELOAD=$PWD/tools/ab/$v.so; fi
  echo "== $v"; bash tools/gpu_kb_prof.sh "$C" fwd 32 $v 2>&1 | grep -v amdgpu.ids
done
unset LD_PRELOAD
