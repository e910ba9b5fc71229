C="E16 E32a E64a"
export TG_TILE_DB=0
for v in base tileprobe1 tileprobe2 tileprobe4 tileprobe7; do
  if [ $v = base ]; then unset LD_PRELOAD; else export LD_PRELOAD=$PWD/tools/ab/$v.so; fi
  echo "== $v"
  for n in 16 64; do bash tools/gpu_kb_prof.sh "$C" fwd $n ${v}_$n 2>&1 | grep -v amdgpu.ids | sed 's/ | pack_weights.*//; s/_ZN12_GLOBAL__N_116//'; done
done
unset LD_PRELOAD
