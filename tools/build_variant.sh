#!/bin/bash
# tools/build_variant.sh <name> <file.hip> "<extra flags>" [<file.hip> "<flags>" ...]: an A/B build of the library
# (tools/ab/<name>.so, same ABI, loaded with TG_LIB_PATH) in which the listed sources get extra compiler flags
set -e
NAME=$1; shift
B=/root/repo/gpurun_out/build_$NAME
mkdir -p $B/twingan_amd/csrc $B/include /root/repo/tools/ab
cp /root/repo/twingan_amd/csrc/*.hip /root/repo/twingan_amd/csrc/*.h /root/repo/twingan_amd/csrc/Makefile $B/twingan_amd/csrc/
cp /root/repo/twingan_amd/csrc/*.o $B/twingan_amd/csrc/ 2>/dev/null || true      # start from the in-tree objects
cp /root/repo/include/twingan_hip.h $B/include/
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable"
cd $B/twingan_amd/csrc
while [ $# -gt 0 ]; do
  f=$1; fl=$2; shift 2
  extra=""; [ $f = flash.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc $BASE $extra $fl -c $f -o ${f%.hip}.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o /root/repo/tools/ab/$NAME.so
ls -la /root/repo/tools/ab/$NAME.so
