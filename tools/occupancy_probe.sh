#!/bin/bash
# needs the library built with other occupancy targets next to the default one:
#   cd kanpyo_amd/csrc && for w in 5 6; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DKGPU_POOL_WPE=$w -shared -o ../libkanpyo_gpu_w$w.so kgpu_kernels.hip kgpu_pool.hip kgpu_window.hip kgpu_api.cpp kgpu_index_build.cpp; done
# more wavefronts per CU when LDS is not the limit?  (sentences cut to 65 % of their length) -> gpurun_out/occupancy_probe.txt
mkdir -p gpurun_out
OUT=gpurun_out/occupancy_probe.txt; : > $OUT
run() { timeout 300 python tools/resident_probe.py 4096 8 2>/dev/null | tail -1 | tee -a $OUT; }
for len in 0.65 0.5 1.0; do
  export PROBE_LEN=$len
  KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu.so KGPU_POOL=40:4:48 run
  KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu.so KGPU_POOL=20:2:48 run
  KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w5.so KGPU_POOL=32:4:48 run
  KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w5.so KGPU_POOL=40:5:48 run
  KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w5.so KGPU_POOL=40:4:48 run
  KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w6.so KGPU_POOL=26:4:48 run
  KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w6.so KGPU_POOL=40:6:48 run
  KGPU_LIB=$PWD/kanpyo_amd/libkanpyo_gpu_w6.so KGPU_POOL=20:3:64 run
done
