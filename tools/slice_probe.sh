#!/bin/bash
# needs the library built with other occupancy targets next to the default one:
#   cd kanpyo_amd/csrc && for w in 5 6; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DKGPU_POOL_WPE=$w -shared -o ../libkanpyo_gpu_w$w.so kgpu_kernels.hip kgpu_pool.hip kgpu_window.hip kgpu_api.cpp kgpu_index_build.cpp; done
# fixed LDS slices (one wavefront per workgroup, no page pool) against the shared pools; any-order launches -> gpurun_out/slice_probe.txt
mkdir -p gpurun_out
OUT=gpurun_out/slice_probe.txt; : > $OUT
timeout 200 python tools/anyorder_probe.py 2>&1 | grep "one stream" | tee -a $OUT
run() { timeout 300 python tools/resident_probe.py "$@" 2>/dev/null | tail -1 | tee -a $OUT; }
L=$PWD/kanpyo_amd
for len in 1.0 0.65; do
  export PROBE_LEN=$len
  KGPU_LIB=$L/libkanpyo_gpu.so KGPU_POOL=40:4:48 run 4096 8
  KGPU_LIB=$L/libkanpyo_gpu.so KGPU_POOL=10:1:64 run 4096 8
  KGPU_LIB=$L/libkanpyo_gpu.so KGPU_POOL=10:1:64 run 102400 2
  KGPU_LIB=$L/libkanpyo_gpu.so KGPU_POOL=10:1:64 KGPU_POOL_WG=4096 run 16384 4
  KGPU_LIB=$L/libkanpyo_gpu_w5.so KGPU_POOL=8:1:64 run 4096 8
  KGPU_LIB=$L/libkanpyo_gpu_w5.so KGPU_POOL=8:1:64 run 102400 2
  KGPU_LIB=$L/libkanpyo_gpu_w6.so KGPU_POOL=6:1:64 run 4096 8
  KGPU_LIB=$L/libkanpyo_gpu_w6.so KGPU_POOL=6:1:64 run 102400 2
done
