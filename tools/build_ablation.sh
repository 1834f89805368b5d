#!/bin/bash
# Measurement builds of the GEMM kernels: gemm_x3.hip compiled with -DMMA_ABL=<bits> (see the top of that file) and linked with the
# product objects into scratch/abl/libmma_amd_abl<bits>.so; tools/gemm_micro.py runs them through MMA_LIB_OVERRIDE (ctypes binding).
#   bash tools/build_ablation.sh 2 4 6 14         (here, before the gpurun call: hipcc cross-compiles without a GPU)
# With `post` as the first argument: tower_post.hip with -DMMA_POST_ABL=<bits> (1 no MFMAs, 2 no B loads after the first step, 4 no split,
# 16 no MFMAs in the fp32 forward) into scratch/abl/libmma_amd_post<bits>.so, for tools/post_micro.py.
#   bash tools/build_ablation.sh post 1 2 5 7 16
cd "$(dirname "$0")/../mma_amd/csrc" || exit 1
make -s all || exit 1
mkdir -p ../../scratch/abl
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function"
if [ "$1" = "post" ]; then
  shift
  for b in "$@"; do
    /opt/rocm/bin/hipcc $FLAGS -DMMA_POST_ABL=$b -c tower_post.hip -o ../../scratch/abl/tower_post_abl$b.o || exit 1
    /opt/rocm/bin/hipcc $FLAGS "-DMMA_BUILD_SHA_SUFFIX=\"+postabl$b\"" -c abi.hip -o ../../scratch/abl/abi_post$b.o || exit 1
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scratch/abl/libmma_amd_post$b.so \
        ../../scratch/abl/abi_post$b.o nc_fused.o spmm_rows.o gr_fused.o gemm_x3.o tower.o ../../scratch/abl/tower_post_abl$b.o train_step.o pack.o || exit 1
    echo "built scratch/abl/libmma_amd_post$b.so"
  done
  exit 0
fi
for b in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DMMA_ABL=$b -c gemm_x3.hip -o ../../scratch/abl/gemm_x3_abl$b.o || exit 1
  /opt/rocm/bin/hipcc $FLAGS "-DMMA_BUILD_SHA_SUFFIX=\"+abl$b\"" -c abi.hip -o ../../scratch/abl/abi_abl$b.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scratch/abl/libmma_amd_abl$b.so \
      ../../scratch/abl/abi_abl$b.o nc_fused.o spmm_rows.o gr_fused.o ../../scratch/abl/gemm_x3_abl$b.o tower.o tower_post.o train_step.o pack.o || exit 1
  echo "built scratch/abl/libmma_amd_abl$b.so"
done
