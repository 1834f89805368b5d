#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the C-ABI library (no GPU needed; sanitizers on the GPU are not available on this pool):
# builds mma_amd/csrc with -fsanitize=address,undefined on the HOST code only (-fno-gpu-sanitize) into a scratch directory and
# drives every entry point of include/mma_amd.h with structured random arguments (tests/sanitize_driver.py).
#   bash tools/sanitize_host.sh [calls_per_function] [seed] [build_dir]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CALLS=${1:-400}; SEED=${2:-0}; OUT=${3:-$(mktemp -d /tmp/mma_asan.XXXXXX)}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
[ -n "$RT" ] || { echo "no ASan runtime under /opt/rocm/lib/llvm"; exit 3; }
cd "$ROOT/mma_amd/csrc"
printf "%s\n" abi nc_fused spmm_rows gr_fused gemm_x3 tower tower_post train_step pack | xargs -P 8 -I{} $HIPCC -O1 -g -std=c++17 -fPIC -ffp-contract=off \
  --offload-arch=gfx950 -fno-gpu-rdc -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -Wno-unused-function \
  -c {}.hip -o "$OUT/{}.o"
$HIPCC -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -o "$OUT/libmma_amd.so" "$OUT"/*.o
cd "$ROOT"
# The driver hands FAKE device pointers to launchers whose arguments pass validation: with a GPU visible those launches would
# succeed and fault on the device.  Hide every device from this process (the driver also refuses to run if it can see one).
export HIP_VISIBLE_DEVICES=-1 ROCR_VISIBLE_DEVICES= CUDA_VISIBLE_DEVICES=
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python3 tests/sanitize_driver.py "$OUT/libmma_amd.so" "$CALLS" "$SEED"
