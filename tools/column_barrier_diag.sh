#!/bin/bash
# Builds the diagnosis variant of the library (the column kernel with -DINTERPN_COLUMN_DIAG) next to the product
# one and runs tools/column_barrier_diag.py on it:   gpurun --timeout 900 -- bash tools/column_barrier_diag.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V="$R/tools/_variants/diag"
mkdir -p "$V/build" "$R/gpurun_out/column_barrier_diag"
cd "$R/interpn_amd/csrc" || exit 1
[ -f build/abi_create.o ] || make -j8 > /dev/null || exit 1   # the object files do not travel with gpurun (.gpurunignore)
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-function"
if [ ! -f "$V/libinterpn_hip.so" ] || [ cubic_column.h -nt "$V/libinterpn_hip.so" ]; then
  /opt/rocm/bin/hipcc $FLAGS -DINTERPN_COLUMN_DIAG -c k_cubic_column.hip -o "$V/build/k_cubic_column.o" || exit 1
  OBJS=$(ls build/*.o | grep -v k_cubic_column.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/libinterpn_hip.so" $OBJS "$V/build/k_cubic_column.o" || exit 1
fi
cd "$R" && INTERPN_AMD_LIB="$V/libinterpn_hip.so" timeout -k 10 300 python3 tools/column_barrier_diag.py "$@" 2>&1 | grep -v amdgpu.ids | tee "$R/gpurun_out/column_barrier_diag/out.txt"
