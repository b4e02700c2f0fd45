#!/bin/bash
# Build an A/B variant of libvulkansift.so: one .hip file recompiled with extra -D flags, everything else taken from the
# regular object directory. Usage: tools/build_variant.sh <name> <hip file basename, e.g. features> [-DFLAG ...]
# SRC=<path> compiles another source file in its place (e.g. a previous revision: git show HEAD:... > /tmp/x.hip).
# Result: vulkansift_amd/lib/variants/lib_<name>.so (git-ignored; select it with VKSIFT_LIB=<path>).
set -e
cd "$(dirname "$0")/.."
name=$1; file=$2; shift 2
python -m vulkansift_amd.build > /dev/null
OBJ=vulkansift_amd/lib/obj
[ -d $OBJ ] || OBJ=$(python - <<'PY'
from vulkansift_amd import build
print(build.OBJ_DIR)
PY
)
mkdir -p vulkansift_amd/lib/variants
# the per-file flags of the regular build (vulkansift_amd/build.py: HIP_EXTRA — e.g. -fno-slp-vectorize for pyramid / features): a variant
# built without them measures the flag, not the source change
extra=$(python - <<PY
from vulkansift_amd import build
print(" ".join(build.HIP_EXTRA.get("hip/$file.hip", [])))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-gpu-rdc $extra \
  -Iinclude -Ivulkansift_amd/csrc/host -Ivulkansift_amd/csrc -Ivulkansift_amd/csrc/hip "$@" -c ${SRC:-vulkansift_amd/csrc/hip/$file.hip} -o /tmp/variant_${name}_$file.o
objs=$(ls $OBJ/*.o | grep -v "/$file.hip.o" | grep -v "\.asan\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vulkansift_amd/lib/variants/lib_$name.so $objs /tmp/variant_${name}_$file.o \
  -L/opt/rocm/lib -lroctx64 -lm -ldl -Wl,-rpath,/opt/rocm/lib
echo vulkansift_amd/lib/variants/lib_$name.so
