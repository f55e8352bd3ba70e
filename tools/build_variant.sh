# Link a variant of the library with ONE kernel file replaced (ablation / A-B experiments on the same GPU box):
#   bash tools/build_variant.sh <name> <path/to/variant.hip> <stem it replaces, e.g. conv3d> [extra hipcc flags...]
# -> gpurun_scratch/lib_<name>.so ; run with DV_LIB_PATH=gpurun_scratch/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
N=$1; SRC=$2; STEM=$3; shift 3
mkdir -p gpurun_scratch /tmp/dv_variant
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -ffp-contract=on -fno-slp-vectorize \
  -Idiffuvolume_amd/csrc -Iinclude "$@" -c "$SRC" -o /tmp/dv_variant/$N.o
OBJS=$(ls diffuvolume_amd/csrc/build/*.o | grep -v "/$STEM.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o gpurun_scratch/lib_$N.so $OBJS /tmp/dv_variant/$N.o
echo built gpurun_scratch/lib_$N.so
