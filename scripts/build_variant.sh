#!/bin/bash
# Builds a variant of libtrackdlo_hip.so with extra compiler flags into scripts/tmp/libtrackdlo_<name>.so (objects under scripts/tmp/build_<name>/),
# e.g.  bash scripts/build_variant.sh stamps -DTDLO_ESTEP_STAMPS     (the instrumented builds scripts/gpu_estamps.py / gpu_ephases.py / gpu_timeline.py load)
# usage: bash scripts/build_variant.sh <name> [flags ...]
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/trackdlo_amd/csrc
B=$R/scripts/tmp/build_$name
mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result $*"
cd $S
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -c tdlo_device.hip -o $B/tdlo_device.o &
/opt/rocm/bin/hipcc $F -c tdlo_estep2.hip -o $B/tdlo_estep2.o &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -c tdlo_mstep_big.hip -o $B/tdlo_mstep_big.o &
/opt/rocm/bin/hipcc $F -c tdlo_mstep_chain.hip -o $B/tdlo_mstep_chain.o &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -c tdlo_mstep_band.hip -o $B/tdlo_mstep_band.o &
/opt/rocm/bin/hipcc $F -c tdlo_cloud.hip -o $B/tdlo_cloud.o &
/opt/rocm/bin/hipcc $F -c tdlo_reg.hip -o $B/tdlo_reg.o &
/opt/rocm/bin/hipcc $F -x hip -c tdlo_api.cpp -o $B/tdlo_api.o &
/opt/rocm/bin/hipcc $F -x hip -c tdlo_rccl.cpp -o $B/tdlo_rccl.o &
g++ -O3 -std=c++17 -fPIC -ffp-contract=off $* -c tdlo_host.cpp -o $B/tdlo_host.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/tmp/libtrackdlo_$name.so $B/*.o -ldl
ls -la $R/scripts/tmp/libtrackdlo_$name.so
