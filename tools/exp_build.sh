#!/bin/bash
# Dev tool: build a one-variant (F16x3, width 64) experiment library with extra compiler flags.
#   tools/exp_build.sh NAME [extra hipcc flags...]   ->  build/exp/NAME/libpinn_hip.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
D=$ROOT/build/exp/$NAME
mkdir -p $D
echo 'PINN_VARIANT(F16, 3, 64)' > $D/variants.def
CS=${CS:-$ROOT/pinn_elastodynamics_amd/csrc}
FLAGS="--offload-arch=gfx950 ${OPT:--O3} -std=c++17 -fPIC -I$CS -Wno-unused-value -Rpass-analysis=kernel-resource-usage $*"
/opt/rocm/bin/hipcc $FLAGS -DPINN_VARIANTS_DEF="\"$D/variants.def\"" -c $CS/pinn_capi.hip -o $D/capi.o > $D/build.log 2>&1 &
/opt/rocm/bin/hipcc $FLAGS -DPINN_INST_OP=F16 -DPINN_INST_SPLIT=3 -DPINN_INST_WIDTH=64 -c $CS/pinn_inst.hip -o $D/inst.o > $D/inst.log 2>&1
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libpinn_hip.so $D/capi.o $D/inst.o
grep -A14 "fused_wave_kernelINS_5OpF16ELi3ELi64ELi8" $D/inst.log | grep -E "VGPRs:|ScratchSize|VGPRs Spill" | sed 's/.*remark: *//' | tr '\n' ' '; echo " <- $NAME"
