#!/bin/bash
# Dev tool: build a one-variant (F16x3, padded width W = 64 unless set) experiment library with extra compiler flags.
#   [W=96] tools/exp_build.sh NAME [extra hipcc flags...]   ->  build/exp/NAME/libpinn_hip.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
D=$ROOT/build/exp/$NAME
W=${W:-64}
mkdir -p $D
echo "PINN_VARIANT(F16, 3, $W)" > $D/variants.def
CS=${CS:-$ROOT/pinn_elastodynamics_amd/csrc}
FLAGS="--offload-arch=gfx950 ${OPT:--O3} -std=c++17 -fPIC -I$CS -Wno-unused-value -Rpass-analysis=kernel-resource-usage $*"
/opt/rocm/bin/hipcc $FLAGS -DPINN_VARIANTS_DEF="\"$D/variants.def\"" -c $CS/pinn_capi.hip -o $D/capi.o > $D/build.log 2>&1 &
/opt/rocm/bin/hipcc $FLAGS -DPINN_INST_OP=F16 -DPINN_INST_SPLIT=3 -DPINN_INST_WIDTH=$W -c $CS/pinn_inst.hip -o $D/inst.o > $D/inst.log 2>&1
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libpinn_hip.so $D/capi.o $D/inst.o
grep -A14 "fused_wave_kernelINS_5OpF16ELi3ELi${W}ELi8" $D/inst.log | grep -E "VGPRs:|ScratchSize|VGPRs Spill" | sed 's/.*remark: *//' | tr '\n' ' '; echo " <- $NAME"
