#!/bin/bash
# Dev tool: device ISA of the one-variant experiment build (F16x3, width 64):  tools/exp_isa.sh NAME [flags]  ->  build/exp/NAME/inst.s
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
D=$ROOT/build/exp/$NAME
mkdir -p $D
CS=${CS:-$ROOT/pinn_elastodynamics_amd/csrc}
/opt/rocm/bin/hipcc --offload-arch=gfx950 ${OPT:--O3} -std=c++17 -I$CS -Wno-unused-value --cuda-device-only -S "$@" -DPINN_INST_OP=F16 -DPINN_INST_SPLIT=3 -DPINN_INST_WIDTH=${W:-64} $CS/pinn_inst.hip -o $D/inst.s
