#!/bin/bash
# Dev tool: ISA + resource usage of ONE fused instantiation (default F16x3, width 64, 8 layers, 4 streams) -> build/x/NAME.s
#   tools/one_isa.sh NAME [-DX_W=96 -DX_NS=5 ... extra flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
CS=${CS:-$ROOT/pinn_elastodynamics_amd/csrc}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$CS -Wno-unused-value --cuda-device-only -S -Rpass-analysis=kernel-resource-usage "$@" $ROOT/build/x/one.hip -o $ROOT/build/x/$NAME.s 2>&1 | grep -E "VGPRs:|ScratchSize|Spill|Occupancy|LDS Size" | sed 's/.*remark: *//' | tr '\n' ' '; echo " <- $NAME"
