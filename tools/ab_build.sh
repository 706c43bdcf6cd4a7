#!/bin/bash
# Quick A/B library: tools/ab_build.sh <name> [extra -D flags...]
# -> sip_optimal_control_amd/lib/diag/lib<name>.so ; compare with tools/ab.sh on the GPU box.
# Only sip_lqr_amd.hip is compiled, with the C3 kernel alone (-DSIP_QW16_QUICK); the tree / KKT objects
# come from the last full build (build/obj), so run `python -c "import __graft_entry__ as g; g.build()"` first.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p sip_optimal_control_amd/lib/diag build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DSIP_QW16_NO_EXTRA -DSIP_QW16_QUICK "$@" \
  -save-temps=obj -c sip_optimal_control_amd/csrc/sip_lqr_amd.hip -o build/ab/$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DSIP_QW16_QUICK "$@" \
  -c sip_optimal_control_amd/csrc/qw16_split.hip -o build/ab/${NAME}_split.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/ab/$NAME.o build/ab/${NAME}_split.o build/obj/sip_lqr_tree/sip_lqr_tree.o \
  build/obj/sip_kkt_amd/sip_kkt_amd.o build/obj/tree_qw16/tree_qw16.o -o sip_optimal_control_amd/lib/diag/lib$NAME.so
echo sip_optimal_control_amd/lib/diag/lib$NAME.so
