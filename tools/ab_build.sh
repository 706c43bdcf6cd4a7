#!/bin/bash
# Quick A/B library of the main kernels only (no qw16_extra slices): tools/ab_build.sh <name> [extra -D flags...]
# -> sip_optimal_control_amd/lib/diag/lib<name>.so ; compare with tools/ab.sh on the GPU box.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p sip_optimal_control_amd/lib/diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 -DSIP_QW16_NO_EXTRA "$@" \
  sip_optimal_control_amd/csrc/sip_lqr_amd.hip sip_optimal_control_amd/csrc/sip_lqr_tree.hip sip_optimal_control_amd/csrc/sip_kkt_amd.hip \
  sip_optimal_control_amd/csrc/tree_qw16.hip -o sip_optimal_control_amd/lib/diag/lib$NAME.so
echo sip_optimal_control_amd/lib/diag/lib$NAME.so
