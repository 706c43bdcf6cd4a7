#!/bin/bash
# Diagnostic (stamped) build of the HIP library -> sip_optimal_control_amd/lib/diag/libsip_lqr_amd.so
set -e
cd "$(dirname "$0")/.."
mkdir -p sip_optimal_control_amd/lib/diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 -DSIP_LQR_STAMPS -DSIP_QW16_NO_EXTRA \
  sip_optimal_control_amd/csrc/sip_lqr_amd.hip sip_optimal_control_amd/csrc/sip_lqr_tree.hip sip_optimal_control_amd/csrc/sip_kkt_amd.hip sip_optimal_control_amd/csrc/tree_qw16.hip sip_optimal_control_amd/csrc/qw16_split.hip -o sip_optimal_control_amd/lib/diag/libsip_lqr_amd.so
echo sip_optimal_control_amd/lib/diag/libsip_lqr_amd.so
