#!/bin/bash
# A/B library with another build of sip_kkt_amd.hip: tools/kkt_ab_build.sh <name> [extra -D flags...]
# -> sip_optimal_control_amd/lib/diag/libkkt_<name>.so (the other objects come from the last full build)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build/ab sip_optimal_control_amd/lib/diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c sip_optimal_control_amd/csrc/sip_kkt_amd.hip -o build/ab/kkt_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/sip_lqr_amd/sip_lqr_amd.o $(ls build/obj/qw16_split_*/qw16_split_*.o) build/obj/sip_lqr_tree/sip_lqr_tree.o build/ab/kkt_$NAME.o build/obj/tree_qw16/tree_qw16.o build/obj/chain_mt16/chain_mt16.o $(ls build/obj/qw16_extra_*/qw16_extra_*.o) build/obj/build_stamp.o -o sip_optimal_control_amd/lib/diag/libkkt_$NAME.so
echo sip_optimal_control_amd/lib/diag/libkkt_$NAME.so
