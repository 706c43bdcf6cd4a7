#!/bin/bash
# build a tree-only variant lib: tools... name flags
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab sip_optimal_control_amd/lib/diag
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -save-temps=obj -c sip_optimal_control_amd/csrc/tree_qw16.hip -o build/ab/tree_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/sip_lqr_amd/sip_lqr_amd.o $(ls build/obj/qw16_split_*/qw16_split_*.o) build/obj/sip_lqr_tree/sip_lqr_tree.o build/obj/sip_kkt_amd/sip_kkt_amd.o build/obj/chain_mt16/chain_mt16.o build/ab/tree_$NAME.o $(ls build/obj/qw16_extra_*/qw16_extra_*.o) build/obj/build_stamp.o -o sip_optimal_control_amd/lib/diag/libtree_$NAME.so
echo sip_optimal_control_amd/lib/diag/libtree_$NAME.so
