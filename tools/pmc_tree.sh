#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): SQ counter passes of the fused tree kernel (tools/bench_tree.py), per-kernel means by
# tools/pmc_table.py.  usage: tools/pmc_tree.sh <tag>      (raw output under ${PMC_OUT:-gpurun_out/prof}/<tag>)
set -o pipefail
TAG=${1:-tree_pmc}
OUT=${PMC_OUT:-gpurun_out/prof}/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- /usr/bin/python3 tools/bench_tree.py --steps 3 > $OUT/sq1.log 2>&1 || echo "sq1 failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/sq2 -- /usr/bin/python3 tools/bench_tree.py --steps 3 > $OUT/sq2.log 2>&1 || echo "sq2 failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $OUT/sq3 -- /usr/bin/python3 tools/bench_tree.py --steps 3 > $OUT/sq3.log 2>&1 || echo "sq3 failed"
python3 tools/pmc_table.py $OUT tree_factor_solve_qw16
