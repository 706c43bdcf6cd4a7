#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel trace + separate PMC passes of one
# bench.py workload.  Output under gpurun_out/prof/<tag>/.
# usage: tools/profile_gpu.sh <tag> [bench args...]   (e.g. --workload c4)
set -o pipefail
TAG=${1:-r01}; shift
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- /usr/bin/python3 bench.py $ARGS > $OUT/kt.log 2>&1 || echo "kt failed" >> $OUT/kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- /usr/bin/python3 bench.py $ARGS > $OUT/fetch.log 2>&1 || echo "fetch failed" >> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- /usr/bin/python3 bench.py $ARGS > $OUT/write.log 2>&1 || echo "write failed" >> $OUT/write.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- /usr/bin/python3 bench.py $ARGS > $OUT/sq.log 2>&1 || echo "sq failed" >> $OUT/sq.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- /usr/bin/python3 bench.py $ARGS > $OUT/mfma.log 2>&1 || echo "mfma failed" >> $OUT/mfma.log
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
echo "profile done: $OUT"
