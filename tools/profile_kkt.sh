#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel trace + separate FETCH_SIZE / WRITE_SIZE passes of the
# Newton-KKT step (tests/bench_kkt.py).  Output under gpurun_out/prof/<tag>/.
set -o pipefail
TAG=${1:-r01_kkt}; shift
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 10 --cpu-seconds 0 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- /usr/bin/python3 tests/bench_kkt.py $ARGS > $OUT/kt.log 2>&1 || echo "kt failed" >> $OUT/kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- /usr/bin/python3 tests/bench_kkt.py $ARGS > $OUT/fetch.log 2>&1 || echo "fetch failed" >> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- /usr/bin/python3 tests/bench_kkt.py $ARGS > $OUT/write.log 2>&1 || echo "write failed" >> $OUT/write.log
echo "profile done: $OUT"
