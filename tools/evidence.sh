#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the measurements a round's profiles/ is made from.
# usage: tools/evidence.sh <tag>      (output under gpurun_out/<tag>/ and gpurun_out/prof/<tag>_*)
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for w in c3 c2 c4 kkt; do
  timeout -k 10 200 python bench.py --workload $w --steps 30 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err || echo "bench $w failed"
done
timeout -k 10 200 python tests/bench_kkt.py --theta 8 --steps 10 2>/dev/null | tail -1 > $OUT/bench_theta8.json
timeout -k 10 200 python tools/bench_tree.py 2>/dev/null | tail -1 > $OUT/bench_tree.json
timeout -k 10 300 python tools/kkt_grid_times.py > $OUT/kkt_grid.md 2>/dev/null
timeout -k 10 300 tools/profile_gpu.sh ${TAG}_c3 > $OUT/prof_c3.log 2>&1
timeout -k 10 300 tools/profile_gpu.sh ${TAG}_c4 --workload c4 > $OUT/prof_c4.log 2>&1
timeout -k 10 300 tools/profile_gpu.sh ${TAG}_c2 --workload c2 > $OUT/prof_c2.log 2>&1
timeout -k 10 300 tools/profile_gpu.sh ${TAG}_c3full --workload c3 --layout full > $OUT/prof_c3full.log 2>&1
timeout -k 10 300 tools/profile_kkt.sh ${TAG}_kkt > $OUT/prof_kkt.log 2>&1
timeout -k 10 300 tools/profile_kkt.sh ${TAG}_theta8 --theta 8 > $OUT/prof_theta8.log 2>&1
timeout -k 10 120 python bench.py --workload c4 --steps 10 --no-cpu-baseline > $OUT/bench_c4_again.json 2>/dev/null
( timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --blocks 2 > $OUT/bench_g2_rehearsal.json 2> $OUT/bench_g2_rehearsal.err ) || echo "2-rank rehearsal failed"
# counters that could separate DRAM from Infinity-Cache traffic (MI355X_MICROARCH.md: FETCH_SIZE counts fabric requests)
(rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) | grep -iE "HBM|DRAM|MALL|EA0?_|TCC_EA|MC_|UMC|DF_|TCC_.*(REQ|MISS|HIT)" | head -150 > $OUT/counters_memory_side.txt
echo "evidence done: $OUT"
