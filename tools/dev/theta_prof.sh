# per-kernel times of the theta path: tools/dev/theta_prof.sh <tag> [ENV=VALUE ...]  (run on the GPU box)
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/th/prof_$tag -o theta -- python3 /root/repo/tests/bench_kkt.py --theta 8 --steps 5 > /dev/null 2>&1
