#!/bin/bash
# A/B timing of library variants on the GPU box: tools/ab.sh libA.so libB.so ... (3 interleaved rounds)
# AB_ARGS="--workload c4" selects another bench workload
for round in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "
    SIP_LQR_LIB=$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f us  %.2f Msweeps/s' % (d['roofline']['kernel_ms']*1e3, d['value']/1e6))"
  done
done
