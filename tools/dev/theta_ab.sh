mkdir -p gpurun_out/th
timeout -k 10 600 python -m pytest tests/test_gpu_kkt.py -x -q > gpurun_out/th/tests.log 2>&1; tail -3 gpurun_out/th/tests.log
for pipe in 4 0 2 6 8 4; do echo -n "pipe=$pipe: "; SIP_KKT_THETA_PIPE=$pipe python tests/bench_kkt.py --theta 8 --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_factor_theta'], d['ms_solve_theta'], d['max_rel_err_vs_oracle'])"; done
