# A/B on odd-m shapes: step / solve / K x of the grid, and the theta path at m = 3 (libs: "base" = the built library)
for v in "$@"; do echo "== $v"; if [ $v != base ]; then export SIP_LQR_LIB=$PWD/sip_optimal_control_amd/lib/diag/libkkt_$v.so; else unset SIP_LQR_LIB; fi
python tools/kkt_grid_times.py 2>/dev/null | grep -E "^\| \((12|8), (1|3)\)" | cut -d'`' -f1,3 | cut -c1-100
python tests/bench_kkt.py --theta 8 --m 3 --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('theta m=3:', round(d['ms_factor_theta'],3), round(d['ms_solve_theta'],3), round(d['ms_add_Kx_to_y_theta'],3), d['max_rel_err_vs_oracle'])"
done
