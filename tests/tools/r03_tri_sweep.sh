set -e
timeout -k 10 900 python -m pytest tests/test_host_gpu.py -m gpu -x -q -k "ilu or node_blocked or gave_up or icc or split_role" 2>&1 | tail -3
timeout -k 10 300 python tests/tools/fem_ilu_apply.py 2>&1 | tail -1
FEM_OPTS="-pc_factor_hipmi355x_trisolve_order column" timeout -k 10 300 python tests/tools/fem_ilu_apply.py 2>&1 | tail -1
MI355X_TRISOLVE_SPLIT=0 FEM_OPTS="-pc_factor_hipmi355x_trisolve_order column" timeout -k 10 300 python tests/tools/fem_ilu_apply.py 2>&1 | tail -1
