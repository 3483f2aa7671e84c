set -e
python -m pytest tests/test_host_gpu.py -m gpu -x -q -k "ilu0 or node_blocked or gave_up" 2>&1 | tail -3
V=petsc-dev_amd/csrc/variants
for o in "-pc_factor_hipmi355x_trisolve_order level" ""; do FEM_OPTS="$o" python tests/tools/fem_ilu_apply.py 2>&1 | tail -1; done
