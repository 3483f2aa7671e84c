FEM_OPTS="-pc_factor_hipmi355x_trisolve_order level" python tests/tools/fem_ilu_apply.py 2>&1 | tail -1
FEM_OPTS="" python tests/tools/fem_ilu_apply.py 2>&1 | tail -1
bash petsc-dev_amd/csrc/variants/build_tri_trace.sh && MI355X_KERNELS_LIB=$PWD/petsc-dev_amd/csrc/variants/libmi355x_kernels_tritrace.so python tests/tools/tri_trace_nodes.py 2>&1 | head -12
