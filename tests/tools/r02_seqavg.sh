set -x
mkdir -p gpurun_out/r02b
V=petsc-dev_amd/csrc/variants
for lib in default seq128 seq400; do
  if [ $lib = default ]; then unset MI355X_KERNELS_LIB; else export MI355X_KERNELS_LIB=$PWD/$V/libmi355x_kernels_$lib.so; fi
  python3 tests/tools/cfg4_spmv.py irr 30 > gpurun_out/r02b/irr_$lib.log 2>&1 || exit 1
  python3 tests/tools/cfg4_spmv.py fem 30 > gpurun_out/r02b/fem_$lib.log 2>&1 || exit 1
  PETSC_OPTIONS_EXTRA="-mat_no_inode" python3 tests/tools/cfg4_spmv.py fem 30 > gpurun_out/r02b/fem_noinode_$lib.log 2>&1 || exit 1
done
export MI355X_KERNELS_LIB=$PWD/$V/libmi355x_kernels_seq400.so
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "spmv" > gpurun_out/r02b/pytest_spmv_seq400.log 2>&1
unset MI355X_KERNELS_LIB
python3 bench.py --steps 30 --warmup 5 > gpurun_out/r02b/bench.log 2>&1
tail -n 3 gpurun_out/r02b/*.log
