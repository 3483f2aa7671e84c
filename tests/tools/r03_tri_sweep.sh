set -e
for a in 8 16 32; do for sl in 1 2 4; do
  MI355X_TRISOLVE_AHEAD=$a MI355X_TRISOLVE_SLEEP=$sl timeout -k 10 300 python tests/tools/fem_ilu_apply.py 2>&1 | tail -1
done; done
