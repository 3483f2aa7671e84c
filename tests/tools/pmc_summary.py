"""Turn rocprofv3 --pmc counter_collection CSVs (one pass per counter, as gpurun requires) into the per-kernel summary
bench.py reads for roofline.traffic.

Run on the GPU box (from the repo root; rocprofv3 wants `cd /tmp && export TMPDIR=/tmp` first):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 10 --warmup 2
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 10 --warmup 2
  python3 tests/tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w > gpurun_out/pmc_summary.csv
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads 1/2 of the fetched bytes (profiles/r01_fetch_calibration.md) --
the factor is applied by the reader (bench.py), not here."""
import csv
import glob
import os
import sys
from collections import OrderedDict

NOTES = {"FETCH_SIZE": "FETCH_SIZE reads 1/2 of the fetched bytes on gfx950 (calibrated: profiles/r01_fetch_calibration.md)",
         "WRITE_SIZE": "exact for 16-B stores"}


def main():
    args = sys.argv[1:]
    if args and args[0] == "--stamp":     # first line: hashes of the kernel sources the counters were collected with (bench.py checks them)
        import hashlib
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        print("# " + "; ".join("%s sha256/16 = %s" % (src, hashlib.sha256(open(os.path.join(root, "petsc-dev_amd", "csrc", src), "rb").read()).hexdigest()[:16])
                               for src in ("spmv_csr.hip", "vec_kernels.hip")))
        args = args[1:]
    w = csv.writer(sys.stdout)
    w.writerow(["counter", "kernel", "dispatches", "avg_value_KB", "note"])
    for d in args:
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = OrderedDict()
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    key = (row["Counter_Name"], row["Kernel_Name"])
                    s = acc.setdefault(key, [0, 0.0])
                    s[0] += 1
                    s[1] += float(row["Counter_Value"])
            for (counter, kernel), (n, tot) in acc.items():
                w.writerow([counter, kernel, n, "%.1f" % (tot / n), NOTES.get(counter, "")])


if __name__ == "__main__":
    main()
