"""Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the SpMV kernel uses
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads exactly 1/2 of a 16-B-per-lane stream; other widths are
uncalibrated).  Known byte counts: n = 2^26 doubles per array (512 MiB, larger than the 256 MiB Infinity Cache).
  A: axpy on 16-B aligned arrays      -> double2 loads, 16 B/lane   reads 2 x 8n
  B: axpy on arrays offset by 8 bytes -> scalar loads,   8 B/lane   reads 2 x 8n
  C: pack with identity index         -> int loads 4 B/lane + 8-B gathers (coalesced)  reads 4n + 8n
Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/fetch_calib.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu import Dev  # noqa: E402

dev = Dev()
k = dev.k
n = 1 << 26
x = dev.alloc(8 * (n + 2)); y = dev.alloc(8 * (n + 2))
k.mi355x_vec_set(dev.h, n + 2, 1.0, x); k.mi355x_vec_set(dev.h, n + 2, 2.0, y)
idx = dev.put(np.arange(n, dtype=np.int32))
for _ in range(3):
    k.mi355x_vec_axpy(dev.h, n, 0.5, x, y)                                           # A
    k.mi355x_vec_axpy(dev.h, n, 0.5, C.c_void_p(x.value + 8), C.c_void_p(y.value + 8))   # B
    k.mi355x_pack(dev.h, n, idx, x, y)                                               # C
dev.sync()
print("calibration kernels done: n = %d doubles per array" % n)
