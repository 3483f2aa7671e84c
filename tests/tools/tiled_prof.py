"""Timeline of the column-tiled SpMV's workgroups from the TL_PROFILE build of the kernel library (csrc/variants/build_tiled.sh prof
-DTL_PROFILE): when each panel's workgroup started and ended (100 MHz wall clock), where it ran, how long its wavefront 0 waited in
tile-switch barriers.   MI355X_KERNELS_LIB=.../libmi355x_kernels_prof.so python3 tests/tools/tiled_prof.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out = "/tmp/tl_prof.bin"
    env = dict(os.environ, MI355X_TILED_PROF=out, CFG4_CACHE=os.environ.get("CFG4_CACHE", "/tmp"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "tiled_probe.py"), "irr", "1024"], env=env, capture_output=True, text=True)
    print(r.stdout[-600:])
    if r.returncode:
        print(r.stderr[-2000:]); sys.exit(1)
    a = np.fromfile(out, dtype=np.uint64).reshape(-1, 8)
    t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0                     # microseconds
    dur = t1 - t0
    xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
    hw = a[:, 2].astype(np.int64) & 0xffffffff
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    print("panels %d  span %.1f us  workgroup duration min / median / max: %.1f / %.1f / %.1f us" % (a.shape[0], t1.max(), dur.min(), np.median(dur), dur.max()))
    print("sum of durations / span = %.1f workgroups active on average (512 slots)" % (dur.sum() / t1.max()))
    bar = a[:, 3].astype(np.float64) / 2400.0                              # us at 2.4 GHz
    print("wavefront 0 in tile-switch barriers: median %.1f us of %.1f (%.0f%%)" % (np.median(bar), np.median(dur), 100 * np.median(bar / dur)))
    for name, col in (("loader: a tile's loads, first issue to arrival (sum over the tiles)", 4), ("loader: at the barriers", 5), ("wavefront 0: its gather loop", 6)):
        v = a[:, col].astype(np.float64) / 2400.0
        print("  %-72s median %.1f us (%.0f%% of the workgroup's time)" % (name, np.median(v), 100 * np.median(v / dur)))
    first = t0 < 10.0
    print("workgroups started in the first 10 us: %d, duration median %.1f; the others: %d, median %.1f" % (first.sum(), np.median(dur[first]), (~first).sum(), np.median(dur[~first])))
    edges = np.linspace(0, t1.max(), 21)
    act = [(np.minimum(t1, edges[i + 1]) - np.maximum(t0, edges[i])).clip(0).sum() / (edges[i + 1] - edges[i]) for i in range(20)]
    print("active workgroups per twentieth of the span:", " ".join("%d" % v for v in act))
    for x in range(8):
        s = xcc == x
        if s.any():
            print("  XCC %d: %d panels, first start %.1f, last end %.1f, distinct (se, sh, cu) %d, median duration %.1f" % (
                x, s.sum(), t0[s].min(), t1[s].max(), len(set(zip(se[s], sh[s], cu[s]))), np.median(dur[s])))
    order = np.argsort(t0)
    print("start times of the 520th .. 530th workgroup to start:", " ".join("%.1f" % v for v in t0[order][520:530]))


if __name__ == "__main__":
    main()
