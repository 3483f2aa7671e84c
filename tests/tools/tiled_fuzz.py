import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tiled
import test_tiled_gpu as T
from gpu import Dev
dev = Dev()
g = tiled.geometry(dev.k)
bad = 0
for seed in range(1000, 1000 + int(sys.argv[1])):
    rng = np.random.default_rng(seed)
    m = int(rng.choice([1, 2, 5, 64, 65, 257, 900, 2500, 7000]))
    n = int(rng.choice([1, 2, 63, g["tw"] - 1, g["tw"], g["tw"] + 1, 2 * g["tw"] + 129, (1 << 18) - 1, (1 << 18) + 1, (1 << 19) + 4097]))
    lens = np.minimum(rng.integers(0, int(rng.choice([3, 12, 60])), m), n)
    if rng.random() < 0.4:
        lens[rng.integers(0, m)] = min(n, int(rng.choice([129, 700, 3000])))
    ai, aj, aa = T.random_csr(rng, m, n, lens, band=int(rng.choice([1, 50, 3000, n])), far_frac=float(rng.choice([0.0, 0.1, 0.5, 1.0])))
    smin = int(rng.choice([1, 4, 33, 500, 10 ** 9]))
    try:
        T.run_case(dev, ai, aj, aa, n, stage_min=smin, seed=seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "m", m, "n", n, "nnz", aj.size, "smin", smin, "FAILED:", repr(e)[:300], flush=True)
print("fuzz done:", int(sys.argv[1]), "cases,", bad, "failures", flush=True)
