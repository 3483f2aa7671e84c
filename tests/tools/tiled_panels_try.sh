# the column-tiled product over the number of row panels (MI355X_TILED_PANELS) and one / two workgroups per CU, IRR stand-in
R=${GRAFT_REPO_ROOT:-/root/repo}
export CFG4_CACHE=/tmp
for np in 255 512 765 1024; do for two in "" 1; do echo "== $np panels${two:+, two workgroups per CU}"; env MI355X_TILED_PANELS=$np ${two:+MI355X_TILED_TWO_PER_CU=1} timeout -k 10 300 python3 $R/tests/tools/tiled_probe.py irr 1024 2>&1 | grep "^tiled" | cut -c1-60,100-330; done; done
