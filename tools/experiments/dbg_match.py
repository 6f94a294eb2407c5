import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
ctx = pkg.Context(width=640, height=480)
rng = np.random.default_rng(1)
for n_from, n_to in [(2000, 2000), (513, 257), (1, 5), (2, 3), (300, 1), (32, 64), (33, 64), (64, 64)]:
    f = rng.integers(0, 256, (n_from, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (n_to, 32), dtype=np.uint8)
    got = ctx.match_knn2(f, t)
    ref = orc.match_knn2_raw(f, t)
    names = ["idx0", "idx1", "d0", "d1"]
    for nm, a, b in zip(names, got, ref):
        bad = np.nonzero(a != b)[0]
        print(n_from, n_to, nm, "mismatches", len(bad), [(int(i), int(a[i]), int(b[i])) for i in bad[:6]])
