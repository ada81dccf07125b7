"""k_lbfgs_pre geometry sweep at the headline vector length (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frx_import  # noqa
import fast_racing_amd as frx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 704
for geom in [(4, 3, 8, 4), (6, 2, 8, 4), (2, 6, 16, 4), (4, 4, 8, 4), (8, 2, 4, 4)]:
    try:
        err, us = frx.dv_selftest(n, B=32, m=128, iters=170, geom=geom)
        print("geom", geom, "err %.1e" % err, "us %.1f" % us)
    except Exception as ex:
        print("geom", geom, "failed", ex)
