"""PCIe-inclusive timing of the drop-in call (host NumPy arrays in, host PDFs out)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import bench
from frankenz_amd import BruteForce, PDFDict
N, M = 1000000, 100000
Y, Ye, Ym, X, Xe, Xm, z, ze = bench.make_problem(N, M, 1)
d = PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
bf = BruteForce(Y, Ye, Ym)
for rep in range(4):            # (rep 0 pays the page-locking of the 5.6 GB result block; later reps reuse it from the pool once `p` is dropped)
    t0 = time.perf_counter()
    p = bf.fit_predict(X, Xe, Xm, z, ze, label_dict=d, save_fits=False, verbose=False)
    dt = time.perf_counter() - t0
    print("host path: %d x %d in %.3f s = %.3g evals/s (%.3g PDFs/s), PDF bytes %.2f GB, rows sum to 1: %s" % (
        N, M, dt, N * M / dt, N / dt, p.nbytes / 1e9, bool(np.allclose(p[::997].sum(axis=1), 1.0, rtol=1e-9))))
    del p
