#!/usr/bin/env python
"""
bench.py -- object-template likelihood evals/s (+ PDFs/s) of the fused
BruteForce.fit_predict path (bruteforce.py:505-631, save_fits=False) on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus 8 --steps 3 --warmup 1

One "step" = one pass of the hot path over one batch: N_obj objects x N_model
models x 5 bands -> N_obj PDFs on the 701-point redshift grid.  Default workload is
BASELINE.json's headline configuration (1e6 x 1e5 x 5, config index 2).  Inputs are
synthetic (SURVEY.md section 8d generator) and already resident in HBM when the timed
region starts; outputs stay in HBM.  Multi-GPU (north_star): the N_obj axis of the SAME
workload is sharded over the ranks (strong scaling: 1e6 objects in total), models / labels /
dictionary are replicated, and the step is the library's own multi-GPU call,
frankenz_amd.sharded.sharded_fit_predict: objects dealt out block-cyclically in --chunks
rounds, each round's PDF rows written by the kernel straight into the full (N, 701) device
array and all-gathered in place over xGMI (RCCL, async) while the next round is computed.
The gather is inside the timed region; compute time and the exposed (not hidden) gather time
are reported separately.  --scaling weak gives every rank --nobj objects, --no-gather leaves
the collective out.  bench.py holds no collective code of its own.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SDSS_SIGMA = np.array([0.873, 0.348, 0.418, 0.873, 3.476])   # SDSS ugriz 1-sigma depths
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector FP64 (public spec; = FP64 matrix peak)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec


def flops_per_eval(mode, B, fused=True):
    """algorithmic flops per object-model evaluation (add/mul/div/log/exp = 1, fma = 2): mode A
    8 per band + 10 for the chi2-distribution epilogue; mode B 4 per band (inter, shape) + 1 + 5 per band
    (residual form) + 10; + 8 for the fused softmax / threshold / KDE scatter (+ 4 for one statistics pass).
    B = 5: 58 / 64 fused, as SURVEY 8d states."""
    if mode not in ("A", "B", "Ai", "An", "Bn"):
        return None
    core = (9 * B + 11) if mode in ("B", "Bn") else (8 * B + 10)
    return core + (8 if fused else 4)


MODES = {"A": {}, "B": {"free_scale": True, "ignore_model_err": True},
         "Ai": {"ignore_model_err": True},
         "An": {"dim_prior": False}, "Bn": {"free_scale": True, "ignore_model_err": True, "dim_prior": False},
         # free scale WITH model errors: the fixed-point loop of pdf.py:196-223 (iterations are data
         # dependent, so SURVEY 8d asks for evals/s without a roofline fraction)
         "C": {"free_scale": True}}


def make_problem(n_obj, n_model, seed, B=5, noise=1.0):
    """lognormal model fluxes, SDSS-depth noise (B != 5, a side
    experiment of --nband: the five SDSS depths repeated / truncated; noise != 1, a side
    experiment of --noise-scale: broader likelihoods, so more models pass wt_thresh)."""
    rs = np.random.RandomState(seed)
    sig = np.resize(SDSS_SIGMA, B) * noise
    Y = rs.lognormal(mean=1.0, sigma=1.0, size=(n_model, B))
    Ye = np.tile(sig, (n_model, 1))
    Ym = np.ones((n_model, B))
    pick = rs.randint(0, n_model, size=n_obj)
    X = Y[pick] + sig * rs.standard_normal((n_obj, B))
    Xe = np.tile(sig, (n_obj, 1))
    Xm = np.ones((n_obj, B))
    z = rs.uniform(0.0, 6.0, n_model)
    ze = np.full(n_model, 0.05)
    return Y, Ye, Ym, X, Xe, Xm, z, ze


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(Y, Ye, Ym, X, Xe, Xm, z, ze, kw, budget_s):
    """The oracle (NumPy port of the reference loop) on a bounded sample of the SAME
    workload: as many objects as fit in ~budget_s seconds against the FULL model set,
    one process / one core like the reference."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import frankenz_oracle as fo
    kd = fo.KernelDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    yi, ysi = kd.fit(z, ze)
    n = 0
    t0 = time.perf_counter()
    while n < len(X):
        r = fo.logprob(X[n].copy(), Xe[n].copy(), Xm[n].copy(), Y, Ye, Ym, **kw)
        fo._pdf_from_lnprob(r[2], z, ze, kd, None, yi, ysi, {})
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n * len(Y) / dt, "unit": "evals/s", "pdfs_per_s": n / dt, "cores": 1,
            "kind": "port", "cpu": cpu_model(),
            "sample": "%d objects x %d models (full model set), %.1f s, oracle/frankenz_oracle.py "
                      "logprob+logsumexp+gauss_kde_dict loop" % (n, len(Y), dt)}


def _cpu_worker(job):
    """one worker of the all-cores CPU baseline: a contiguous block of objects."""
    (Y, Ye, Ym, X, Xe, Xm, z, ze, kw, budget_s) = job
    r = cpu_baseline(Y, Ye, Ym, X, Xe, Xm, z, ze, kw, budget_s)
    return r["pdfs_per_s"] * 1.0, int(r["sample"].split()[0])


def cpu_baseline_all(Y, Ye, Ym, X, Xe, Xm, z, ze, kw, budget_s):
    """the same loop on every host core (one process each, the objects split in blocks);
    spawned BEFORE this process touches the GPU."""
    import multiprocessing as mp
    ncpu = os.cpu_count() or 1                      # every host core (the count is printed)
    per = max(1, min(len(X) // ncpu, 4096))
    jobs = [(Y, Ye, Ym, X[i * per:(i + 1) * per], Xe[i * per:(i + 1) * per], Xm[i * per:(i + 1) * per], z, ze, kw,
             budget_s) for i in range(ncpu)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(ncpu) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    nobj = sum(r[1] for r in res)
    rate = sum(r[0] for r in res)                    # objects/s, workers timed individually (excludes spawn cost)
    return {"value": rate * len(Y), "unit": "evals/s", "pdfs_per_s": rate, "cores": ncpu, "kind": "port", "cpu": cpu_model(),
            "sample": "%d objects x %d models over %d processes, ~%.0f s each (wall %.1f s incl. start-up)"
                      % (nobj, len(Y), ncpu, budget_s, wall)}


def _sub(fn):
    """a sub-record of the default line: the measurement, or what went wrong (the headline never depends on it)"""
    try:
        return fn()
    except Exception as e:           # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def sub_planes_predict(eng, dev, torch, pd, steps=3):
    """BASELINE configs[1]: 1e5 objects x 1e4 models x 5 bands, materialising BruteForce.fit (lnlike + chi2 planes: HBM-write bound,
    16 B per eval) and BruteForce.predict from the stored plane (8 B per eval read once)."""
    from frankenz_amd.engine import kde_opts, like_opts
    N, M = 100000, 10000
    Y, Ye, Ym, X, Xe, Xm, z, ze = make_problem(N, M, 20260101, 5, 1.0)
    eng.upload_models(Y, Ye, Ym)
    eng.set_labels(z, ze, label_dict=pd)
    dX, dXe, dXm = (torch.from_numpy(a).to(dev) for a in (X, Xe, Xm))
    d_lnl = torch.empty((N, M), dtype=torch.float64, device=dev)
    d_chi2 = torch.empty((N, M), dtype=torch.float64, device=dev)
    d_pdf = torch.empty((N, pd.Ngrid), dtype=torch.float64, device=dev)
    d_lm = torch.empty(N, dtype=torch.float64, device=dev); d_le = torch.empty(N, dtype=torch.float64, device=dev)
    opts, ko = like_opts({}), kde_opts({"wt_thresh": 1e-3})
    out = {}
    eng.fit(dX, dXe, dXm, opts, d_lnl, d_chi2, n=N); eng.sync(); eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.fit(dX, dXe, dXm, opts, d_lnl, d_chi2, n=N)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    tm = eng.timing()
    ms = tm["ms_planes"] / max(tm["n_planes"], 1)
    gbs = N * M * 16 / (max(tm["n_planes"], 1) / steps) / (ms * 1e-3) / 1e9
    out["planes"] = {"workload": "BASELINE configs[1]: BruteForce.fit, %d x %d x 5, lnlike + chi2 planes" % (N, M), "value": N * M / dt,
                     "unit": "evals/s", "ms_per_step": dt * 1e3,
                     "roofline": {"bound": "hbm", "kernel": "k_planes", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": gbs / HBM_PEAK_GBS, "bytes_per_eval": 16, "avg_launch_ms": ms}}
    eng.predict_logwt(d_lnl, ko, d_pdf, d_lm, d_le, n=N); eng.sync(); eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.predict_logwt(d_lnl, ko, d_pdf, d_lm, d_le, n=N)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    tm = eng.timing()
    ms = (tm["ms_stats"] + tm["ms_kde"] + tm["ms_fused"]) / steps
    gbs = N * M * 8 / (ms * 1e-3) / 1e9
    out["predict"] = {"workload": "BruteForce.predict(logwt=fit_lnprob): %d x %d plane -> %d PDFs" % (N, M, N), "value": N / dt,
                      "unit": "PDFs/s", "ms_per_step": dt * 1e3,
                      "roofline": {"bound": "hbm", "kernel": eng.last_form(), "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gbs / HBM_PEAK_GBS, "bytes_per_eval": 8, "note": "algorithmic: the plane read once"}}
    return out


def sub_modec(eng, dev, torch, pd, steps=2):
    """free scale WITH model errors (pdf.py:196-223): the fixed point's iteration count is data dependent, SURVEY 8d asks for evals/s
    and the iterations, no roofline fraction"""
    from frankenz_amd.engine import kde_opts, like_opts
    N, M = 20000, 10000
    Y, Ye, Ym, X, Xe, Xm, z, ze = make_problem(N, M, 20260101, 5, 1.0)
    Ye = Ye * np.random.RandomState(77).uniform(0.5, 1.5, size=Ye.shape)
    eng.upload_models(Y, Ye, Ym)
    eng.set_labels(z, ze, label_dict=pd)
    dX, dXe, dXm = (torch.from_numpy(a).to(dev) for a in (X, Xe, Xm))
    d_pdf = torch.empty((N, pd.Ngrid), dtype=torch.float64, device=dev)
    d_lm = torch.empty(N, dtype=torch.float64, device=dev); d_le = torch.empty(N, dtype=torch.float64, device=dev)
    opts, ko = like_opts(MODES["C"]), kde_opts({"wt_thresh": 1e-3})
    eng.fit_predict_prior(dX, dXe, dXm, opts, ko, None, d_pdf, d_lm, d_le, n=N); eng.sync(); eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.fit_predict_prior(dX, dXe, dXm, opts, ko, None, d_pdf, d_lm, d_le, n=N)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    tm = eng.timing()
    info = eng.modec_info()
    return {"workload": "mode C (free scale with model errors): %d x %d x 5, fused fit_predict" % (N, M), "value": N * M / dt, "unit": "evals/s",
            "ms_per_step": dt * 1e3, "iterations_of_the_slowest_object": tm["n_modec"] / steps - 2, "modec_info": [int(v) for v in info],
            "note": "evals/s counts each (object, model) pair once, whatever its iteration count"}


def sub_nz_stack(eng, dev, torch, d_pdf, N, G, steps=3):
    """BASELINE configs[4], second half: one sweep of the hierarchical n(z) sampler over the PDFs of the step -- the per-object
    categorical draw from pdf_i * nz (samplers.py:498-499, 519-520) and the ln-likelihood of the stack (samplers.py:60-80).  Both read
    the (N, G) PDF stack once: HBM-bound, 8 B per PDF entry each."""
    nz = torch.full((G,), 1.0 / G, dtype=torch.float64, device=dev)
    u = torch.rand(N, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    bins = torch.empty(N, dtype=torch.int64, device=dev); counts = torch.zeros(G, dtype=torch.int64, device=dev)
    overlap = torch.empty(N, dtype=torch.float64, device=dev)
    eng.nz_assign(d_pdf, nz, u, bins, counts, n=N); ll = eng.overlap_nz(d_pdf, nz, None, 0.0, overlap, n=N); eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        counts.zero_()
        eng.nz_assign(d_pdf, nz, u, bins, counts, n=N)
        ll = eng.overlap_nz(d_pdf, nz, None, 0.0, overlap, n=N)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    gbs = 2 * N * G * 8 / dt / 1e9
    return {"workload": "BASELINE configs[4], the n(z) half: one Gibbs sweep (categorical draw per object + ln-likelihood of the stack) over "
                        "%d PDFs x %d grid points" % (N, G), "value": N / dt, "unit": "objects/s per sweep", "ms_per_step": dt * 1e3,
            "counts_sum_to_N": bool(int(counts.sum().item()) == N), "loglike_finite": bool(np.isfinite(ll)),
            "roofline": {"bound": "hbm", "kernel": "k_nz_assign + k_overlap", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "bytes_per_pdf_entry": 16, "note": "algorithmic: the PDF stack read once by each kernel"}}


def sub_knn(dev, torch, pd, local, steps=2):
    """BASELINE configs[3] slice: NearestNeighbors.fit_predict(save_fits=False), K = 25 Monte-Carlo feature sets, k = 20, 1e5 objects x
    1e5 models, device resident.  The roofline figure is the matrix pipe's: the exact search multiplies 64-model tiles of fp32
    features (padded to 8) against 16-query tiles, 2 x 8 flops per (query, model) it LOOKS at -- it skips tiles its k-d boxes exclude,
    so `tiles_frac_of_exhaustive` < 1 is pruning, not idleness; `achieved` counts the exhaustive scan's flops over the search time."""
    from frankenz_amd import NearestNeighbors
    from frankenz_amd.engine import get_engine
    N, M, Kt, kk = 100000, 100000, 25, 20
    Y, Ye, Ym, X, Xe, Xm, z, ze = make_problem(N, M, 20260101, 5, 1.0)
    fk = dict(skynoise=SDSS_SIGMA, zeropoints=10 ** (0.4 * 23.9))
    nn = NearestNeighbors(Y, Ye, Ym, K=Kt, feature_map="luptitude", fmap_kwargs=fk, rstate=np.random.RandomState(1), verbose=False)
    nn._device = local
    q = nn._query_features(X, Xe, np.random.RandomState(2))
    dQ = torch.from_numpy(q).to(dev)
    dX, dXe, dXm = (torch.from_numpy(a).to(dev) for a in (X, Xe, Xm))
    prep = nn.prepare_fit_predict(z, ze, label_dict=pd, kde_kwargs={"wt_thresh": 1e-3}, lprob_kwargs={}, k=kk)
    d_pdf = torch.empty((N, pd.Ngrid), dtype=torch.float64, device=dev)
    d_lm = torch.empty(N, dtype=torch.float64, device=dev); d_le = torch.empty(N, dtype=torch.float64, device=dev)
    eng = get_engine(local)
    prep.run(dX, dXe, dXm, out=(d_pdf, d_lm, d_le), query_features=dQ); eng.sync(); eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        prep.run(dX, dXe, dXm, out=(d_pdf, d_lm, d_le), query_features=dQ)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    tm = eng.timing()
    ms_knn = tm["ms_knn"] / steps
    flops = 2.0 * 8 * Kt * N * M                      # exhaustive scan, fp32 features padded to 8
    s = d_pdf[:4096].sum(dim=1)
    return {"workload": "BASELINE configs[3] slice: NearestNeighbors.fit_predict(save_fits=False), %d objects x %d models, K=%d k=%d" % (N, M, Kt, kk),
            "value": N / dt, "unit": "objects/s", "ms_per_step": dt * 1e3, "search_evals_per_s": Kt * N * M / dt, "kernel_ms_per_step": ms_knn,
            "pdfs_normalised": bool(float((s - 1).abs().max().item()) < 1e-9),
            "roofline": {"bound": "mfma", "kernel": "k_knn_mfma + k_knn_subset", "achieved": flops / (ms_knn * 1e-3) / 1e12, "peak": 157.3,
                         "unit": "TFLOP/s (fp32 matrix, exhaustive-scan equivalent)", "frac": flops / (ms_knn * 1e-3) / 1e12 / 157.3,
                         "note": "the exact k-d-ordered search multiplies only the tiles its boxes cannot exclude (~7 % of them, "
                                 "profiles/README.md): a fraction above the tiles' share means the pipe did useful work faster than an exhaustive scan could"}}


def sub_host_path(local, pd):
    """the drop-in call as the reference's users make it: NumPy in, NumPy out, BruteForce.fit_predict(save_fits=False) at BASELINE
    configs[2] (PCIe inclusive: 120 MB of objects in, 5.6 GB of PDFs out through page-locked blocks)"""
    from frankenz_amd import BruteForce
    N, M = 1000000, 100000
    Y, Ye, Ym, X, Xe, Xm, z, ze = make_problem(N, M, 20260101, 5, 1.0)
    bf = BruteForce(Y, Ye, Ym, device=local)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        p = bf.fit_predict(X, Xe, Xm, z, ze, label_dict=pd, save_fits=False, verbose=False)
        ts.append(time.perf_counter() - t0)
        ok = bool(abs(float(p[:1024].sum(axis=1).max()) - 1) < 1e-9)
        del p
    return {"workload": "BruteForce.fit_predict(save_fits=False), NumPy -> NumPy, %d x %d x 5" % (N, M), "first_call_s": ts[0],
            "steady_call_s": min(ts[1:]), "value": N * M / min(ts[1:]), "unit": "evals/s (PCIe inclusive)", "pdfs_normalised": ok,
            "note": "the first call page-locks the 5.6 GB result block; later calls reuse it from the library's pool"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nobj", type=int, default=1000000, help="objects per GPU per step")
    ap.add_argument("--nmodel", type=int, default=100000)
    ap.add_argument("--mode", choices=sorted(MODES), default="A")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): --nobj objects IN TOTAL, sharded over the ranks.  weak: --nobj objects per rank")
    ap.add_argument("--no-gather", action="store_true", help="leave the RCCL all-gather of the PDF shards out of the step")
    ap.add_argument("--gather", action="store_true", help="(default for N > 1; kept for compatibility)")
    ap.add_argument("--chunks", type=int, default=4, help="rounds per step of the overlapped all-gather (N > 1)")
    ap.add_argument("--workload", choices=["fit_predict", "fit", "predict", "knn", "summarize"], default="fit_predict",
                    help="fit_predict: headline fused path (default). fit: materialising BruteForce.fit "
                         "planes (BASELINE configs[1] when --nobj 100000 --nmodel 10000). predict: "
                         "BruteForce.predict from the stored (N,M) ln-prob plane of a fit. knn: KMCkNN "
                         "search + subset PDFs (configs[3])")
    ap.add_argument("--mask-frac", type=float, default=0.0,
                    help="fraction of object bands flagged unobserved (exercises the masked kernels)")
    ap.add_argument("--model-mask-frac", type=float, default=0.0,
                    help="fraction of MODEL bands flagged missing (models_mask; N_dim then differs from pair to pair: the "
                         "segmented form of the one-pass kernel)")
    ap.add_argument("--prior", type=int, default=0,
                    help="P > 0: add an ln-prior table of P rows (one row index per object) to the "
                         "fused path (the device form of a custom lprob_func, SURVEY 8f-1)")
    ap.add_argument("--model-err", choices=["const", "varying"], default="const",
                    help="const: ye = sigma_b for every model (SURVEY 8d configs; the library folds band-constant model "
                         "errors into the object variances).  varying: ye = sigma_b * U(0.5, 1.5) per model and band "
                         "(the general mode A kernels)")
    ap.add_argument("--nband", type=int, default=5, help="band count (headline: 5, the SDSS configuration)")
    ap.add_argument("--wt-thresh", type=float, default=1e-3, help="kde_kwargs wt_thresh (reference default 1e-3)")
    ap.add_argument("--noise-scale", type=float, default=1.0,
                    help="multiply the SDSS depths (side experiment: low-S/N objects keep far more models above wt_thresh)")
    ap.add_argument("--kde", choices=["dict", "grid"], default="dict",
                    help="dict: gauss_kde_dict on the 701-point grid with the 500-kernel dictionary (the reference demos; "
                         "default).  grid: the direct gauss_kde on the same grid (pdf.py:444-526)")
    ap.add_argument("--label-err", choices=["const", "varying"], default="const",
                    help="const: every label carries sigma_z = 0.05 (one dictionary kernel: the histogram + one "
                         "convolution path; every demo of the reference).  varying: sigma_z = U(0.01, 0.1) per model "
                         "(many dictionary kernels: each selected model's window is added)")
    ap.add_argument("--exact", action="store_true", help="lprob_kwargs exact_evidence=True: every weight and the ln-evidence in fp64")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    kw_cpu = MODES[args.mode]                               # what the oracle's logprob understands
    kw = dict(kw_cpu, exact_evidence=True) if args.exact else dict(kw_cpu)
    M = args.nmodel
    strong = args.scaling == "strong" and world > 1
    sys.path.insert(0, ROOT)
    from frankenz_amd.sharded import shard_slice
    do_gather = world > 1 and not args.no_gather and args.workload in ("fit_predict", "knn")
    if world > 1 and (strong or do_gather):
        # the SAME problem on every rank count: one seed; every rank holds the (small) full object arrays and the
        # library deals the objects out (sharded_fit_predict).  Weak scaling: --nobj objects per rank.
        N_total = args.nobj if strong else args.nobj * world
        Y, Ye, Ym, X, Xe, Xm, z, ze = make_problem(N_total, M, 20260101, args.nband, args.noise_scale)
        if do_gather:
            N = N_total                                 # the rank's tensors hold every object; it computes its share
        else:
            sl = shard_slice(N_total, world, rank)
            X, Xe, Xm = X[sl], Xe[sl], Xm[sl]
            N = sl.stop - sl.start
        n_pad = N
    else:
        N = args.nobj
        N_total = N * world
        n_pad = N
        Y, Ye, Ym, X, Xe, Xm, z, ze = make_problem(N, M, 20260101 + rank, args.nband, args.noise_scale)
    if args.model_err == "varying":
        Ye = Ye * np.random.RandomState(77).uniform(0.5, 1.5, size=Ye.shape)
    if args.label_err == "varying":
        ze = np.random.RandomState(78).uniform(0.01, 0.1, size=ze.shape)
    # CPU baselines first: worker processes are spawned before this process initialises the GPU
    cpu1 = cpuall = None
    if world == 1 and not args.no_cpu and args.workload == "fit_predict" and not args.prior:
        cpu1 = cpu_baseline(Y, Ye, Ym, X, Xe, Xm, z, ze, kw_cpu, args.cpu_seconds)
        if (os.cpu_count() or 1) > 1:
            cpuall = cpu_baseline_all(Y, Ye, Ym, X, Xe, Xm, z, ze, kw_cpu, min(args.cpu_seconds, 10.0))
    import torch
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # FZ_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing tests on a 1-GPU box)
    backend = os.environ.get("FZ_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from frankenz_amd import PDFDict
    from frankenz_amd.engine import get_engine, kde_opts, like_opts

    if args.mask_frac > 0:
        Xm[np.random.RandomState(5).rand(*Xm.shape) < args.mask_frac] = 0.0
    if args.model_mask_frac > 0:
        Ym[np.random.RandomState(6).rand(*Ym.shape) < args.model_mask_frac] = 0.0
    pd = PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    G = pd.Ngrid

    eng = get_engine(local)                                 # the process-wide engine of this GPU (the drop-in classes use the same one)
    eng.upload_models(Y, Ye, Ym)
    if args.kde == "grid":
        eng.set_labels(z, ze, label_grid=np.ascontiguousarray(pd.grid, dtype=np.float64))
    else:
        eng.set_labels(z, ze, label_dict=pd)
    dev = torch.device("cuda", local)
    dX, dXe, dXm = (torch.from_numpy(a).to(dev) for a in (X, Xe, Xm))
    d_pdf = None if do_gather else torch.zeros((n_pad, G), dtype=torch.float64, device=dev)   # (N > 1: the sharded call owns the result)
    d_lm = torch.empty(N, dtype=torch.float64, device=dev)
    d_le = torch.empty(N, dtype=torch.float64, device=dev)
    gathered = None
    bf = None
    if do_gather and args.workload == "fit_predict":
        # the library's multi-GPU call (drop-in class + sharded driver); the engine above is the same process-wide one
        from frankenz_amd import BruteForce, sharded
        os.environ["FRANKENZ_DEVICE"] = str(local)
        bf = BruteForce(Y, Ye, Ym, device=local)
        grid_np = np.ascontiguousarray(pd.grid, dtype=np.float64)
        bf_prep = bf.prepare_fit_predict(z, ze, label_dict=pd if args.kde == "dict" else None,
                                         label_grid=None if args.kde == "dict" else grid_np,
                                         kde_kwargs={"wt_thresh": args.wt_thresh}, lprob_kwargs=kw)     # uploads once, outside the timed steps (inputs are resident by contract)
        d_pdf = None                                             # the sharded call owns the (N, G) result
    opts, ko = like_opts(kw), kde_opts({"wt_thresh": args.wt_thresh})
    prior = None
    if args.prior > 0:
        gen = torch.Generator(device=dev); gen.manual_seed(11 + rank)
        d_tab = torch.log_softmax(torch.randn((args.prior, M), dtype=torch.float64, device=dev, generator=gen), dim=1)
        d_rows = torch.randint(0, args.prior, (N,), dtype=torch.int64, device=dev, generator=gen)
        prior = (d_tab, args.prior, d_rows)
    if args.workload in ("fit", "predict"):
        d_lnl = torch.empty((N, M), dtype=torch.float64, device=dev)
        d_chi2 = torch.empty((N, M), dtype=torch.float64, device=dev) if args.workload == "fit" else None
    if args.workload == "predict":
        eng.fit(dX, dXe, dXm, opts, d_lnl, None, n=N)          # the stored plane predict() reads (bruteforce.py:263-264)
    if args.workload == "knn":
        from frankenz_amd import NearestNeighbors
        Kt, kk = 25, 20
        fk = dict(skynoise=SDSS_SIGMA, zeropoints=10 ** (0.4 * 23.9))
        nn = NearestNeighbors(Y, Ye, Ym, K=Kt, feature_map="luptitude", fmap_kwargs=fk,
                              rstate=np.random.RandomState(1), verbose=False)
        nn._device = local
        # knn.py:830-832 for every object, once (host RNG: the stream is NumPy's); the steps then run device-resident through the
        # drop-in class: query features, neighbour table and PDFs never leave HBM
        q = nn._query_features(X, Xe, np.random.RandomState(2))
        dQ = torch.from_numpy(q).to(dev)
        nn_prep = nn.prepare_fit_predict(z, ze, label_dict=pd, kde_kwargs={"wt_thresh": args.wt_thresh}, lprob_kwargs=kw, k=kk)
        if do_gather:
            from frankenz_amd import sharded
            d_pdf = None

    if args.workload == "summarize":
        # pdf.pdfs_summarize on the PDFs of one fused pass (device-resident stack)
        eng.fit_predict(dX, dXe, dXm, opts, ko, d_pdf, d_lm, d_le, n=N)
        gt = np.asarray(pd.grid, dtype=np.float64)
        kg = (gt[:, None] - gt[None, :]) / ((1. + gt[:, None]) * 0.15)          # host-side (G,G) table, pdf.py:1003-1023
        d_loss = torch.from_numpy(np.ascontiguousarray(1.0 - 1. / (1. + np.square(kg)))).to(dev)
        d_u = torch.rand(N, dtype=torch.float64, device=dev)
        d_stats = torch.empty((21, N), dtype=torch.float64, device=dev)
        grid_np = np.ascontiguousarray(pd.grid, dtype=np.float64)

    def step():
        if args.workload == "summarize":
            eng.pdfs_summarize(d_pdf, grid_np, True, d_u, d_loss, None, 0.03, d_stats, n=N)
            return
        if args.workload == "fit":
            eng.fit(dX, dXe, dXm, opts, d_lnl, d_chi2, n=N)
            return
        if args.workload == "predict":
            eng.predict_logwt(d_lnl, ko, d_pdf, d_lm, d_le, n=N)
            return
        if args.workload == "knn":
            if do_gather:
                # N > 1: ONE shared problem, objects dealt out block-cyclically, PDF rows gathered in place behind the next round's search
                res = sharded.sharded_fit_predict(nn, dX, dXe, dXm, z, ze, gather='pdfs', chunks=args.chunks, label_dict=pd,
                                                  lprob_kwargs=kw, kde_kwargs={"wt_thresh": args.wt_thresh}, save_fits=False,
                                                  prepared=nn_prep, query_features=dQ, k=kk)
                last[0] = res
                st = sharded.last_stats
                split[0] += st["ms_compute"] * 1e-3; split[1] += st["ms_gather_exposed"] * 1e-3
                return
            nn_prep.run(dX, dXe, dXm, out=(d_pdf, d_lm, d_le), query_features=dQ)      # NearestNeighbors.fit_predict(save_fits=False), device-resident
            return
        if bf is not None:
            # N > 1: shard compute + the overlapped RCCL all-gather of the PDF rows, as one library call
            res = sharded.sharded_fit_predict(bf, dX, dXe, dXm, z, ze, gather='pdfs', chunks=args.chunks, label_dict=pd if args.kde == "dict" else None,
                                              label_grid=None if args.kde == "dict" else grid_np,
                                              lprob_kwargs=kw, kde_kwargs={"wt_thresh": args.wt_thresh}, save_fits=False, prepared=bf_prep)
            last[0] = res
            st = sharded.last_stats
            split[0] += st["ms_compute"] * 1e-3; split[1] += st["ms_gather_exposed"] * 1e-3
            return
        t0 = time.perf_counter()
        eng.fit_predict_prior(dX, dXe, dXm, opts, ko, prior, d_pdf, d_lm, d_le, n=N)       # returns when the PDFs are in HBM
        split[0] += time.perf_counter() - t0

    split = [0.0, 0.0]              # seconds in compute / in the exposed part of the all-gather over the timed steps
    last = [None]                   # N > 1: the last step's gathered (pdfs, (lmap, levid))

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()        # every torch-side input (prior tables, uniforms) is complete before the library reads it
    for _ in range(args.warmup):
        step()
    fence()
    eng.timing_reset()
    split[0] = split[1] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tm = eng.timing()
    if dist is not None:
        t = torch.tensor([dt, split[0], split[1]], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, split[0], split[1] = (float(v) for v in t.tolist())

    # sanity: PDFs are normalised
    ok = True
    if last[0] is not None:
        d_pdf = last[0][0]
        N_chk = int(d_pdf.shape[0])
        assert N_chk == N_total
        s = d_pdf[:: max(1, N_chk // 4096)].sum(dim=1)           # rows of every rank's share
        ok = bool(torch.isfinite(s).all().item()) and float((s - 1).abs().max().item()) < 1e-9
    elif args.workload != "fit":
        s = d_pdf[: min(N, 4096)].sum(dim=1)
        fin = torch.isfinite(s)
        # masked models against masked objects: a pair without a common band makes the reference's own row nan (gammaln(0), pdf.py:92);
        # those rows are counted, every other one must be normalised
        undefined = int((~fin).sum().item())
        allow = (args.mask_frac > 0 and args.model_mask_frac > 0)
        ok = bool((fin.all().item() or (allow and undefined < 0.05 * len(s)))) and float((s[fin] - 1).abs().max().item()) < 1e-9
    if args.workload != "fit_predict" and rank == 0:
        # secondary workloads: their own JSON line (not the driver's headline contract)
        if args.workload == "summarize":
            flops = 2.0 * N * G * G * args.steps
            print(json.dumps({"metric": "pdfs_summarize objects/sec (risk GEMM N x G x G fp64 MFMA + per-object statistics)",
                              "value": N_total * args.steps / dt, "unit": "objects/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": "pdf.pdfs_summarize: %d PDFs x %d grid points, lorentz kernel" % (N, G)},
                              "roofline": {"bound": "mfma", "kernel": "k_gemm_f64", "achieved": flops / dt / 1e12,
                                           "peak": 78.6, "unit": "TFLOP/s", "frac": flops / dt / 1e12 / 78.6,
                                           "note": "whole call (normalise + GEMM + statistics) over the GEMM's 2 N G^2 flops"}}))
        elif args.workload == "fit":
            ms = tm["ms_planes"] / max(tm["n_planes"], 1)
            per_launch = N * M / (max(tm["n_planes"], 1) / args.steps)
            gbs = per_launch * 16 / (ms * 1e-3) / 1e9
            print(json.dumps({"metric": "object-template likelihood evals/sec (materialising fit, lnlike+chi2 planes)",
                              "value": N_total * M * args.steps / dt, "unit": "evals/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                              "dtype": "f64", "data": "synthetic",
                              "config": {"workload": "BruteForce.fit: %d x %d x 5, mode %s, 2 fp64 planes" % (N, M, args.mode)},
                              "roofline": {"bound": "hbm", "kernel": "k_planes", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                           "bytes_per_eval": 16, "avg_launch_ms": ms}}))
        elif args.workload == "predict":
            ms = (tm["ms_stats"] + tm["ms_kde"] + tm["ms_fused"]) / args.steps
            gbs = N * M * 8 / (ms * 1e-3) / 1e9
            print(json.dumps({"metric": "BruteForce.predict PDFs/sec from a stored (N,M) ln-prob plane",
                              "value": N_total * args.steps / dt, "unit": "PDFs/s",
                              "evals_per_s": N_total * M * args.steps / dt, "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                              "pdfs_normalised": ok, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": "BruteForce.predict(logwt=fit_lnprob): %d x %d plane -> %d PDFs" % (N, M, N)},
                              "kernel_ms_per_step": {k: tm["ms_" + k] / args.steps for k in ("fused", "stats", "kde", "other")},
                              "roofline": {"bound": "hbm", "kernel": eng.last_form(),
                                           "achieved": gbs, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                           "bytes_per_eval": 8, "note": "algorithmic: the plane read once"}}))
        else:
            sprof = None      # (counter profiles of the search live in profiles/, stamped with the build they were taken on: not repeated here)
            print(json.dumps({"metric": "KMCkNN objects/sec (K=25 exact top-20 searches + subset PDFs)",
                              "value": N_total * args.steps / dt, "unit": "objects/s",
                              "search_evals_per_s": 25.0 * N_total * M * args.steps / dt, "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                              "kernel_ms_per_step": tm["ms_knn"] / args.steps, "pdfs_normalised": ok, "search_profile": sprof,
                              "dtype": "f64 (fp32 MFMA screen of the search, every admitted distance re-checked in fp64)", "data": "synthetic",
                              "scaling": "strong" if (strong or world == 1) else "weak",
                              "ms_compute": (split[0] / args.steps * 1e3) if do_gather else None,
                              "ms_gather_exposed": (split[1] / args.steps * 1e3) if do_gather else None,
                              "gather": ({"library_call": "frankenz_amd.sharded.sharded_fit_predict (NearestNeighbors, block-cyclic rounds, in-place all-gather)",
                                          "rounds": sharded.last_stats.get("chunks"), "bytes_total": N_total * G * 8,
                                          "ms_fence": sharded.last_stats.get("ms_fence")} if do_gather else None),
                              "config": {"workload": "BASELINE configs[3] slice: NearestNeighbors.fit_predict(save_fits=False, device-resident): %d objects x %d models, "
                                                     "K=25 k=20%s" % (N_total, M, (" sharded over %d GPUs (object axis)" % world) if world > 1 else "")}}))

    if rank == 0 and args.workload == "fit_predict":
        evals = float(N_total) * M * args.steps
        value = evals / dt
        n_local = (N_total / world) if do_gather else N         # objects this rank's kernels processed per step
        # dominant kernel, HIP events on the library's own stream (per launch)
        fam = max(("fused", "stats", "kde", "modec"), key=lambda k: tm["ms_" + k])
        ms_launch = tm["ms_" + fam] / max(tm["n_" + fam], 1)
        launches_per_step = max(tm["n_" + fam], 1) / args.steps
        evals_per_launch = n_local * M / launches_per_step
        flops_eval = flops_per_eval(args.mode, args.nband, fam == "fused") if fam != "modec" else None
        ach = evals_per_launch * flops_eval / (ms_launch * 1e-3) / 1e12 if flops_eval else None
        form = eng.last_form() if fam == "fused" else {"stats": "k_stats + k_kde", "kde": "k_stats + k_kde", "modec": "k_modec_step"}[fam]
        # HBM traffic: PMC passes of THIS build at THIS launch shape, kept in profiles/pmc_latest.json by tools/pmc.sh
        # (rocprofv3 cannot run inside the bench); only a matching (kernel form, objects per launch) entry is reported
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(prof):
            try:
                from frankenz_amd._lib import source_id
                sid = source_id()
                for ent in json.load(open(prof)).get("entries", []):
                    if ent.get("source_id") != sid:           # counters of another build: not reported
                        continue
                    if ent.get("form") == form and ent.get("mode") == args.mode and ent.get("model_err") == args.model_err \
                            and ent.get("n_band", 5) == args.nband and abs(ent.get("evals_per_launch", 0) - evals_per_launch) <= 0.01 * evals_per_launch:
                        traffic = ent["hbm_bytes_per_launch"]; traffic_src = ent.get("source")
            except Exception:
                traffic = None
        is_cfg2 = (N_total == 1000000 and M == 100000 and args.nband == 5)
        what = ("BASELINE configs[2]: " if is_cfg2 else "") + "%d objects x %d models x %d bands%s" % (
            N_total, M, args.nband, (" sharded over %d GPUs (object axis, block-cyclic rounds)" % world) if world > 1 else "")
        kde_txt = ("gauss_kde_dict on the 701-pt grid (500-kernel dictionary" + (", per-model label errors" if args.label_err == "varying" else "") + ")") \
            if args.kde == "dict" else "direct gauss_kde on the 701-pt grid"
        exact = bool(kw.get("exact_evidence")) or bool(os.environ.get("FZ_EXACT_EVIDENCE"))
        out = {
            "metric": "object-template likelihood evals/sec (fused fit_predict -> PDFs)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None,
            "dtype": ("f64 throughout (chi2, ln-likes, every weight, ln-evidence, PDFs; an fp32 estimate of the weight only CLASSIFIES which "
                      "pairs can matter to an fp64 sum, DESIGN.md 3.1)" if exact or form.startswith("k_hist") else
                      "f64 (chi2, ln-likes, PDFs and the weight of every model within wt_thresh of the best); the sum of the "
                      "remaining sub-threshold weights in the ln-evidence runs in fp32 on this kernel form (k_fused's weight-space "
                      "body, DESIGN.md 3.1b; lprob_kwargs={'exact_evidence': True} gives the all-fp64 body)"),
            "data": "synthetic",
            "config": {"workload": "%s, BruteForce.fit_predict(save_fits=False), likelihood mode %s%s, %s" % (
                           what, args.mode, " (per-model errors)" if args.model_err == "varying" else "", kde_txt),
                       "kernel_form": form,
                       "n_obj_total": N_total, "n_obj_per_gpu": n_local, "n_model": M, "n_band": args.nband, "mode": args.mode,
                       "lprob_kwargs": kw, "gather_pdfs": bool(do_gather),
                       "mask_frac": args.mask_frac, "model_mask_frac": args.model_mask_frac, "prior_rows": args.prior, "model_err": args.model_err,
                       "noise_scale": args.noise_scale, "kde": args.kde, "label_err": args.label_err},
            "note": ("band-constant model errors (the SURVEY 8d configuration): xe^2 + ye^2 is formed once per object "
                     "and mode A runs on the mode-Ai kernels; roofline_general (--model-err varying) times the general mode A kernels"
                     if (args.model_err == "const" and args.mode in ("A", "An")) else None),
            "pdfs_per_s": float(N_total) * args.steps / dt,
            "ms_compute": split[0] / args.steps * 1e3,
            "ms_gather_exposed": (split[1] / args.steps * 1e3) if do_gather else None,
            "gather": ({"collective": ("in-place all_gather_into_tensor per round (RCCL, async_op, overlapped with the next round's kernel)"
                                       if backend == "nccl" else "all_gather per round (gloo through host memory: plumbing only)"),
                        "library_call": "frankenz_amd.sharded.sharded_fit_predict", "rounds": sharded.last_stats.get("chunks"),
                        "bytes_total": N_total * G * 8, "bytes_received_per_rank": N_total * G * 8 * (world - 1) / world,
                        "ms_total_per_step": sharded.last_stats.get("ms_total"),
                        "busbw_GBs_if_fully_exposed": (N_total * G * 8 * (world - 1) / world / (split[1] / args.steps) / 1e9) if split[1] > 0 else None}
                       if do_gather else None),
            "pdfs_normalised": ok,
            "kernel_ms_per_step": {k: tm["ms_" + k] / args.steps for k in
                                   ("fused", "stats", "kde", "planes", "modec", "other")},
            "roofline": {"bound": "valu", "pipe": "fp64 vector ALU (its rate equals the dense fp64 MFMA peak on gfx950; no MFMA in this kernel: "
                                                   "K = 5 contractions, DESIGN.md 3.6)",
                         "kernel": form, "achieved": ach,
                         "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP64_VALU_PEAK_TFLOPS if ach else None, "traffic": traffic,
                         "traffic_source": traffic_src or "no PMC entry of THIS build (source stamp) for this kernel form and launch shape in profiles/pmc_latest.json",
                         "flops_per_eval": flops_eval,
                         "avg_launch_ms": ms_launch, "evals_per_launch": evals_per_launch,
                         "modec_iterations_per_step": (tm["n_modec"] / args.steps - 2) if fam == "modec" else None},   # minus the two timed scopes (iteration driver, final pass)
        }
        if world == 1 and args.mode == "A" and args.model_err == "const" and not args.prior and args.mask_frac == 0 and args.model_mask_frac == 0 \
                and args.kde == "dict" and args.label_err == "const" and args.noise_scale == 1.0 and not exact \
                and not os.environ.get("FZ_BENCH_NO_EXTRA"):
            # the headline configuration has band-constant model errors (the easy case of mode A): the same workload (a) on the GENERAL
            # mode A kernels (per-model errors) and (b) with the free scale (mode B), two steps each, so that the driver's record holds
            # the cases real data run on.  (Every one of them is fp64 throughout since round 4: there is no separate all-fp64 line.)
            def extra(Ye2, kw2, mode2, Ym2=None, dXm2=None, ze2=None):
                eng.upload_models(Y, Ye2, Ym if Ym2 is None else Ym2)
                eng.set_labels(z, ze if ze2 is None else ze2, label_dict=pd)   # labels belong to the model set they were uploaded with
                o2 = like_opts(kw2)
                xm = dXm if dXm2 is None else dXm2
                eng.fit_predict_prior(dX, dXe, xm, o2, ko, None, d_pdf, d_lm, d_le, n=N)
                eng.timing_reset()
                t0 = time.perf_counter()
                for _ in range(2):
                    eng.fit_predict_prior(dX, dXe, xm, o2, ko, None, d_pdf, d_lm, d_le, n=N)
                eng.sync()
                dt2 = (time.perf_counter() - t0) / 2
                tm2 = eng.timing()
                ms2 = tm2["ms_fused"] / max(tm2["n_fused"], 1)
                fl = flops_per_eval(mode2, args.nband)
                a2 = N * M / (max(tm2["n_fused"], 1) / 2) * fl / (ms2 * 1e-3) / 1e12
                return {"value": N * M / dt2, "unit": "evals/s", "ms_per_step": dt2 * 1e3, "achieved": a2, "peak": FP64_VALU_PEAK_TFLOPS,
                        "frac": a2 / FP64_VALU_PEAK_TFLOPS, "flops_per_eval": fl, "avg_launch_ms": ms2, "kernel": eng.last_form()}
            out["nz_stack"] = _sub(lambda: sub_nz_stack(eng, dev, torch, d_pdf, N, pd.Ngrid))      # (on the PDFs of the timed steps)
            out["roofline_fp64"] = dict(out["roofline"], value=value, ms_per_step=dt / args.steps * 1e3,
                                        note="the headline line IS the all-fp64 form since round 4 (same numbers, kept under the old key)")
            out["roofline_general"] = dict(extra(Ye * np.random.RandomState(77).uniform(0.5, 1.5, size=Ye.shape), {}, "A"),
                                           note="mode A with per-model errors (--model-err varying): the general kernels")
            out["roofline_modeB"] = dict(extra(Ye, MODES["B"], "B"), note="free scale, model errors ignored (--mode B)")
            # the shape of a real training catalogue (pdf.py:76-87 with models_mask and per-model models_err): per-model errors, 2 % of the
            # object bands and 2 % of the model bands missing -- N_dim differs from pair to pair
            Ym_c = Ym.copy(); Ym_c[np.random.RandomState(6).rand(*Ym.shape) < 0.02] = 0.0
            Xm_c = Xm.copy(); Xm_c[np.random.RandomState(5).rand(*Xm.shape) < 0.02] = 0.0
            out["roofline_catalogue"] = dict(extra(Ye * np.random.RandomState(77).uniform(0.5, 1.5, size=Ye.shape), {}, "A", Ym_c,
                                                   torch.from_numpy(Xm_c).to(dev)),
                                             note="mode A, per-model errors, 2 % of object bands and 2 % of model bands missing "
                                                  "(--model-err varying --mask-frac 0.02 --model-mask-frac 0.02)")
            # ... and with per-model LABEL errors on top (gauss_kde_dict with ~30 kernel widths, pdf.py:599-620): what a training set is
            out["roofline_catalogue_widths"] = dict(extra(Ye * np.random.RandomState(77).uniform(0.5, 1.5, size=Ye.shape), {}, "A", Ym_c,
                                                          torch.from_numpy(Xm_c).to(dev), np.random.RandomState(78).uniform(0.01, 0.1, size=ze.shape)),
                                                    note="the catalogue line with per-model label errors "
                                                         "(--model-err varying --mask-frac 0.02 --model-mask-frac 0.02 --label-err varying)")
        if "roofline_general" in out:
            # every BASELINE config the line's own workload does not cover, as sub-records (2-3 steps each; a failure is recorded, not raised)
            pp = _sub(lambda: sub_planes_predict(eng, dev, torch, pd))
            out["planes"] = pp.get("planes", pp); out["predict"] = pp.get("predict", pp)
            out["modeC"] = _sub(lambda: sub_modec(eng, dev, torch, pd))
            out["knn"] = _sub(lambda: sub_knn(dev, torch, pd, local))
            del dX, dXe, dXm, d_pdf
            torch.cuda.empty_cache()
            out["host_path"] = _sub(lambda: sub_host_path(local, pd))
        if cpu1 is not None:
            out["cpu_baseline"] = cpu1
            out["speedup_vs_cpu_core"] = value / cpu1["value"]
        if cpuall is not None:
            out["cpu_baseline_all_cores"] = cpuall
            out["speedup_vs_cpu_all_cores"] = value / cpuall["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
