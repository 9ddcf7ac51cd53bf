"""Checks that need torch tensors / streams on the GPU, run in ONE fresh process by tests/test_hip_device_resident.py (the suite's
own process never brings up a second GPU runtime stack).  Prints one ``<NAME>_OK`` marker per check.
    python tests/_gpu_torch_checks.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIG = np.array([0.873, 0.348, 0.418, 0.873, 3.476])


def problem(N, M, seed):
    rs = np.random.RandomState(seed)
    Y = rs.lognormal(1., 1., size=(M, 5)); Ye = np.tile(SIG, (M, 1)); Ym = np.ones((M, 5))
    X = Y[rs.choice(M, N)] + SIG * rs.randn(N, 5); Xe = np.tile(SIG, (N, 1)); Xm = np.ones((N, 5))
    z = rs.uniform(0, 6, M); ze = np.full(M, 0.05)
    return Y, Ye, Ym, X, Xe, Xm, z, ze


def check_stream_contract():
    """fz_set_producer_stream: with the caller's stream named, a library call neither waits for nor is delayed by work in flight
    on ANOTHER stream (the place of RCCL's all-gather in sharded._overlapped); the default device-wide wait drains it."""
    import torch
    from frankenz_amd import BruteForce, PDFDict
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    pd = PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(20000, 3000, 11)
    bf = BruteForce(Y, Ye, Ym, device=0)
    prep = bf.prepare_fit_predict(z, ze, label_dict=pd)
    dX, dXe, dXm = (torch.from_numpy(a).to(dev) for a in (X, Xe, Xm))
    outs = [(torch.empty((len(X), pd.Ngrid), dtype=torch.float64, device=dev), torch.empty(len(X), dtype=torch.float64, device=dev),
             torch.empty(len(X), dtype=torch.float64, device=dev)) for _ in range(2)]
    prep.run(dX, dXe, dXm, out=outs[0])                       # warm-up: allocations, first-launch probe
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    # a spin kernel on the side stream that outlasts the call by far: calibrate cycles -> seconds first
    t0 = time.perf_counter(); torch.cuda._sleep(50_000_000); torch.cuda.synchronize(); per = (time.perf_counter() - t0) / 50_000_000
    cycles = int(1.5 / per)
    eng = prep.eng

    def trial(mode, out):
        if mode:
            eng.set_producer_stream(torch.cuda.current_stream().cuda_stream, 1)
        else:
            eng.set_producer_stream(None, 0)
        ev = torch.cuda.Event()
        with torch.cuda.stream(side):
            torch.cuda._sleep(cycles)
            ev.record(side)
        t0 = time.perf_counter()
        prep.run(dX, dXe, dXm, out=out)
        dt = time.perf_counter() - t0
        done = ev.query()                                     # has the side stream's work finished by the time the call returned?
        torch.cuda.synchronize()
        eng.set_producer_stream(None, 0)
        return dt, done
    dt1, done1 = trial(1, outs[0])
    dt0, done0 = trial(0, outs[1])
    assert not done1, "the call waited for the side stream although the producer stream was named (%.3f s)" % dt1
    assert done0, "the default contract should have drained the device"
    assert dt1 < 0.5 and dt0 > 1.0, (dt1, dt0)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # inputs produced on the named stream ARE waited for: the objects are written by a kernel queued behind a sleep on that stream
    eng.set_producer_stream(torch.cuda.current_stream().cuda_stream, 1)
    dX2 = torch.zeros_like(dX)
    torch.cuda._sleep(int(0.2 / per))
    dX2.copy_(dX)                                             # queued behind the sleep on the current stream
    o3 = tuple(torch.empty_like(t) for t in outs[0])
    prep.run(dX2, dXe, dXm, out=o3)
    eng.set_producer_stream(None, 0)
    torch.cuda.synchronize()
    for a, b in zip(outs[0], o3):
        assert torch.equal(a, b)
    print("STREAM_CONTRACT_OK dt_named=%.4f dt_default=%.4f" % (dt1, dt0))


def check_knn_device_resident():
    """NearestNeighbors.fit_predict with device tensors + out= + query_features: the NumPy call's results, nothing through the host"""
    import torch
    from frankenz_amd import NearestNeighbors, PDFDict
    dev = torch.device('cuda', 0)
    pd = PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(5000, 4000, 12)
    Y = Y * 3; Ye = 0.05 * Y
    fk = dict(skynoise=SIG, zeropoints=10 ** (0.4 * 23.9))
    nn = NearestNeighbors(Y, Ye, Ym, K=7, feature_map='luptitude', fmap_kwargs=fk, rstate=np.random.RandomState(1), verbose=False, device=0)
    Xc = X.copy()
    p, (lm, le) = nn.fit_predict(Xc, Xe.copy(), Xm.copy(), z, ze, rstate=np.random.RandomState(2), k=6, label_dict=pd, return_gof=True,
                                 save_fits=False, verbose=False)
    # the same through stored fits (the two-step surface of the reference): identical PDFs
    nn.fit(X.copy(), Xe.copy(), Xm.copy(), rstate=np.random.RandomState(2), k=6, verbose=False)
    p2 = nn.predict(z, ze, label_dict=pd, verbose=False)
    np.testing.assert_allclose(p2, p, rtol=1e-12, atol=1e-15)
    assert nn.neighbors.shape == (len(X), 42) and (nn.Nneighbors > 0).all()
    # device-resident
    q = nn._query_features(X, Xe, np.random.RandomState(2))
    dX, dXe, dXm, dQ = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (X, Xe, Xm, q))
    out = (torch.empty((len(X), pd.Ngrid), dtype=torch.float64, device=dev), torch.empty(len(X), dtype=torch.float64, device=dev),
           torch.empty(len(X), dtype=torch.float64, device=dev))
    r = nn.fit_predict(dX, dXe, dXm, z, ze, k=6, label_dict=pd, return_gof=True, save_fits=False, out=out, query_features=dQ)
    assert r[0] is out[0]
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[0].cpu().numpy(), p, rtol=1e-12, atol=1e-15, equal_nan=True)
    np.testing.assert_array_equal(out[1].cpu().numpy(), lm)
    np.testing.assert_allclose(out[2].cpu().numpy(), le, rtol=1e-12, equal_nan=True)
    print("KNN_DEVICE_RESIDENT_OK")


def check_prepared_handle_survives_other_fitters():
    """the engine is process-wide: a ``prepare_fit_predict`` handle re-establishes its model set / labels when another fitter
    has used the device in between (content keys: free when nothing changed)"""
    from frankenz_amd import BruteForce, PDFDict
    pd = PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(300, 2000, 13)
    Y2, Ye2, Ym2, X2, Xe2, Xm2, z2, ze2 = problem(200, 1500, 14)
    bf = BruteForce(Y, Ye, Ym, device=0)
    prep = bf.prepare_fit_predict(z, ze, label_dict=pd)
    a = prep.run(X.copy(), Xe.copy(), Xm.copy())
    BruteForce(Y2, Ye2, Ym2, device=0).fit_predict(X2, Xe2, Xm2, z2, ze2, label_grid=np.linspace(0, 6, 400), save_fits=False, verbose=False)
    b = prep.run(X.copy(), Xe.copy(), Xm.copy())
    for u, v in zip(a, b):
        np.testing.assert_array_equal(np.asarray(u), np.asarray(v))
    print("PREPARED_HANDLE_OK")


if __name__ == '__main__':
    check_stream_contract()
    check_knn_device_resident()
    check_prepared_handle_survives_other_fitters()
