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


def check_overlapped_rounds_through_rccl_on_one_rank():
    """sharded._overlapped end to end with backend nccl (= RCCL) in a one-rank group: block-cyclic rounds, the stream contract,
    async in-place all_gather_into_tensor per round, one fence -- for BruteForce and for NearestNeighbors -- must return what the
    plain calls return (with one rank the collective moves nothing, but every call of the N > 1 path is made)."""
    import torch
    import torch.distributed as dist
    from frankenz_amd import BruteForce, NearestNeighbors, PDFDict, sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device('cuda', 0)
    pd = PDFDict(np.arange(0, 7 + 1e-5, .01), np.linspace(.005, 2, 500))
    Y, Ye, Ym, X, Xe, Xm, z, ze = problem(30011, 2500, 15)
    X[3, 1] = np.nan
    bf = BruteForce(Y, Ye, Ym, device=0)
    p, (lm, le) = bf.fit_predict(X.copy(), Xe.copy(), Xm.copy(), z, ze, label_dict=pd, save_fits=False, return_gof=True, verbose=False)
    dX, dXe, dXm = (torch.from_numpy(a.copy()).to(dev) for a in (X, Xe, Xm))
    for chunks in (1, 4, 7):
        q, (qm, qe) = sharded.sharded_fit_predict(bf, dX.clone(), dXe.clone(), dXm.clone(), z, ze, gather='pdfs', chunks=chunks,
                                                  label_dict=pd, save_fits=False, rounds_on_one_rank=True)
        assert sharded.last_stats['world'] == 1 and sharded.last_stats['chunks'] == chunks
        torch.cuda.synchronize()
        np.testing.assert_allclose(q.cpu().numpy(), p, rtol=1e-12, atol=1e-16, equal_nan=True)
        # (the one call of 30 011 objects samples how broad the likelihoods are and may take the direct form; the rounds are too short
        #  to sample and take the screen form: the two form the best weight by different -- equally exact -- fp64 expressions, so
        #  ln-max agrees to two ulps, not bit for bit)
        np.testing.assert_allclose(qm.cpu().numpy(), lm, rtol=1e-14, equal_nan=True)
        np.testing.assert_allclose(qe.cpu().numpy(), le, rtol=1e-12, equal_nan=True)
    # NumPy in, NumPy out (the caller's arrays get the in-place clean)
    Xc = X.copy()
    q, (qm, qe) = sharded.sharded_fit_predict(bf, Xc, Xe.copy(), Xm.copy(), z, ze, gather='pdfs', chunks=3, label_dict=pd,
                                              save_fits=False, rounds_on_one_rank=True)
    assert np.isfinite(Xc).all()
    np.testing.assert_allclose(q, p, rtol=1e-12, atol=1e-16, equal_nan=True)
    # k-NN
    Y2 = Y * 3; Ye2 = 0.05 * Y2
    X2 = Y2[np.random.RandomState(4).choice(len(Y2), 6001)] + SIG * np.random.RandomState(5).randn(6001, 5); Xe2 = np.tile(SIG, (6001, 1)); Xm2 = np.ones((6001, 5))
    nn = NearestNeighbors(Y2, Ye2, Ym, K=7, feature_map='luptitude', fmap_kwargs=dict(skynoise=SIG, zeropoints=10 ** (0.4 * 23.9)),
                          rstate=np.random.RandomState(1), verbose=False, device=0)
    p2, (lm2, le2) = nn.fit_predict(X2.copy(), Xe2.copy(), Xm2.copy(), z, ze, rstate=np.random.RandomState(2), k=6, label_dict=pd,
                                    return_gof=True, save_fits=False, verbose=False)
    q2, (qm2, qe2) = sharded.sharded_fit_predict(nn, X2.copy(), Xe2.copy(), Xm2.copy(), z, ze, gather='pdfs', chunks=4, label_dict=pd,
                                                 save_fits=False, rstate=np.random.RandomState(2), k=6, rounds_on_one_rank=True)
    np.testing.assert_allclose(q2, p2, rtol=1e-12, atol=1e-16)
    np.testing.assert_array_equal(qm2, lm2)
    dist.destroy_process_group()
    print("OVERLAPPED_RCCL_ONE_RANK_OK ms_compute=%.2f ms_fence=%.3f" % (sharded.last_stats['ms_compute'], sharded.last_stats['ms_fence']))


if __name__ == '__main__':
    check_stream_contract()
    check_knn_device_resident()
    check_prepared_handle_survives_other_fitters()
    check_overlapped_rounds_through_rccl_on_one_rank()
