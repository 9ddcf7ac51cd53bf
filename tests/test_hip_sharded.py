"""The N>1 path with the REAL HIP engine (north_star: the N_obj axis shards, all-gather of the stacked
PDFs): ``sharded_fit_predict`` in one process, and in two gloo ranks that share GPU 0 (each a fresh
process started before it touches the GPU), must give what the unsharded call gives.  Includes the
configs[4] substitute end to end: catalogue -> fused PDFs -> stack  sum_i pdf_i  (all-reduce) ->
``samplers.loglike_nz``, against the reference's outputs (golden g12)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import EVID, load_golden

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def run_ranks(name, world, tmp_path):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_gpu_shard_worker.py'), name, str(r), str(world), str(port), str(tmp_path)],
                              env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(world)]


def unsharded(name):
    sys.path.insert(0, HERE)
    import _gpu_shard_worker as w
    mk, X, Xe, Xm, z, ze, kw = w.scenario(name)
    if name == 'knn':
        kw = dict(kw, rstate=np.random.RandomState(2))
    Xc, Xmc = X.copy(), Xm.copy()
    p, (lm, le) = mk().fit_predict(Xc, Xe.copy(), Xmc, z, ze, return_gof=True, verbose=False, **kw)
    return p, lm, le, Xc, Xmc


@pytest.mark.parametrize('name,world', [('bf_big', 1), ('bf_big', 2), ('bf_ragged', 4), ('knn', 2)])
def test_sharded_hip_matches_the_unsharded_call(name, world, tmp_path):
    """bruteforce.py:602-631 has no cross-object state: blocks of objects computed by different ranks and
    all-gathered equal the single call (same launch geometry per object block: bitwise for ln-max, rounding
    level for PDFs / ln-evidence); the stack is the sum of the rows; the in-place clean reaches the caller's
    block on each rank; k-NN Monte-Carlo draws do not depend on the rank count."""
    p, lm, le, Xc, Xmc = unsharded(name)
    res = run_ranks(name, world, tmp_path)
    for r, out in enumerate(res):
        np.testing.assert_allclose(out['pdfs'], p, rtol=1e-12, atol=1e-16, equal_nan=True)
        # (ln-max: ln L(mode) + ln of the best exact weight -- bit for bit when both launches take the same kernel form, two ulps when the
        #  sampled breadth of the likelihoods sends one of them to the direct form and the other to the screen form)
        np.testing.assert_allclose(out['lmap'], lm, rtol=1e-14, equal_nan=True)
        np.testing.assert_allclose(out['levid'], le, equal_nan=True, **EVID)
        assert out['pdfs'].shape == p.shape
        np.testing.assert_allclose(out['stack'], np.nansum(p, axis=0), rtol=1e-11, atol=1e-14)
        from frankenz_amd.sharded import shard_slice
        sl = shard_slice(len(p), world, r)
        np.testing.assert_array_equal(out['x'][sl], Xc[sl]); np.testing.assert_array_equal(out['xm'][sl], Xmc[sl])


@pytest.mark.parametrize('tag,world', [('grid', 1), ('train', 2)])
def test_configs4_substitute_catalogue_to_stacked_nz(tag, world, tmp_path):
    """BASELINE configs[4] (SURVEY 8d "Config 5" substitute): SDSS-like catalogue from the reference's simulator
    -> fused BruteForce PDFs -> stacked n(z) = all-reduce of  sum_i pdf_i  -> population ln-likelihood
    (samplers.py:60-80), against the reference's own BruteForce / loglike_nz outputs (golden g12)."""
    from frankenz_amd import samplers
    g = load_golden('g12_catalogue_stack')
    out = run_ranks('g12_' + tag, world, tmp_path)[0]
    p = out['pdfs']
    np.testing.assert_allclose(p[::10], g[tag + '_pdfs_every10'], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(out['lmap'], g[tag + '_lmap'], rtol=1e-9)
    np.testing.assert_allclose(out['levid'], g[tag + '_levid'], **EVID)
    np.testing.assert_allclose(out['stack'], g[tag + '_stack'], rtol=1e-9, atol=1e-12)
    stack = out['stack']
    for nm, nzv in (('stack', stack / stack.sum()), ('flat', np.full(len(stack), 1. / len(stack)))):
        ll, ov = samplers.loglike_nz(nzv, p, return_overlap=True)
        np.testing.assert_allclose(ov, g['%s_overlap_%s' % (tag, nm)], rtol=1e-8, atol=1e-300)
        np.testing.assert_allclose(ll, float(g['%s_llnz_%s' % (tag, nm)]), rtol=1e-10)


@pytest.mark.gpu
def test_rccl_accepts_the_in_place_gather_of_the_overlapped_path():
    """The overlapped sharded call hands RCCL a slab of the result and, as input, the rank's own rows INSIDE that slab
    (`all_gather_into_tensor` in place, async).  A one-rank `nccl` group on the box's GPU runs that exact call form through
    RCCL (a fresh process: the suite's own process holds no process group); more ranks need more GPUs (bench.py --gpus N)."""
    import subprocess, sys, os
    code = r'''
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
buf = torch.arange(12 * 7, dtype=torch.float64, device="cuda").reshape(12, 7).clone()
ref = buf.clone()
works = []
for base in (0, 4, 8):
    out_v, in_v = buf[base:base + 4], buf[base:base + 4]
    works.append(dist.all_gather_into_tensor(out_v, in_v, async_op=True))
for w in works: w.wait()
torch.cuda.synchronize()
assert torch.equal(buf, ref)
dist.destroy_process_group()
print("RCCL_IN_PLACE_OK")
'''
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "RCCL_IN_PLACE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
