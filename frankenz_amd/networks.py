"""
Mapping models onto the nodes of a trained network (SOM / GNG) -- the inference step
``_Network.populate_network`` of the reference (frankenz/networks.py:176-356).
The (Nmodel, Nnode) likelihoods, the hot part, run on the GPU through the same kernels as
``BruteForce.fit`` with the nodes as noiseless models (networks.py:305-307).  The thresholding,
the per-model max / logsumexp and the ragged per-node lists are segmented NumPy reductions over
the (model, node) pair list of the whole plane -- no per-model Python loop.  Training the network
is out of scope.
"""
import numpy as np
from scipy.special import logsumexp

from .engine import HostObjects, get_engine, like_opts

__all__ = ["populate_network"]


class NetworkMap(object):
    """the attributes ``_Network._populate_network`` fills (networks.py:296-303)"""
    pass


def populate_network(nodes, models, models_err, models_mask, lpnet_kwargs=None, wt_thresh=1e-3,
                     cdf_thresh=2e-4, track_scale=True, device=None):
    """Map every model onto the nodes it is compatible with (networks.py:244-354).
    Returns an object with ``nodes_idxs, nodes_logwts, nodes_bmus, nodes_scales,
    nodes_scales_err, nodes_Nmatch, models_lmap, models_levid`` exactly as the reference
    leaves them on the network, plus ``results``: the per-model tuples the generator yields."""
    if lpnet_kwargs is None:
        lpnet_kwargs = {'free_scale': True, 'ignore_model_err': True, 'return_scale': True}
    if wt_thresh is None and cdf_thresh is None:
        wt_thresh = -np.inf
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    Nnodes, Nmodels = len(nodes), len(models)
    eng = get_engine(device)
    eng.upload_models(nodes, np.zeros_like(nodes), np.ones_like(nodes))     # networks.py:305-307
    obj = HostObjects(models, models_err, models_mask)
    opts = like_opts(lpnet_kwargs)
    free = bool(opts.free_scale)
    lnprob = np.empty((Nmodels, Nnodes))
    scale = np.ones((Nmodels, Nnodes)); scale_err = np.zeros((Nmodels, Nnodes))
    want_scale = track_scale and free
    eng.fit(obj.x, obj.xe, obj.xm, opts, lnprob, None, None, scale if want_scale else None,
            scale_err if want_scale else None)
    obj.writeback()
    if track_scale and not free:
        raise ValueError("track_scale=True needs a likelihood that returns the scale (free_scale=True)")
    return _lists_from_plane(lnprob, scale, scale_err, wt_thresh, cdf_thresh, track_scale)


def _lists_from_plane(lnprob, scale, scale_err, wt_thresh, cdf_thresh, track_scale):
    """the (Nmodels, Nnodes) ln-prob plane -> the per-model selections and the per-node lists"""
    Nmodels, Nnodes = lnprob.shape
    out = NetworkMap()
    # Everything below works on the whole (Nmodels, Nnodes) plane at once; the reference walks the models
    # one by one (networks.py:310-354).  What it leaves behind is reproduced through the ORDER of the
    # (model, node) pairs: models ascending, and inside a model the reference's own node order
    # (ascending node index under wt_thresh, ascending ln-prob under the CDF rule).
    rows = np.arange(Nmodels)
    out.models_bmu = np.argmax(lnprob, axis=1)                       # best-matching unit per model
    if wt_thresh is not None:
        keep = lnprob > (np.log(wt_thresh) + lnprob.max(axis=1))[:, None]          # strict, networks.py:319-321
        pair_model, pair_node = np.nonzero(keep)                     # row-major: node index ascending inside a model
    else:
        order = np.argsort(lnprob, axis=1)                           # networks.py:323-327: ascending sort, keep cdf <= 1 - cdf_thresh
        prob = np.exp(lnprob - logsumexp(lnprob, axis=1)[:, None])
        cdf = np.cumsum(np.take_along_axis(prob, order, axis=1), axis=1)
        pair_model, col = np.nonzero(cdf <= (1. - cdf_thresh))
        pair_node = order[pair_model, col]
    pair_lnp = lnprob[pair_model, pair_node]
    # per-model max and logsumexp of the kept entries (segmented reductions over the pair list)
    counts = np.bincount(pair_model, minlength=Nmodels)
    starts = np.concatenate(([0], np.cumsum(counts)))[:-1]
    has = counts > 0
    lmap = np.full(Nmodels, -np.inf)
    lmap[has] = np.maximum.reduceat(pair_lnp, starts[has])
    shifted = np.exp(pair_lnp - lmap[pair_model])
    ssum = np.zeros(Nmodels)
    ssum[has] = np.add.reduceat(shifted, starts[has])
    with np.errstate(divide='ignore'):
        levid = lmap + np.log(ssum)
    levid[~has] = -np.inf
    out.models_lmap, out.models_levid = lmap, levid
    pair_logwt = pair_lnp - levid[pair_model]                        # networks.py:331
    if track_scale:
        pair_s, pair_se = scale[pair_model, pair_node], scale_err[pair_model, pair_node]
    else:
        pair_s, pair_se = np.ones_like(pair_node), np.zeros_like(pair_node)        # integer ones / zeros like the reference
    # per-node lists: the pairs regrouped by node, models ascending within a node (stable sort)
    by_node = np.argsort(pair_node, kind='stable')
    cuts = np.cumsum(np.bincount(pair_node, minlength=Nnodes))[:-1]
    out.nodes_idxs = [a.tolist() for a in np.split(pair_model[by_node], cuts)]
    out.nodes_logwts = [a.tolist() for a in np.split(pair_logwt[by_node], cuts)]
    out.nodes_scales = [a.tolist() for a in np.split(pair_s[by_node], cuts)]
    out.nodes_scales_err = [a.tolist() for a in np.split(pair_se[by_node], cuts)]
    out.nodes_Nmatch = np.bincount(pair_node, minlength=Nnodes).astype('int')
    bmu_order = np.argsort(out.models_bmu, kind='stable')
    out.nodes_bmus = [a.tolist() for a in np.split(rows[bmu_order], np.cumsum(np.bincount(out.models_bmu, minlength=Nnodes))[:-1])]
    # the per-model tuples the reference's generator yields
    mcuts = np.cumsum(counts)[:-1]
    out.results = list(zip(np.split(pair_node, mcuts), np.split(pair_logwt, mcuts), np.split(pair_s, mcuts), np.split(pair_se, mcuts)))
    return out
