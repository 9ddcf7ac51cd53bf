"""
Mapping models onto the nodes of a trained network (SOM / GNG) -- the inference step
``_Network.populate_network`` of the reference (frankenz/networks.py:176-356).
The (Nmodel, Nnode) likelihoods, the hot part, run on the GPU through the same kernels as
``BruteForce.fit`` with the nodes as noiseless models (networks.py:305-307); the ragged
per-node lists are host bookkeeping, as in the reference.  Training the network is out of
scope.
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
    out = NetworkMap()
    out.nodes_idxs = [[] for _ in range(Nnodes)]
    out.nodes_logwts = [[] for _ in range(Nnodes)]
    out.nodes_bmus = [[] for _ in range(Nnodes)]
    out.nodes_scales = [[] for _ in range(Nnodes)]
    out.nodes_scales_err = [[] for _ in range(Nnodes)]
    out.nodes_Nmatch = np.zeros(Nnodes, dtype='int')
    out.models_lmap = np.zeros(Nmodels) - np.inf
    out.models_levid = np.zeros(Nmodels) - np.inf
    out.results = []
    for i in range(Nmodels):
        node_lnprob = lnprob[i]
        out.nodes_bmus[int(np.argmax(node_lnprob))].append(i)
        if wt_thresh is not None:
            lwt_min = np.log(wt_thresh) + np.max(node_lnprob)
            n_idxs = np.arange(Nnodes)[node_lnprob > lwt_min]
        else:
            idx_sort = np.argsort(node_lnprob)
            node_prob = np.exp(node_lnprob - logsumexp(node_lnprob))
            node_cdf = np.cumsum(node_prob[idx_sort])
            n_idxs = idx_sort[node_cdf <= (1. - cdf_thresh)]
        n_lnprobs = node_lnprob[n_idxs]
        n_lmap, n_levid = np.max(n_lnprobs), logsumexp(n_lnprobs)
        n_lnprobs -= n_levid
        out.models_lmap[i] = n_lmap
        out.models_levid[i] = n_levid
        if track_scale:
            n_scales, n_scales_err = scale[i][n_idxs], scale_err[i][n_idxs]
        else:
            n_scales, n_scales_err = np.ones_like(n_idxs), np.zeros_like(n_idxs)
        for j, lwt, s, serr in zip(n_idxs, n_lnprobs, n_scales, n_scales_err):
            out.nodes_idxs[j].append(i)
            out.nodes_logwts[j].append(lwt)
            out.nodes_scales[j].append(s)
            out.nodes_scales_err[j].append(serr)
            out.nodes_Nmatch[j] += 1
        out.results.append((n_idxs, n_lnprobs, n_scales, n_scales_err))
    return out
