"""
Inference through a trained network (SOM / GNG): the reference's ``_Network`` without its training (frankenz/networks.py:120-1473,
SURVEY 8f-4).  The network is DATA here -- node positions in data space -- and everything the reference then does with it runs
through the library:

* ``populate_network`` (networks.py:244-356): (Nmodel, Nnode) likelihoods with the nodes as noiseless models (``k_planes``), the
  thresholding rule, per-model max / logsumexp of the kept entries (``fz_net_select``); the per-node lists are the transpose of the
  kept (model, node) pairs -- integer bookkeeping, done with NumPy sorts.
* ``fit`` / ``predict`` / ``fit_predict`` (networks.py:782-936, 938-1128, 1130-1473): node likelihoods of the objects, thresholding
  (``fz_net_select``), then either the selected nodes as the models (``nodes_only``: ``fz_net_gather`` / ``fz_net_stack`` on the node
  PDFs) or the union of the selected nodes' model lists (``fz_net_table``) through the k-NN subset kernel (``fz_knn_fit_predict``:
  first-appearance de-dup = pandas.unique, likelihood, weights, KDE).
* ``get_pdfs`` (networks.py:413-560): per-node KDE of the member models (``fz_knn_predict_logwt``), scaled by exp(levid).

A foreign ``lpnet_func`` / ``lprob_func`` is the user's code: it is called on the host per object exactly as the reference calls it;
the selection, the unions and the PDFs still run on the device.
"""
import sys

import numpy as np

from . import pdf as _pdf
from .engine import HostObjects, get_engine, kde_opts, like_opts, merge_kde_args

__all__ = ["populate_network", "Network"]

_NET_CHUNK = 1 << 16          # objects per device call


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
    return _lists_from_plane(lnprob, scale, scale_err, wt_thresh, cdf_thresh, track_scale, eng)


def _select(eng, lnprob, wt_thresh, cdf_thresh, match=None, csr_off=None, want_stats=False):
    """fz_net_select on a host plane: (nsel, sel[, rawlen][, lmap, levid])"""
    n, nn = lnprob.shape
    nsel = np.zeros(n, dtype=np.int32); sel = np.zeros((n, nn), dtype=np.int32)
    rawlen = np.zeros(n, dtype=np.int64) if csr_off is not None else None
    lmap = np.zeros(n) if want_stats else None; levid = np.zeros(n) if want_stats else None
    use_wt = wt_thresh is not None
    eng.net_select(lnprob, use_wt, wt_thresh if use_wt else 0.0, 0.5 if use_wt or cdf_thresh is None else cdf_thresh, match, csr_off,
                   nsel, sel, rawlen, lmap, levid)
    return nsel, sel, rawlen, lmap, levid


def _lists_from_plane(lnprob, scale, scale_err, wt_thresh, cdf_thresh, track_scale, eng=None):
    """the (Nmodels, Nnodes) ln-prob plane -> the per-model selections and the per-node lists"""
    Nmodels, Nnodes = lnprob.shape
    out = NetworkMap()
    # The selection of every model's nodes (either rule, in the reference's order) and the max / logsumexp over the kept entries
    # come from the device (fz_net_select, networks.py:316-333); what follows is the transpose of the kept (model, node) pairs into
    # per-node lists -- models ascending inside a node, as the reference's model loop appends them.
    rows = np.arange(Nmodels)
    out.models_bmu = np.argmax(lnprob, axis=1)                       # best-matching unit per model
    eng = eng if eng is not None else get_engine(None)
    nsel, sel, _, lmap, levid = _select(eng, np.ascontiguousarray(lnprob), wt_thresh, cdf_thresh, want_stats=True)
    return _lists_from_selection(lnprob, scale, scale_err, nsel, sel, lmap, levid, track_scale)


def _lists_from_selection(lnprob, scale, scale_err, nsel, sel, lmap, levid, track_scale):
    """the transpose: per-model selections (nsel, sel: fz_net_select) -> per-node lists, as networks.py:335-352 appends them"""
    Nmodels, Nnodes = lnprob.shape
    out = NetworkMap()
    rows = np.arange(Nmodels)
    out.models_bmu = np.argmax(lnprob, axis=1)
    keep = np.arange(Nnodes)[None, :] < nsel[:, None]
    pair_model = np.nonzero(keep)[0]
    pair_node = sel[keep].astype(np.int64)
    pair_lnp = lnprob[pair_model, pair_node]
    counts = nsel.astype(np.int64)
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


def _csr(lists):
    """ragged per-node lists -> (offsets int64[Nnodes + 1], items int64)"""
    lens = np.array([len(v) for v in lists], dtype=np.int64)
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    items = np.concatenate([np.asarray(v, dtype=np.int64) for v in lists]) if off[-1] else np.zeros(1, dtype=np.int64)
    return off, np.ascontiguousarray(items)


def _is_default(func):
    return func is None or func is _pdf.logprob


class Network(object):
    """The inference half of the reference's ``_Network`` (networks.py:120-1473): same constructor, methods, keywords, defaults and
    attributes; the network itself is handed over as data (``set_nodes``) instead of being trained here."""

    def __init__(self, models, models_err, models_mask, device=None):
        self.models, self.models_err, self.models_mask = models, models_err, models_mask
        self.NMODEL, self.NDIM = models.shape
        self.models_lmap = np.zeros(self.NMODEL) - np.inf
        self.models_levid = np.zeros(self.NMODEL) - np.inf
        self.fit_lnprior = self.fit_lnlike = self.fit_lnprob = self.fit_Ndim = self.fit_chi2 = None
        self.fit_scale = self.fit_scale_err = None
        self.nodes = self.nodes_pos = self.nodes_idxs = self.nodes_logwts = self.nodes_bmus = None
        self.nodes_scales = self.nodes_scales_err = self.nodes_Nmatch = self.nodes_only = None
        self.NNODE, self.NPROJ = None, None
        self.neighbors = self.Nneighbors = None
        self._device = device

    # -- the trained network, as data --------------------------------------------------------------------
    def set_nodes(self, nodes, nodes_pos=None):
        """node positions in data space (Nnode, Nfilt) [and on the manifold (Nnode, Nproj)]: what ``train_network`` leaves behind"""
        self.nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        self.NNODE = len(self.nodes)
        if nodes_pos is not None:
            self.nodes_pos = np.asarray(nodes_pos)
            self.NPROJ = self.nodes_pos.shape[1]
        return self

    def _eng(self):
        return get_engine(self._device)

    # -- networks.py:176-356 -----------------------------------------------------------------------------
    def populate_network(self, lpnet_func=None, wt_thresh=1e-3, cdf_thresh=2e-4, lpnet_args=None, lpnet_kwargs=None,
                         track_scale=True, verbose=True):
        for _ in self._populate_network(lpnet_func, wt_thresh, cdf_thresh, lpnet_args, lpnet_kwargs, track_scale):
            pass
        if verbose:
            sys.stderr.write('\rMapping objects 100%\n'); sys.stderr.flush()

    def _populate_network(self, lpnet_func=None, wt_thresh=1e-3, cdf_thresh=2e-4, lpnet_args=None, lpnet_kwargs=None,
                          track_scale=True):
        if self.nodes is None:
            raise ValueError("Network has not been trained!")
        if lpnet_func is None:
            lpnet_func = _pdf.logprob
        if lpnet_args is None:
            lpnet_args = []
        if lpnet_kwargs is None:
            lpnet_kwargs = {'free_scale': True, 'ignore_model_err': True, 'return_scale': True}
        if wt_thresh is None and cdf_thresh is None:
            wt_thresh = -np.inf
        self.lpnet_func, self.lpnet_args, self.lpnet_kwargs = lpnet_func, lpnet_args, lpnet_kwargs
        if _is_default(lpnet_func) and not lpnet_args:
            m = populate_network(self.nodes, self.models, self.models_err, self.models_mask, lpnet_kwargs=lpnet_kwargs, wt_thresh=wt_thresh,
                                 cdf_thresh=cdf_thresh, track_scale=track_scale, device=self._device)
        else:
            # the user's node likelihood: called per model as networks.py:310-312 does; selection and lists as above
            y, ye, ym = self.nodes, np.zeros_like(self.nodes), np.ones_like(self.nodes, dtype='bool')
            lnprob = np.empty((self.NMODEL, self.NNODE)); sc = np.ones_like(lnprob); se = np.zeros_like(lnprob)
            for i, (x, xe, xm) in enumerate(zip(self.models, self.models_err, self.models_mask)):
                r = lpnet_func(x, xe, xm, y, ye, ym, *lpnet_args, **lpnet_kwargs)
                lnprob[i] = r[2]
                if track_scale:
                    sc[i], se[i] = r[5], r[6]
            m = _lists_from_plane(lnprob, sc, se, wt_thresh, cdf_thresh, track_scale, self._eng())
        self.nodes_idxs, self.nodes_logwts, self.nodes_bmus = m.nodes_idxs, m.nodes_logwts, m.nodes_bmus
        self.nodes_scales, self.nodes_scales_err, self.nodes_Nmatch = m.nodes_scales, m.nodes_scales_err, m.nodes_Nmatch
        self.models_lmap, self.models_levid = m.models_lmap, m.models_levid
        for r in m.results:
            yield r

    # -- networks.py:358-411 -----------------------------------------------------------------------------
    def get_node(self, idx=None, pos=None, discrete=False):
        if idx is None and pos is None:
            raise ValueError("Either `idx` or `pos` must be specified.")
        elif idx is not None and pos is not None:
            raise ValueError("Both `idx` and `pos` cannot be specified.")
        elif pos is not None:
            idx = np.argmin([sum((pos - p)**2) for p in self.nodes_pos])
        if discrete:
            idxs = self.nodes_bmus[idx]
            logwts = np.zeros_like(idxs)
        else:
            idxs, logwts = self.nodes_idxs[idx], self.nodes_logwts[idx]
        return (idx, self.nodes[idx], self.nodes_pos[idx] if self.nodes_pos is not None else None, idxs, logwts,
                self.nodes_scales[idx], self.nodes_scales_err[idx])

    # -- networks.py:413-560 -----------------------------------------------------------------------------
    def _labels(self, eng, model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs):
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        kw = merge_kde_args(kde_args, kde_kwargs, label_dict is not None)
        eng.upload_models(self.models, self.models_err, self.models_mask)
        eng.set_labels(model_labels, model_label_errs, label_dict=label_dict, label_grid=label_grid, kde_kwargs=kw)
        return kde_opts(kw), (label_dict.Ngrid if label_dict is not None else len(label_grid))

    def get_pdfs(self, model_labels, model_label_errs, label_dict=None, label_grid=None, kde_args=None, kde_kwargs=None,
                 return_gof=False, discrete=False, verbose=True):
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        if self.nodes_idxs is None:
            raise ValueError("Network has not been trained!")
        eng = self._eng()
        ko, G = self._labels(eng, model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs)
        lists = self.nodes_bmus if discrete else self.nodes_idxs
        lens = np.array([len(v) for v in lists], dtype=np.int64)
        W = max(int(lens.max()), 1)
        nbr = np.zeros((self.NNODE, W), dtype=np.int64); lwt = np.full((self.NNODE, W), -np.inf)
        for i, v in enumerate(lists):
            nbr[i, :len(v)] = v
            lwt[i, :len(v)] = 0.0 if discrete else self.nodes_logwts[i]
        pdfs = np.zeros((self.NNODE, G)); lmap = np.zeros(self.NNODE); levid = np.zeros(self.NNODE)
        eng.knn_predict_logwt(lwt, nbr, lens, W, ko, pdfs, lmap, levid)
        empty = lens == 0
        pdfs[empty] = 0.0; lmap[empty] = -np.inf; levid[empty] = -np.inf
        pdfs *= np.exp(levid)[:, None]                              # scale to the associated object density (networks.py:548)
        if verbose:
            sys.stderr.write('\rGenerating node PDF {0}/{1}\n'.format(self.NNODE, self.NNODE)); sys.stderr.flush()
        return (pdfs, (lmap, levid)) if return_gof else pdfs

    def _get_pdfs(self, model_labels, model_label_errs, label_dict=None, label_grid=None, kde_args=None, kde_kwargs=None,
                  discrete=False):
        pdfs, (lmap, levid) = self.get_pdfs(model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs, True,
                                             discrete, False)
        for i in range(self.NNODE):
            yield pdfs[i], (lmap[i], levid[i])

    def get_pdf(self, idx, model_labels, model_label_errs, label_dict=None, label_grid=None, kde_args=None, kde_kwargs=None,
                return_gof=False, discrete=False):
        pdfs, (lmap, levid) = self.get_pdfs(model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs, True,
                                             discrete, False)
        return (pdfs[idx], (lmap[idx], levid[idx])) if return_gof else pdfs[idx]

    # -- the shared body of fit / fit_predict (networks.py:856-936, 1389-1473) ------------------------------
    def _run(self, data, data_err, data_mask, lprob_func, nodes_only, wt_thresh, cdf_thresh, lprob_args, lprob_kwargs, track_scale,
             discrete, save_fits, predict=None):
        """predict: None, or (model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs).  Returns per-object
        (idxs, Nidx, results[, pdf, (lmap, levid)]) tuples; stores the fits like the reference when save_fits."""
        if self.nodes_idxs is None:
            raise ValueError("Network has not been trained!")
        lprob_args = lprob_args or []
        lprob_kwargs = lprob_kwargs or {}
        if wt_thresh is None and cdf_thresh is None:
            wt_thresh = -np.inf
        eng = self._eng()
        Ndata = len(data)
        match_sel = np.arange(self.NNODE)[self.nodes_Nmatch > 0]
        match32 = np.ascontiguousarray(match_sel, dtype=np.int32)
        y = np.ascontiguousarray(self.nodes[match_sel]); Nn = len(y)
        self.nodes_only = nodes_only
        if save_fits:
            self.NDATA = Ndata
            self.Nneighbors = np.zeros(Ndata, dtype='int'); self.neighbors = []
            self.fit_lnprior, self.fit_lnlike, self.fit_lnprob, self.fit_Ndim, self.fit_chi2 = [], [], [], [], []
            self.fit_scale, self.fit_scale_err = [], []
        node_pdfs = None
        ko = G = None
        if predict is not None:
            labels, label_errs, label_dict, label_grid, kde_args, kde_kwargs = predict
            if nodes_only:
                node_pdfs = np.ascontiguousarray(self.get_pdfs(labels, label_errs, label_dict, label_grid, kde_args, kde_kwargs,
                                                               False, discrete, False))
                G = node_pdfs.shape[1]
        off, items = _csr(self.nodes_bmus if discrete else self.nodes_idxs)
        lp_default = _is_default(self.lpnet_func) and not self.lpnet_args
        host_lprob = not (_is_default(lprob_func) and not lprob_args)
        out = []
        for i0 in range(0, Ndata, _NET_CHUNK):
            sl = slice(i0, min(i0 + _NET_CHUNK, Ndata)); n = sl.stop - sl.start
            obj = HostObjects(data[sl], data_err[sl], data_mask[sl])
            # node likelihoods (networks.py:880-882): the matched nodes as noiseless, unmasked models
            lnp = np.empty((n, Nn)); chi2 = np.empty((n, Nn)); ndim = np.empty((n, Nn), dtype=np.int64)
            sc = np.ones((n, Nn)); se = np.zeros((n, Nn))
            if lp_default:
                eng.upload_models(y, np.zeros_like(y), np.ones_like(y))
                nopts = like_opts(self.lpnet_kwargs)
                free = bool(nopts.free_scale)
                eng.fit(obj.x, obj.xe, obj.xm, nopts, lnp, chi2, ndim, sc if free else None, se if free else None)
                obj.writeback()
                nres = [np.zeros((n, Nn)), lnp, lnp, ndim, chi2, sc, se]
            else:
                ye, ym = np.zeros_like(y), np.ones_like(y, dtype='bool')
                rows = [self.lpnet_func(x, xe, xm, y, ye, ym, *self.lpnet_args, **self.lpnet_kwargs)
                        for x, xe, xm in zip(data[sl], data_err[sl], data_mask[sl])]
                nres = [np.ascontiguousarray(np.array([r[k] for r in rows])) for k in range(len(rows[0]))]
                lnp = np.ascontiguousarray(nres[2], dtype=np.float64)
            nsel, sel, rawlen, _, _ = _select(eng, lnp, wt_thresh, cdf_thresh, match32, None if nodes_only else off)
            if nodes_only:
                W = max(int(nsel.max()), 1)
                res = []
                for k, pl in enumerate(nres):
                    pl = np.ascontiguousarray(pl)
                    if pl.dtype not in (np.float64, np.int64):
                        pl = pl.astype(np.float64)
                    g = np.empty((n, W), dtype=pl.dtype)
                    eng.net_gather(pl, nsel, sel, W, 0, g)
                    res.append(g)
                nb = np.empty((n, W), dtype=np.int64)
                eng.net_gather(np.ascontiguousarray(np.broadcast_to(match_sel.astype(np.int64), (n, Nn))), nsel, sel, W, -99, nb)
                nn_ = nsel.astype(np.int64)
                pdfs = lmap = levid = None
                if predict is not None:
                    pdfs = np.empty((n, G)); lmap = np.empty(n); levid = np.empty(n)
                    eng.net_stack(lnp, nsel, sel, match32, node_pdfs, pdfs, lmap, levid)
            else:
                W = max(int(rawlen.max()), 1)
                idx = np.empty((n, W), dtype=np.int64)
                eng.net_table(nsel, sel, match32, off, items, W, idx)
                nb = np.empty((n, W), dtype=np.int64); nn_ = np.empty(n, dtype=np.int64)
                eng.upload_models(self.models, self.models_err, self.models_mask)
                pdfs = lmap = levid = None
                if predict is not None:
                    ko, G = self._labels(eng, labels, label_errs, label_dict, label_grid, kde_args, kde_kwargs)
                    pdfs = np.empty((n, G)); lmap = np.empty(n); levid = np.empty(n)
                opts = like_opts({} if host_lprob else lprob_kwargs)
                lnl = np.empty((n, W)); c2 = np.empty((n, W)); nd = np.empty((n, W), dtype=np.int64)
                s_ = np.empty((n, W)); se_ = np.empty((n, W))
                eng.knn_fit_predict(obj.x, obj.xe, obj.xm, idx, W, opts, ko, nb, nn_, lnl, c2, nd, s_, se_,
                                    None if host_lprob else pdfs, None if host_lprob else lmap, None if host_lprob else levid)
                obj.writeback()
                if host_lprob:
                    # the user's likelihood on each object's model subset (networks.py:925-928), the PDFs from its ln-posteriors
                    lw = np.full((n, W), -np.inf)
                    rows = []
                    for k in range(n):
                        ii = nb[k, :nn_[k]]
                        r = lprob_func(data[sl][k], data_err[sl][k], data_mask[sl][k], self.models[ii], self.models_err[ii],
                                       self.models_mask[ii], *lprob_args, **lprob_kwargs)
                        rows.append(r); lw[k, :nn_[k]] = r[2]
                    if predict is not None:
                        # (the user's function may have used this device's context itself -- the package's own logprob does: the model and
                        #  label sets are put back before the PDFs are stacked)
                        ko, G = self._labels(eng, labels, label_errs, label_dict, label_grid, kde_args, kde_kwargs)
                        eng.knn_predict_logwt(lw, nb, nn_, W, ko, pdfs, lmap, levid)
                    res = rows
                else:
                    res = [np.zeros((n, W)), lnl, lnl, nd, c2, s_, se_]
            for k in range(n):
                m = int(nn_[k])
                idxs = nb[k, :m].copy()
                if isinstance(res, list) and len(res) and isinstance(res[0], tuple):
                    results = res[k]
                else:
                    results = [r[k, :m].copy() for r in res]
                    # (pdf.logprob returns seven arrays with return_scale, else five: pdf.py:404-411)
                    seven = (len(nres) > 5 and (not lp_default or self.lpnet_kwargs.get('return_scale', False))) if nodes_only \
                        else bool(lprob_kwargs.get('return_scale', False))
                    if not seven:
                        results = results[:5]
                if save_fits:
                    self.Nneighbors[i0 + k] = m
                    self.neighbors.append(np.array(idxs))
                    self.fit_lnprior.append(results[0]); self.fit_lnlike.append(results[1]); self.fit_lnprob.append(results[2])
                    self.fit_Ndim.append(results[3]); self.fit_chi2.append(results[4])
                    if track_scale:
                        self.fit_scale.append(results[5]); self.fit_scale_err.append(results[6])
                if predict is not None:
                    out.append((idxs, m, results, pdfs[k], (lmap[k], levid[k])))
                else:
                    out.append((idxs, m, results))
        return out

    # -- networks.py:782-936 -----------------------------------------------------------------------------
    def fit(self, data, data_err, data_mask, lprob_func=None, nodes_only=False, wt_thresh=1e-3, cdf_thresh=2e-4, lprob_args=None,
            lprob_kwargs=None, track_scale=False, discrete=False, verbose=True):
        self._run(data, data_err, data_mask, lprob_func, nodes_only, wt_thresh, cdf_thresh, lprob_args, lprob_kwargs, track_scale,
                  discrete, True)
        if verbose:
            sys.stderr.write('\rFitting object {0}/{0}\n'.format(len(data))); sys.stderr.flush()

    def _fit(self, data, data_err, data_mask, lprob_func=None, nodes_only=False, wt_thresh=1e-3, cdf_thresh=2e-4, lprob_args=None,
             lprob_kwargs=None, track_scale=False, discrete=False, save_fits=True):
        for r in self._run(data, data_err, data_mask, lprob_func, nodes_only, wt_thresh, cdf_thresh, lprob_args, lprob_kwargs,
                           track_scale, discrete, save_fits):
            yield r

    # -- networks.py:938-1128 ----------------------------------------------------------------------------
    def predict(self, model_labels, model_label_errs, label_dict=None, label_grid=None, logwt=None, kde_args=None, kde_kwargs=None,
                return_gof=False, discrete=False, verbose=True):
        if logwt is None:
            logwt = self.fit_lnprob
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        if self.fit_lnprob is None and logwt is None:
            raise ValueError("Fits have not been computed and weights have not been provided.")
        eng = self._eng()
        N = self.NDATA
        lens = np.array([len(v) for v in logwt], dtype=np.int64)
        W = max(int(lens.max()), 1)
        lw = np.full((N, W), -np.inf); nb = np.zeros((N, W), dtype=np.int64)
        for i in range(N):
            lw[i, :lens[i]] = logwt[i]
            nb[i, :lens[i]] = self.neighbors[i] if not self.nodes_only else 0
        if self.nodes_only:
            # stack the node PDFs by the relative weights of the stored node fits (networks.py:1113-1115)
            node_pdfs = np.ascontiguousarray(self.get_pdfs(model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs,
                                                           False, discrete, verbose))
            G = node_pdfs.shape[1]
            # columns = positions in the stored lists; `match` maps them to the nodes
            pdfs = np.empty((N, G)); lmap = np.empty(N); levid = np.empty(N)
            sel = np.ascontiguousarray(np.broadcast_to(np.arange(W, dtype=np.int32), (N, W)))
            for i in range(N):
                # (one object at a time: every object has its own column -> node map)
                mt = np.zeros(W, dtype=np.int32); mt[:lens[i]] = self.neighbors[i]
                eng.net_stack(np.ascontiguousarray(lw[i:i + 1]), lens[i:i + 1].astype(np.int32), sel[i:i + 1], mt, node_pdfs,
                              pdfs[i:i + 1], lmap[i:i + 1], levid[i:i + 1])
        else:
            ko, G = self._labels(eng, model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs)
            pdfs = np.empty((N, G)); lmap = np.empty(N); levid = np.empty(N)
            eng.knn_predict_logwt(lw, nb, lens, W, ko, pdfs, lmap, levid)
        if verbose:
            sys.stderr.write('\rGenerating PDF {0}/{0}\n'.format(N)); sys.stderr.flush()
        return (pdfs, (lmap, levid)) if return_gof else pdfs

    def _predict(self, model_labels, model_label_errs, node_pdfs=None, label_dict=None, label_grid=None, logwt=None, kde_args=None,
                 kde_kwargs=None):
        if self.nodes_only and node_pdfs is None:
            raise ValueError("Fits were only computed to nodes in the network but the relevant `node_pdfs` are not provided.")
        pdfs, (lmap, levid) = self.predict(model_labels, model_label_errs, label_dict, label_grid, logwt, kde_args, kde_kwargs, True,
                                           False, False)
        for i in range(len(pdfs)):
            yield pdfs[i], (lmap[i], levid[i])

    # -- networks.py:1130-1473 ---------------------------------------------------------------------------
    def fit_predict(self, data, data_err, data_mask, model_labels, model_label_errs, lprob_func=None, nodes_only=False, wt_thresh=1e-3,
                    cdf_thresh=2e-4, label_dict=None, label_grid=None, kde_args=None, kde_kwargs=None, lprob_args=None,
                    lprob_kwargs=None, return_gof=False, track_scale=False, discrete=False, verbose=True, save_fits=True):
        if label_dict is None and label_grid is None:
            raise ValueError("`label_dict` or `label_grid` must be specified.")
        res = self._run(data, data_err, data_mask, lprob_func, nodes_only, wt_thresh, cdf_thresh, lprob_args, lprob_kwargs, track_scale,
                        discrete, save_fits, (model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs))
        pdfs = np.array([r[3] for r in res]); lmap = np.array([r[4][0] for r in res]); levid = np.array([r[4][1] for r in res])
        if verbose:
            sys.stderr.write('\rGenerating PDF {0}/{0}\n'.format(len(data))); sys.stderr.flush()
        return (pdfs, (lmap, levid)) if return_gof else pdfs

    def _fit_predict(self, data, data_err, data_mask, model_labels, model_label_errs, lprob_func=None, node_pdfs=None, wt_thresh=1e-3,
                     cdf_thresh=2e-4, label_dict=None, label_grid=None, kde_args=None, kde_kwargs=None, lprob_args=None,
                     lprob_kwargs=None, track_scale=False, discrete=False, save_fits=True):
        for r in self._run(data, data_err, data_mask, lprob_func, node_pdfs is not None, wt_thresh, cdf_thresh, lprob_args, lprob_kwargs,
                           track_scale, discrete, save_fits, (model_labels, model_label_errs, label_dict, label_grid, kde_args, kde_kwargs)):
            yield r[3], r[4]
