// Inference through a trained network (SOM / GNG): the steps of _Network._fit / _fit_predict / _populate_network between the node
// likelihoods (k_planes with the nodes as noiseless models) and the subset likelihood (k_knn_subset), networks.py:310-333, 880-921,
// 1413-1473.  One wave per object; the object's row of node ln-probabilities sits in LDS.
#pragma once
#include "fz_device.h"

namespace fz {

// Which nodes an object's ln-probabilities select, in the reference's order:
//   use_wt: lnprob > ln(wt_thresh) + max(lnprob), strict (networks.py:887-889), node index ascending (a boolean mask);
//   else:   the CDF rule (networks.py:892-895): nodes by ascending ln-prob (ties by index), the prefix whose running probability
//           exp(l - logsumexp) stays <= 1 - cdf_thresh.  The order is found by rank counting (Nn^2 / 64 compares per wave: this rule
//           is the rare one), the running sum is a wave scan -- it rounds differently from numpy's sequential cumsum, which can move
//           a node only if its cdf lies within rounding of the threshold.
// Outputs: nsel[i], sel[i][0 .. nsel) (column indices), rawlen[i] = summed length of the selected nodes' lists, and max / logsumexp
// over the SELECTED entries (networks.py:330-333).  A row holding a nan selects nothing (numpy: every comparison false).
__global__ __launch_bounds__(256) void k_net_select(const double* __restrict__ lnprob, int64_t N, int Nn, int use_wt, double wt_thresh,
                                                    double cdf_thresh, const int32_t* __restrict__ match, const int64_t* __restrict__ csr_off,
                                                    int32_t* __restrict__ nsel, int32_t* __restrict__ sel, int64_t* __restrict__ rawlen,
                                                    double* __restrict__ lmap, double* __restrict__ levid) {
    extern __shared__ double s_net[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= N) return;
    double* l = s_net + (size_t)wave * 2 * Nn;                  // the row
    int32_t* ord = reinterpret_cast<int32_t*>(l + Nn);          // CDF rule: column by rank | its probability (as two int words) -- 2 Nn ints
    const double* row = lnprob + i * Nn;
    const FastTabs tb = global_tabs();
    double mx = -INFINITY; bool anynan = false;
    for (int c = lane; c < Nn; c += 64) { const double v = row[c]; l[c] = v; anynan |= v != v; mx = fmax(mx, v); }
    mx = wave_max(mx);
    anynan = __any(anynan);
    int32_t* so = sel + i * Nn;
    int n = 0;
    if (anynan) { n = 0; }
    else if (use_wt) {
        const double thr = (wt_thresh > 0.0) ? log(wt_thresh) + mx : -INFINITY;      // wt_thresh = -inf / 0: no clipping
        for (int c0 = 0; c0 < Nn; c0 += 64) {
            const int c = c0 + lane;
            const bool keep = c < Nn && ((wt_thresh > 0.0) ? (l[c] > thr) : true);
            const unsigned long long m = __ballot(keep);
            const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (keep) so[n + pre] = c;
            n += __builtin_popcountll(m);
        }
    } else {
        // logsumexp over the whole row
        double se = 0.0;
        for (int c = lane; c < Nn; c += 64) se += (mx == -INFINITY) ? 0.0 : exp_neg(l[c] - mx, tb);
        se = wave_sum(se);
        const double lse = mx + log_pos(se, tb);
        // rank of every column in ascending (ln-prob, index) order
        for (int c = lane; c < Nn; c += 64) {
            const double v = l[c];
            int r = 0;
            for (int j = 0; j < Nn; ++j) { const double w = l[j]; r += (w < v || (w == v && j < c)) ? 1 : 0; }
            ord[r] = c;
        }
        // running probability in that order; keep the prefix with cdf <= 1 - cdf_thresh
        double carry = 0.0;
        const double lim = 1.0 - cdf_thresh;
        bool open = true;                                            // (the cdf is non-decreasing: the kept set is a prefix)
        for (int r0 = 0; r0 < Nn && open; r0 += 64) {
            const int r = r0 + lane;
            const int c = r < Nn ? ord[r] : 0;
            double p = r < Nn ? ((l[c] == -INFINITY) ? 0.0 : exp_neg(l[c] - lse, tb)) : 0.0;
            // inclusive scan over the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const double q = __shfl_up(p, o, 64); if (lane >= o) p += q; }
            const double cdf = carry + p;
            const bool keep = r < Nn && cdf <= lim;
            const unsigned long long m = __ballot(keep);
            if (keep) so[r] = c;
            n += __builtin_popcountll(m);
            open = (m == ~0ull);
            carry = uniform_d(__shfl(cdf, 63, 64));
        }
    }
    // statistics of the selected entries and the summed list length
    double smx = -INFINITY; long long len = 0;
    for (int s = lane; s < n; s += 64) {
        const int c = so[s];
        smx = fmax(smx, l[c]);
        if (csr_off) { const int nd = match ? match[c] : c; len += (long long)(csr_off[nd + 1] - csr_off[nd]); }
    }
    smx = wave_max(smx);
    double sse = 0.0;
    for (int s = lane; s < n; s += 64) sse += (smx == -INFINITY) ? 0.0 : exp_neg(l[so[s]] - smx, tb);
    sse = wave_sum(sse);
    len = (long long)wave_sum((double)len);
    if (lane == 0) {
        nsel[i] = n;
        if (rawlen) rawlen[i] = len;
        if (lmap) lmap[i] = smx;                                       // (no entry: -inf, as max over nothing would be undefined)
        if (levid) levid[i] = (n > 0) ? smx + log_pos(sse, tb) : -INFINITY;
    }
}

// idx[i][0 .. W) = the node lists of object i's selected nodes, concatenated in order (networks.py:913-918), padded with the row's
// first entry: fz_knn_fit_predict removes repeats in first-appearance order, which is pandas.unique (networks.py:919)
__global__ __launch_bounds__(256) void k_net_table(const int32_t* __restrict__ nsel, const int32_t* __restrict__ sel, int64_t N, int Nn,
                                                   const int32_t* __restrict__ match, const int64_t* __restrict__ csr_off,
                                                   const int64_t* __restrict__ csr_items, int64_t W, int64_t* __restrict__ idx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= N) return;
    int64_t* out = idx + i * W;
    int64_t pos = 0, first = 0;
    const int n = nsel[i];
    for (int s = 0; s < n; ++s) {
        const int c = sel[i * Nn + s];
        const int nd = match ? match[c] : c;
        const int64_t a = csr_off[nd], b = csr_off[nd + 1];
        if (pos == 0 && b > a) first = csr_items[a];
        for (int64_t k = a + lane; k < b; k += 64) if (pos + (k - a) < W) out[pos + (k - a)] = csr_items[k];
        pos += b - a;
    }
    for (int64_t k = pos + lane; k < W; k += 64) out[k] = first;
}

// out[i][s] = plane[i][sel[i][s]] for s < nsel[i], pad beyond (8-byte elements: the node results of nodes_only fits, networks.py:907-909)
__global__ void k_net_gather(const unsigned long long* __restrict__ plane, const int32_t* __restrict__ nsel, const int32_t* __restrict__ sel,
                             int64_t N, int Nn, int W, unsigned long long pad, unsigned long long* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * W) return;
    const int64_t i = e / W; const int s = (int)(e - i * W);
    out[e] = (s < nsel[i]) ? plane[i * Nn + sel[i * Nn + s]] : pad;
}

// nodes_only prediction (networks.py:1463-1470): pdf = wt @ node_pdfs[idxs], wt = exp(lnprob - logsumexp) over the selected nodes,
// then pdf /= pdf.sum(); lmap / levid over the selected ln-probabilities.  A wave per object, lanes along the grid.
__global__ __launch_bounds__(256) void k_net_stack(const double* __restrict__ lnprob, const int32_t* __restrict__ nsel, const int32_t* __restrict__ sel,
                                                   int64_t N, int Nn, const int32_t* __restrict__ match, const double* __restrict__ node_pdfs,
                                                   int G, double* __restrict__ pdfs, double* __restrict__ lmap, double* __restrict__ levid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= N) return;
    const FastTabs tb = global_tabs();
    const int n = nsel[i];
    const double* row = lnprob + i * Nn;
    const int32_t* so = sel + i * Nn;
    double mx = -INFINITY;
    for (int s = lane; s < n; s += 64) mx = fmax(mx, row[so[s]]);
    mx = wave_max(mx);
    double se = 0.0;
    for (int s = lane; s < n; s += 64) se += exp_neg(row[so[s]] - mx, tb);
    se = wave_sum(se);
    const double le = mx + log_pos(se, tb);
    double* out = pdfs + i * G;
    double tot = 0.0;
    for (int t0 = 0; t0 < G; t0 += 64) {
        const int t = t0 + lane;
        double acc = 0.0;
        for (int s = 0; s < n; ++s) {                               // in the reference's order (np.dot over the selected rows)
            const int c = so[s];
            const double w = exp_neg(row[c] - le, tb);
            if (t < G) acc = fma(w, node_pdfs[(int64_t)(match ? match[c] : c) * G + t], acc);
        }
        if (t < G) out[t] = acc;
        tot += (t < G) ? acc : 0.0;
    }
    tot = wave_sum(tot);
    for (int t = lane; t < G; t += 64) out[t] = out[t] / tot;       // pdf /= pdf.sum()  (no node selected: 0 / 0 = nan, as numpy)
    if (lane == 0) { if (lmap) lmap[i] = mx; if (levid) levid[i] = le; }
}

}  // namespace fz
