// The reference's CDF thresholding (kde_kwargs wt_thresh=None; pdf.py:513-516, 593-597):
//
//     idx_sort = argsort(y_wt); y_cdf = cumsum(y_wt[idx_sort]); y_cdf /= y_cdf[-1]
//     sel_arr  = idx_sort[y_cdf <= 1 - cdf_thresh]
//
// i.e. the ASCENDING prefix whose cumulative weight stays within 1 - cdf_thresh: every
// kernel is stacked EXCEPT the few largest weights, the minimal top-K whose sum reaches
// cdf_thresh of the total (K >= 1; K <= cdf_thresh * Ny + 1).  That is a reference quirk
// (it drops the most probable models), reproduced here as it is: find the top-K by
// repeated arg-max (K is 1 for any peaked posterior), stack everything else unthresholded.
// The arg-max walks the row in the strict order (weight descending, column ascending): the
// last (weight, column) taken IS the exclusion set -- everything at or before it in that
// order -- so K is unbounded and no list is kept (a flat posterior over 1e6 models at the
// default cdf_thresh drops 200 of them; K rounds of one pass over the row each).
// A broad posterior with a LARGE cdf_thresh (0.5 over 1e5 models: K ~ n / 2) would make that
// K n / 64 steps per object: after FZ_CDF_SERIAL rounds the kernel switches to a radix
// selection on the weights' bit patterns (non-negative doubles order like their bits): eight
// passes over the row, each with a 256-bucket histogram of weight sums in LDS, find the
// weight value at which the excluded sum crosses the bar; ties at that value are excluded in
// column order, as the arg-max rounds would.
// One object per wave, rows of ln-weights or linear weights read from a plane; optional
// neighbour-table indirection for the k-NN variant.
#pragma once
#include "fz_kernels.h"

#ifndef FZ_CDF_SERIAL
#define FZ_CDF_SERIAL 64          // arg-max rounds before the radix selection takes over
#endif

namespace fz {

static __global__ __launch_bounds__(256) void k_kde_cdf(const KdeView* __restrict__ kvp, int acc_stride, int64_t N, int L,
                                                 int M, const double* __restrict__ rows,
                                                 const int64_t* __restrict__ nbr, const int64_t* __restrict__ nnb,
                                                 int is_log, double cdf_thresh, int normalize,
                                                 double* __restrict__ pdfs, double* __restrict__ lmap,
                                                 double* __restrict__ levid, int* __restrict__ errflag) {
    extern __shared__ double smem[];                 // per wave: row[max(acc_stride, 512)] (the selection's histogram shares it)
    const FastTabs tb = global_tabs();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (i >= N) return;
    double* row = smem + (size_t)wave * (acc_stride > 512 ? acc_stride : 512);
    const KdeView kv = *kvp;
    const double* in = rows + i * (int64_t)L;
    const int n = nnb ? (int)nnb[i] : L;
    if (n < 0 || n > L) { if (lane == 0) atomicExch(errflag, 1); return; }

    // ---- max / evidence (ln-weights), as bruteforce.py:359-360 ----
    double le = 0.0;
    bool ok = true;
    if (is_log) {
        MS st; ms_init(st);
        bool isnan0 = false, anynan = false;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const double l = j < n ? in[j] : -INFINITY;
            if (l != l) { anynan = true; if (j == 0) isnan0 = true; }
            ms_push(st, l, tb);
        }
        const bool fn = __any(isnan0), an = __any(anynan);
        const double mx = wave_max(st.m);
        const double ss = wave_sum(st.s * exp_neg(st.m - mx, tb));
        le = an ? (double)NAN : (mx == INFINITY ? (double)INFINITY : mx + log(ss));
        if (lane == 0) { if (lmap) lmap[i] = fn ? (double)NAN : mx; if (levid) levid[i] = le; }
        ok = (le - le == 0.0);
    }
    // exp_neg clamps at -700 (returns ~1e-304); here an underflowed weight must be an exact 0,
    // because the rule can leave ONLY such weights selected (pdf = 0/0 = nan in the reference)
    auto weight = [&](int j) -> double {
        if (!is_log) return in[j];
        const double a = in[j] - le;
        return (a < -700.0) ? 0.0 : exp_neg(a, tb);
    };

    // ---- total weight and the minimal top-K with sum >= cdf_thresh * total ----
    int K = 0;
    bool bad = false;
    double vlast = INFINITY; int jlast = -1;             // the last (weight, column) excluded; wave-uniform
    if (ok) {
        double tot = 0.0;
        bool wnan = false;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            if (j < n) { const double w = weight(j); tot += w; if (w != w) wnan = true; }
        }
        tot = wave_sum(tot);
        if (__any(wnan)) ok = false;                     // nan cdf: nothing is selected (zeros)
        double excluded = 0.0;
        const double lim = (1.0 - cdf_thresh) * tot;
        while (ok && K < n && K < FZ_CDF_SERIAL && (tot - excluded) > lim) {
            double best = -INFINITY; int bj = 0x7fffffff;
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                if (j < n) {
                    const double w = weight(j);
                    const bool ex = (w > vlast) || (w == vlast && j <= jlast);
                    if (!ex && w > best) { best = w; bj = j; }     // (a lane's columns ascend: the first of equal weights stays)
                }
            }
            const double wbest = wave_max(best);
            int cand = (best == wbest) ? bj : 0x7fffffff;
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) cand = min(cand, __shfl_xor(cand, s, 64));
            if (cand == 0x7fffffff) break;               // nothing left (all -inf / empty)
            vlast = wbest; jlast = cand;
            ++K;
            excluded += wbest;
        }
        if (ok && K == FZ_CDF_SERIAL && K < n && (tot - excluded) > lim) {
            // radix selection over the whole row (from scratch: the rounds above only showed that K is large)
            double* hsum = row; double* hcnt = row + 256;
            unsigned long long prefix = 0ull, mask = 0ull;
            double above = 0.0, ties = 0.0;                        // summed weight strictly above the current prefix range; count at v*
            for (int shift = 56; shift >= 0; shift -= 8) {
                for (int k = lane; k < 512; k += 64) row[k] = 0.0;
                for (int j0 = 0; j0 < n; j0 += 64) {
                    const int j = j0 + lane;
                    if (j < n) {
                        const double w = weight(j);
                        const unsigned long long b = (unsigned long long)__double_as_longlong(w);
                        if ((b & mask) == prefix) { const int d = (int)((b >> shift) & 255ull); unsafeAtomicAdd(&hsum[d], w); unsafeAtomicAdd(&hcnt[d], 1.0); }
                    }
                }
                int dstar = 0; double run = above;
                for (int d = 255; d >= 0; --d) {                   // (every lane walks the same buckets: wave-uniform)
                    const double sd = hsum[d];
                    if (hcnt[d] > 0.0 && (tot - (run + sd)) <= lim) { dstar = d; break; }
                    run += sd;
                }
                above = uniform_d(run); ties = uniform_d(hcnt[dstar]);
                prefix |= (unsigned long long)dstar << shift; mask |= 255ull << shift;
            }
            const double vstar = __longlong_as_double((long long)prefix);
            // ties at v* leave in column order until the bar is met, as the arg-max rounds take them
            double e = above; long long t = 0;
            if (vstar > 0.0) {
                t = (long long)ceil((tot - above - lim) / vstar) - 1;
                if (t < 0) t = 0;
                if (t > (long long)ties) t = (long long)ties;
                e = above + (double)t * vstar;
                while ((tot - e) > lim && t < (long long)ties) { e += vstar; ++t; }
            } else t = (long long)ties;
            if (t < 1) t = 1;
            long long seen = 0; int jl = -1;
            for (int j0 = 0; j0 < n && jl < 0; j0 += 64) {
                const int j = j0 + lane;
                const bool tie = j < n && weight(j) == vstar;
                const unsigned long long m = __ballot(tie);
                const int c = __builtin_popcountll(m);
                if (seen + c >= t) {                               // the t-th tie sits in this group
                    const int want = (int)(t - seen);
                    const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    const unsigned long long hit = __ballot(tie && pre == want - 1);
                    jl = j0 + __builtin_ctzll(hit);
                }
                seen += c;
            }
            vlast = vstar; jlast = jl < 0 ? n - 1 : jl;
            K = n;                                                 // (only its being > 0 matters below)
        }
    }
    // ---- stack every other kernel, unthresholded (pdf.py:599-620 / 519-524) ----
    for (int t = lane; t < acc_stride; t += 64) row[t] = 0.0;
    if (ok) {
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const double w = (j < n) ? weight(j) : 0.0;
            bool sel = (j < n) && !((w > vlast) || (w == vlast && j <= jlast));
            int64_t jm = 0;
            if (j < n) { jm = nbr ? nbr[i * (int64_t)L + j] : j; if (jm < 0 || jm >= M) { bad = true; jm = 0; sel = false; } }
            kde_scatter(kv, row, sel, w, jm, lane);
        }
    }
    if (__any(bad)) { if (lane == 0) atomicExch(errflag, 1); return; }
    // is_log: a non-finite evidence poisons the row (pdf = 0/0) like the reference; linear
    // weights with a nan give zeros
    kde_finalize(kv, row, is_log ? ok : true, normalize, pdfs + i * kv.G, lane);
}

}  // namespace fz
