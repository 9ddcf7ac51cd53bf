// Kernels of the brute-force path.  gfx950, wave64.
//
//   k_planes : BruteForce._fit (bruteforce.py:191-203)             -> (N,M) planes
//   k_fused  : BruteForce._fit_predict, save_fits=False (bruteforce.py:602-631):
//              ONE pass over the models per object -- likelihood, running
//              max / sum-exp, and a candidate list of the models that can still
//              pass the weight threshold; then weights, threshold, kernel stack,
//              normalise from the (short) candidate list.
//   k_stats + k_kde : the same result in two passes over the models (used for
//              materialised ln-weight planes -- predict(), mode C -- and as the
//              fallback when the candidate workspace does not fit).
//
// Work decomposition.  A lane owns one MODEL (its B fluxes/variances live in
// VGPRs, loaded with coalesced 512-B-per-band reads from the SoA model arrays);
// objects are wave-uniform (scalar loads -> SGPR operands).  Every wave owns TW
// objects and streams all M models past them, so all per-object reductions are
// private to a wave: no barriers, no cross-wave atomics.  KDE accumulators live in
// LDS (one 5.6-6 KB row per object for the 701-point redshift grid).
#pragma once
#include "fz_device.h"

namespace fz {

// ---- a source that reads ln-weights from a materialised (N,M) plane ----------
struct PlaneSrc {
    const double* p; int64_t ld;
    struct MR {};
    struct OR { const double* row; };
    __device__ __forceinline__ void load_model(int64_t, MR&) const {}
    __device__ __forceinline__ void load_obj(int64_t i, OR& o) const { o.row = p + i * ld; }
    __device__ __forceinline__ double lnl(const OR& o, const MR&, int64_t j, bool valid) const {
        return valid ? o.row[j] : -INFINITY;
    }
};
template <int BT, int MODE, int VAR>
struct PhotSrc : Phot<BT, MODE, VAR> {
    using P = Phot<BT, MODE, VAR>;
    __device__ __forceinline__ double lnl(const typename P::OR& o, const typename P::MR& m,
                                          int64_t, bool valid) const {
        const double l = P::eval(o, m).lnl;       // pad lanes hold benign data; no divergent branch
        return valid ? l : -INFINITY;
    }
};

// ---- materialising fit ------------------------------------------------------
template <class PH, int TO>
__global__ __launch_bounds__(256) void k_planes(PH ph, int64_t N, int64_t M, double* __restrict__ lnl,
                                                double* __restrict__ chi2, int64_t* __restrict__ ndim,
                                                double* __restrict__ scale, double* __restrict__ serr) {
    const int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const bool valid = j < M;
    typename PH::MR m;
    ph.load_model(j, m);                       // j < Mp (Mp is a multiple of 256)
    const int64_t i0 = (int64_t)blockIdx.x * TO;
    for (int o = 0; o < TO; ++o) {
        const int64_t i = i0 + o;
        if (i >= N) break;
        typename PH::OR ob;
        ph.load_obj(i, ob);
        PairOut r = ph.eval(ob, m);
        if (valid) {
            const int64_t k = i * M + j;
            if (lnl) lnl[k] = r.lnl;
            if (chi2) chi2[k] = r.chi2;
            if (ndim) ndim[k] = r.ndim;
            if (scale) scale[k] = r.scale;
            if (serr) serr[k] = sqrt(1.0 / r.shape);      // pdf.py:232
        }
    }
}

// ---- KDE tables ------------------------------------------------------------------
struct KdeView {
    int64_t G;
    int kmode;             // KDE_HIST / KDE_DICT / KDE_GRID
    // dictionary path (pdf.py:599-620), per model (padded to Mp)
    const int32_t* pos;    // y_idx
    const int32_t* cls;    // y_std_idx
    const double* norm;    // edge-truncated kernel mass (pdf.py:613-617) / in-window sum (pdf.py:521)
    const int64_t* widths; const int64_t* offsets; const double* kern;
    int32_t w0; int64_t koff0;            // single sigma class: width and table offset
    // direct path (pdf.py:499-502, 519-524), per model
    const double* ly; const double* lstd; const int32_t* lo; const int32_t* hi;
    const double* grid;
    int acc_stride;                       // doubles of LDS per object
};
// HIST: every label shares one dictionary kernel -> accumulate w/norm at the label's
// grid index (one LDS atomic per selected model), convolve once at the end.
// DICT / GRID: add each selected model's window, a wave per model.
enum { KDE_HIST = 0, KDE_DICT = 1, KDE_GRID = 2 };

// add the selected lanes' kernels into `row`.  w: the lane's weight; jm: its model.
__device__ __forceinline__ void kde_scatter(const KdeView& kv, double* row, bool sel, double w, int64_t jm,
                                            int lane) {
    if (kv.kmode == KDE_HIST) {
        if (sel) unsafeAtomicAdd(&row[kv.pos[jm] + kv.w0], w / kv.norm[jm]);
        return;
    }
    unsigned long long mask = __ballot(sel);
    while (mask) {
        const int sl = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int64_t js = __shfl(jm, sl, 64);
        const double ws = __shfl(w, sl, 64);
        if (kv.kmode == KDE_DICT) {
            const int p = kv.pos[js], c = kv.cls[js];
            const int wd = (int)kv.widths[c];
            const double wn = ws / kv.norm[js];
            const int lo = max(p - wd, 0), hi = min(p + wd + 1, (int)kv.G);
            const double* kr = kv.kern + kv.offsets[c] + (lo - (p - wd)) - lo;
            for (int t = lo + lane; t < hi; t += 64) row[t] = fma(wn, kr[t], row[t]);
        } else {
            const int lo = kv.lo[js], hi = kv.hi[js];
            const double nrm = kv.norm[js];
            if (nrm != 0.0) {                                   // pdf.py:523
                const double mu = kv.ly[js], sd = kv.lstd[js];
                const double wn = ws / nrm;
                const double gn = 2.5066282746310002 * sd;      // sqrt(2 pi) * std
                for (int t = lo + lane; t < hi; t += 64) {
                    const double z = (kv.grid[t] - mu) / sd;
                    row[t] = fma(wn, exp(-0.5 * (z * z)) / gn, row[t]);
                }
            }
        }
    }
}

// (convolve,) normalise, write one PDF row
__device__ __forceinline__ void kde_finalize(const KdeView& kv, const double* row, bool ok, int normalize,
                                             double* out, int lane) {
    const int G = (int)kv.G;
    if (!ok) { for (int t = lane; t < G; t += 64) out[t] = NAN; return; }
    double tot = 0.0;
    if (kv.kmode == KDE_HIST) {
        const int w2 = 2 * kv.w0;
        const double* kr = kv.kern + kv.koff0;
        for (int t = lane; t < G; t += 64) {
            double v = 0.0;
            for (int h = 0; h <= w2; ++h) v = fma(row[t + h], kr[w2 - h], v);
            out[t] = v;
            tot += v;
        }
    } else {
        for (int t = lane; t < G; t += 64) { const double v = row[t]; out[t] = v; tot += v; }
    }
    if (normalize) {
        tot = wave_sum(tot);
        for (int t = lane; t < G; t += 64) out[t] = out[t] / tot;       // pdf /= pdf.sum()
    }
}

// ---- pass 1: per-object max and logsumexp ------------------------------------
// linear=1: rows are linear weights; only the max is produced (np.max: NaN wins).
template <class SRC, int TW>
__global__ __launch_bounds__(256) void k_stats(SRC src, int64_t N, int64_t M, int linear,
                                               double* __restrict__ lmap, double* __restrict__ levid) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * TW;
    if (i0 >= N) return;
    typename SRC::OR ob[TW];
    MS st[TW];
    unsigned firstnan = 0, anynan = 0;
#pragma unroll
    for (int o = 0; o < TW; ++o) {
        src.load_obj(i0 + o < N ? i0 + o : N - 1, ob[o]);
        ms_init(st[o]);
    }
    for (int64_t jb = 0; jb < M; jb += 64) {
        const int64_t j = jb + lane;
        const bool valid = j < M;
        typename SRC::MR m;
        src.load_model(j, m);
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            const double l = src.lnl(ob[o], m, j, valid);
            if (l != l) { anynan |= 1u << o; if (j == 0) firstnan |= 1u << o; }
            if (linear) { if (l > st[o].m) st[o].m = l; }
            else ms_push(st[o], l);
        }
    }
#pragma unroll
    for (int o = 0; o < TW; ++o) {
        const int64_t i = i0 + o;
        const bool fn = __any((firstnan >> o) & 1u);
        const bool an = __any((anynan >> o) & 1u);
        if (linear) {
            const double mx = wave_max(st[o].m);
            if (lane == 0 && i < N) lmap[i] = an ? (double)NAN : mx;
        } else {
            MS t = wave_ms(st[o]);
            if (lane == 0 && i < N) {
                lmap[i] = fn ? (double)NAN : t.m;           // builtin max: NaN only if first
                if (levid) levid[i] = an ? (double)NAN : (t.m == INFINITY ? (double)INFINITY : t.m + log(t.s));
            }
        }
    }
}

// ---- pass 2: threshold + weighted kernel stack ---------------------------------
template <class SRC, int TW>
__global__ __launch_bounds__(256) void k_kde(SRC src, KdeView kv, int64_t N, int64_t M, int linear,
                                             const double* __restrict__ lmap,
                                             const double* __restrict__ levid, double wt_thresh,
                                             int normalize, double* __restrict__ pdfs) {
    extern __shared__ double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * TW;
    if (i0 >= N) return;
    double* acc = smem + (size_t)wave * TW * kv.acc_stride;
    for (int k = lane; k < TW * kv.acc_stride; k += 64) acc[k] = 0.0;

    typename SRC::OR ob[TW];
    double le[TW], thr[TW], lthr[TW];
    bool ok[TW];
#pragma unroll
    for (int o = 0; o < TW; ++o) {
        const int64_t i = i0 + o < N ? i0 + o : N - 1;
        src.load_obj(i, ob[o]);
        const double lm = lmap[i];
        if (linear) {                       // rows are weights: thr = wt_thresh*max(w)
            le[o] = 0.0;
            ok[o] = (i0 + o < N);
            thr[o] = wt_thresh * lm;        // NaN max -> nothing passes (pdf.py:510)
            lthr[o] = -INFINITY;
        } else {
            le[o] = levid[i];
            ok[o] = (i0 + o < N) && (le[o] - le[o] == 0.0);   // finite evidence
            thr[o] = wt_thresh * exp_neg(lm - le[o]);         // wt_thresh * max(wt)
            lthr[o] = (wt_thresh > 0.0) ? lm + log(wt_thresh) - 1e-3 : -INFINITY;
        }
    }
    for (int64_t jb = 0; jb < M; jb += 64) {
        const int64_t j = jb + lane;
        const bool valid = j < M;
        typename SRC::MR m;
        src.load_model(j, m);
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            if (!ok[o]) continue;                                   // wave-uniform
            const double l = src.lnl(ob[o], m, j, valid);
            const bool cand = valid && (linear ? true : (l > lthr[o]));
            if (!__any(cand)) continue;                             // wave-uniform
            const double w = linear ? l : exp_neg(l - le[o]);
            const bool sel = cand && (w > thr[o]);                  // strict, pdf.py:510/591
            kde_scatter(kv, acc + o * kv.acc_stride, sel, w, j, lane);
        }
    }
#pragma unroll
    for (int o = 0; o < TW; ++o) {
        const int64_t i = i0 + o;
        if (i >= N) break;
        kde_finalize(kv, acc + o * kv.acc_stride, ok[o], normalize, pdfs + i * kv.G, lane);
    }
}

// ---- single pass: likelihood + softmax statistics + candidates -> PDF -----------
struct Cand { double lnl; int32_t j; int32_t pad; };      // 16 B, one dwordx4 store

// Every wave walks its object groups (TW objects each) with a grid stride and owns a
// private candidate buffer of TW x cap entries (cap = M: it can never overflow).
// A pair is recorded when its lnl is within the weight threshold of the best lnl
// seen SO FAR (per lane, tightened every 16 steps with the wave-wide best): a
// superset of the pairs with wt > wt_thresh * max(wt), because the running best only
// grows.  The exact test (pdf.py:510 / 591, strict >) is applied afterwards with the
// final max and evidence.
template <class SRC, int TW>
__global__ __launch_bounds__(256) void k_fused(SRC src, KdeView kv, int64_t N, int64_t M, double wt_thresh,
                                               int normalize, Cand* __restrict__ cand, int64_t cap,
                                               double* __restrict__ lmap, double* __restrict__ levid,
                                               double* __restrict__ pdfs) {
    extern __shared__ double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    double* row = smem + (size_t)wave * kv.acc_stride;            // one object at a time
    Cand* buf = cand + (size_t)gw * TW * cap;
    const double lt = (wt_thresh > 0.0) ? log(wt_thresh) - 1e-3 : -INFINITY;

    for (int64_t g = gw; g * TW < N; g += nwaves) {
        const int64_t i0 = g * TW;
        typename SRC::OR ob[TW];
        MS st[TW];
        double best[TW];                 // wave-wide best lnl (refreshed every 16 steps)
        int cnt[TW];
        unsigned firstnan = 0, anynan = 0;
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            src.load_obj(i0 + o < N ? i0 + o : N - 1, ob[o]);
            ms_init(st[o]);
            best[o] = -INFINITY;
            cnt[o] = 0;
        }
        int tick = 0;
        for (int64_t jb = 0; jb < M; jb += 64) {
            const int64_t j = jb + lane;
            const bool valid = j < M;
            typename SRC::MR m;
            src.load_model(j, m);
#pragma unroll
            for (int o = 0; o < TW; ++o) {
                const double l = src.lnl(ob[o], m, j, valid);
                if (l != l) { anynan |= 1u << o; if (j == 0) firstnan |= 1u << o; }
                ms_push(st[o], l);
                const bool c = l > fmax(st[o].m, best[o]) + lt;       // false for nan / -inf
                const unsigned long long mask = __ballot(c);
                if (mask) {                                           // wave-uniform
                    const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    if (c) { Cand e; e.lnl = l; e.j = (int32_t)j; e.pad = 0; buf[(size_t)o * cap + cnt[o] + pre] = e; }
                    cnt[o] += __builtin_popcountll(mask);
                }
            }
            if ((++tick & 15) == 0) {
#pragma unroll
                for (int o = 0; o < TW; ++o) best[o] = wave_max(st[o].m);
            }
        }
        // the candidate entries were written by other lanes of this wave: make them
        // visible to this CU's loads (stores drained, vector L1 invalidated)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            const int64_t i = i0 + o;
            if (i >= N) break;
            const bool fn = __any((firstnan >> o) & 1u);
            const bool an = __any((anynan >> o) & 1u);
            const MS t = wave_ms(st[o]);
            const double lm = fn ? (double)NAN : t.m;
            const double le = an ? (double)NAN : (t.m == INFINITY ? (double)INFINITY : t.m + log(t.s));
            if (lane == 0) { if (lmap) lmap[i] = lm; if (levid) levid[i] = le; }
            const bool ok = (le - le == 0.0);
            if (ok) {
                for (int k = lane; k < kv.acc_stride; k += 64) row[k] = 0.0;
                const double thr = wt_thresh * exp_neg(t.m - le);
                const Cand* cb = buf + (size_t)o * cap;
                const int n = __builtin_amdgcn_readfirstlane(cnt[o]);
                for (int c0 = 0; c0 < n; c0 += 64) {
                    const int k = c0 + lane;
                    const bool in = k < n;
                    const Cand e = cb[in ? k : 0];
                    const double w = exp_neg(e.lnl - le);
                    kde_scatter(kv, row, in && (w > thr), w, e.j, lane);
                }
            }
            kde_finalize(kv, row, ok, normalize, pdfs + i * kv.G, lane);
        }
    }
}

}  // namespace fz
