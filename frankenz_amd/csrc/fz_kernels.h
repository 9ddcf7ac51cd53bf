// Kernels of the brute-force path.  gfx950, wave64.
//
//   k_planes : BruteForce._fit            (bruteforce.py:191-203)  -> (N,M) planes
//   k_stats  : lmap = max, levid = logsumexp per object (bruteforce.py:359, 619)
//   k_kde    : wt = exp(lnprob - levid), threshold, weighted kernel stack,
//              normalise (bruteforce.py:360-370, 620-629; pdf.py:489-526, 585-622)
//
// Work decomposition.  A lane owns one MODEL (its B fluxes/variances live in
// VGPRs, loaded with coalesced 512-B-per-band reads from the SoA model arrays);
// objects are wave-uniform (scalar loads -> SGPR operands).  k_stats / k_kde give
// every wave its own TW objects and stream all M models past them, so all
// per-object reductions are private to a wave: no barriers, no cross-wave
// atomics.  The KDE accumulators of a wave's objects live in LDS.
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
template <int BT, int MODE, bool MASKED>
struct PhotSrc : Phot<BT, MODE, MASKED> {
    using P = Phot<BT, MODE, MASKED>;
    __device__ __forceinline__ double lnl(const typename P::OR& o, const typename P::MR& m,
                                          int64_t, bool valid) const {
        return valid ? P::eval(o, m).lnl : -INFINITY;
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
            double l = src.lnl(ob[o], m, j, valid);
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
            double mx = st[o].m;
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) mx = fmax(mx, __shfl_xor(mx, s, 64));
            if (lane == 0 && i < N) lmap[i] = an ? (double)NAN : mx;
        } else {
            MS t = wave_ms(st[o]);
            if (lane == 0 && i < N) {
                lmap[i] = fn ? (double)NAN : t.m;           // builtin max: NaN only if first
                if (levid) levid[i] = (t.m == INFINITY && !an) ? (double)INFINITY : t.m + log(t.s);
            }
        }
    }
}

// ---- pass 2: threshold + weighted kernel stack ---------------------------------
struct KdeView {
    int64_t G;
    // dictionary path (pdf.py:599-620), per model (padded to Mp)
    const int32_t* pos;    // y_idx
    const int32_t* cls;    // y_std_idx
    const double* norm;    // edge-truncated kernel mass (pdf.py:613-617)
    const int64_t* widths; const int64_t* offsets; const double* kern;
    int32_t w0; int64_t koff0;            // single sigma class: width and table offset
    // direct path (pdf.py:499-502, 519-524), per model
    const double* ly; const double* lstd; const int32_t* lo; const int32_t* hi;
    const double* grid;
    int acc_stride;                       // doubles of LDS per object
};
enum { KDE_HIST = 0, KDE_DICT = 1, KDE_GRID = 2 };

template <class SRC, int TW, int KMODE>
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
            thr[o] = wt_thresh * exp(lm - le[o]);             // wt_thresh * max(wt)
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
            const double w = linear ? l : exp(l - le[o]);
            const bool sel = cand && (w > thr[o]);                  // strict, pdf.py:510/591
            double* row = acc + o * kv.acc_stride;
            if (KMODE == KDE_HIST) {
                if (sel) unsafeAtomicAdd(&row[kv.pos[j] + kv.w0], w / kv.norm[j]);
            } else {
                unsigned long long mask = __ballot(sel);
                while (mask) {
                    const int sl = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int64_t js = jb + sl;                     // wave-uniform
                    if (KMODE == KDE_DICT) {
                        const int p = kv.pos[js], c = kv.cls[js];
                        const int wd = (int)kv.widths[c];
                        const double wn = __shfl(w, sl, 64) / kv.norm[js];
                        const int lo = max(p - wd, 0), hi = min(p + wd + 1, (int)kv.G);
                        const double* kr = kv.kern + kv.offsets[c] + (lo - (p - wd)) - lo;
                        for (int t = lo + lane; t < hi; t += 64) row[t] = fma(wn, kr[t], row[t]);
                    } else {
                        const int lo = kv.lo[js], hi = kv.hi[js];
                        const double nrm = kv.norm[js];
                        if (nrm != 0.0) {                           // pdf.py:523
                            const double mu = kv.ly[js], sd = kv.lstd[js];
                            const double wn = __shfl(w, sl, 64) / nrm;
                            const double gn = 2.5066282746310002 * sd;   // sqrt(2 pi) * std
                            for (int t = lo + lane; t < hi; t += 64) {
                                const double z = (kv.grid[t] - mu) / sd;
                                row[t] = fma(wn, exp(-0.5 * (z * z)) / gn, row[t]);
                            }
                        }
                    }
                }
            }
        }
    }
    // finalise: (convolve,) normalise, write
#pragma unroll
    for (int o = 0; o < TW; ++o) {
        const int64_t i = i0 + o;
        if (i >= N) break;
        double* out = pdfs + i * kv.G;
        const double* row = acc + o * kv.acc_stride;
        const int G = (int)kv.G;
        if (!ok[o]) { for (int t = lane; t < G; t += 64) out[t] = NAN; continue; }
        double tot = 0.0;
        if (KMODE == KDE_HIST) {
            const int w2 = 2 * kv.w0;
            const double* kr = kv.kern + kv.koff0;
            for (int t = lane; t < G; t += 64) {
                double v = 0.0;
                for (int h = 0; h <= w2; ++h) v = fma(row[t + h], kr[w2 - h], v);
                out[t] = v;
                tot += v;
            }
        } else {
            for (int t = lane; t < G; t += 64) { double v = row[t]; out[t] = v; tot += v; }
        }
        if (normalize) {
            tot = wave_sum(tot);
            for (int t = lane; t < G; t += 64) out[t] = out[t] / tot;
        }
    }
}

}  // namespace fz
