// List-free fused fit_predict for BROAD likelihoods (single dictionary kernel, weight-space body).
//
// k_fused hands the models within the weight threshold of the running best to the PDF stage through
// per-object candidate lists in HBM (16 B out and 16 B back per recorded pair).  At SDSS-depth noise 7 %
// of all (object, model) pairs are recorded and the lists are a side show; on faint data (the reference's
// own mock: 41 %; bench.py --noise-scale 3 / 10: 53 % / 95 %) they ARE the kernel's HBM traffic, 3-4 TB/s.
// Here nothing is recorded.  The dimensionality-prior likelihood chi2^(k/2) e^(-chi2/2) / C is unimodal in
// chi2 with its mode at chi2 = k, so an object's exact maximum ln-like is decided by two numbers: the
// largest chi2 at or below the mode and the smallest above it.
//   pass 1 (k_nl_max):  chi2 of every pair, the two-sided tracking, -> the chi2 of the best model per object;
//   pass 2 (k_nl_main): chi2 again; every pair's weight relative to the (now known, exact) maximum in fp64 -- integer
//                       powers by multiplication, the half power by a Newton-refined v_rsq_f64, one exp: no log; the
//                       reference's strict rule w > wt_thresh * max(w) (max(w) is 1 by construction) decides what goes
//                       straight into the object's LDS histogram (one ds_add_f64; pdf.py:585-622 with the per-index
//                       kernel mass and the one convolution of kde_finalize); every weight enters the evidence sum.
// All fp64 (the evidence too: unlike k_fused's weight-space body there is no fp32 remainder).  2 x the chi2 arithmetic,
// ~0 B/eval of HBM traffic.
// (reference: bruteforce.py:602-631 -> pdf.py:27-100, 585-622)
#pragma once
#include "fz_kernels.h"

namespace fz {

// LDS-DMA copy of one model tile (records | index words), as in k_fused
template <class SRC, int TILE, int NT, bool TAGS>
__device__ __forceinline__ void nl_stage_tile(const SRC& src, const int32_t* posw, int tile, double* dst, int tid, int wave) {
    auto seg = [&](const void* base, int nchunks, double* d) {
        const uint32_t slot = (uint32_t)tid * 16u;
#pragma unroll
        for (int q = 0; q < (nchunks + NT - 1) / NT; ++q) {
            if (tid + q * NT < nchunks)
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(reinterpret_cast<const char*>(base) + (size_t)q * NT * 16 + slot),
                                                 (lds_void*)(d + 2 * (q * NT + wave * 64)), 16, 0, 0);
        }
    };
    const double* rec = (SRC::LMODE == 0) ? src.mv.rec0 : src.mv.rec1;
    seg(rec + (int64_t)tile * (TILE * SRC::RW), SRC::RW * TILE / 2, dst);
    if constexpr (TAGS) seg(posw + (int64_t)tile * TILE, TILE / 4, dst + SRC::RW * TILE);
}


// ---- which form for this launch?  The share of (object, model) pairs within the weight threshold, measured on
// a sample of the launch's objects (one wave per sampled object, two sweeps over the models: best chi2 by the
// two-sided rule of pass 1, then the count).  Lists cost ~40 + 145 f ms per 2.6e10 pairs, the list-free form
// ~76 ms whatever f is: the launcher switches at f = 0.22.  (The sample: 256 objects x every fourth 64-model
// group; the best chi2 of the subsample is a little worse than the true one, which can only raise the estimate.)
template <class SRC>
__global__ __launch_bounds__(256) void k_nl_probe(SRC src_, int64_t N, int M, int S, double wt_thresh, const int* __restrict__ omap,
                                                  unsigned long long* __restrict__ count) {
    constexpr double K = (double)SRC::WPOW;
    SRC src = src_;
    src.tb = global_tabs();
    const int lane = threadIdx.x & 63;
    const int si = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (si >= S) return;
    const int64_t slot = (int64_t)si * N / S;
    typename SRC::OR ob;
    src.load_obj(omap ? (int64_t)omap[slot] : slot, ob);
    double lo = -INFINITY, hi = INFINITY;
    for (int j = lane; j < M; j += 256) {                          // every fourth 64-model group: an estimate is all that is needed
        typename SRC::MR m;
        src.load_model(j, m);
        const double c2 = src.chi2_of(ob, m);
        const bool below = c2 <= K;
        lo = fmax(lo, below ? c2 : -INFINITY);
        hi = fmin(hi, below ? INFINITY : c2);
    }
    lo = wave_max(lo); hi = -wave_max(-hi);
    const double ll = (lo >= 0.0) ? src.lnl_of_chi2(lo) : -INFINITY;
    const double lh = (hi < 1e299) ? src.lnl_of_chi2(hi) : -INFINITY;
    const double lmax = fmax(ll, lh), lt = log(wt_thresh);
    unsigned cnt = 0;
    for (int j = lane; j < M; j += 256) {
        typename SRC::MR m;
        src.load_model(j, m);
        cnt += (src.lnl_of_chi2(src.chi2_of(ob, m)) - lmax > lt) ? 1u : 0u;
    }
    const unsigned long long tot = (unsigned long long)wave_sum((double)cnt);
    if (lane == 0) atomicAdd(count, tot);
}

// ---- pass 1: the chi2 of each object's best model ---------------------------------------------------
template <class SRC, int NW>
__global__ __launch_bounds__(NW * 64) void k_nl_max(SRC src_, int64_t N, int M, const int* __restrict__ omap, double* __restrict__ cbest) {
    constexpr int TW = 2, TILE = (SRC::RW <= 6) ? 1024 : 512, TD = SRC::RW * TILE, NT = NW * 64, OD = SRC::OBJ_DOUBLES;
    constexpr double K = (double)SRC::WPOW;                       // the mode of chi2^(K/2) e^(-chi2/2)
    __shared__ __attribute__((aligned(16))) double tileA[TD];
    __shared__ __attribute__((aligned(16))) double tileB[TD];
    __shared__ __attribute__((aligned(16))) double s_objs[NW * TW * OD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * NW, gw = (int64_t)blockIdx.x * NW + wave;
    const int64_t ngroups = (N + TW - 1) / TW, nrounds = (ngroups + nwaves - 1) / nwaves;
    const int ntiles = (M + TILE - 1) / TILE;
    double* objs = s_objs + wave * (TW * OD);
    SRC src = src_;
    src.tb = global_tabs();
    for (int64_t rnd = 0; rnd < nrounds; ++rnd) {
        const int64_t g = gw + rnd * nwaves;
        const bool work = g < ngroups;
        const int64_t i0 = work ? g * TW : 0;
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            const int64_t os = i0 + o < N ? i0 + o : N - 1;
            src.park_obj(omap ? (int64_t)omap[os] : os, objs + o * OD, lane);
        }
        double lo[TW], hi[TW];
#pragma unroll
        for (int o = 0; o < TW; ++o) { lo[o] = -INFINITY; hi[o] = INFINITY; }
        nl_stage_tile<SRC, TILE, NT, false>(src, nullptr, 0, tileA, tid, wave);
        __syncthreads();
        auto run_tile = [&](const double* cur, double* nxt, int t) {
            if (t + 1 < ntiles) nl_stage_tile<SRC, TILE, NT, false>(src, nullptr, t + 1, nxt, tid, wave);
            if (work) {
#pragma unroll 1
                for (int s = 0; s < TILE / 64; ++s) {
                    const int j = t * TILE + s * 64 + lane;
                    typename SRC::MR m;
                    src.template load_model_lds<TILE>(cur, s * 64 + lane, m);
#pragma unroll
                    for (int o = 0; o < TW; ++o) {
                        typename SRC::OR ob;
                        src.load_obj_lds(objs + o * OD, ob);
                        double c2 = src.chi2_of(ob, m);
                        c2 = (j < M) ? c2 : 1e300;                // pad models: far above the mode, never the smallest
                        const bool below = c2 <= K;
                        lo[o] = fmax(lo[o], below ? c2 : -INFINITY);
                        hi[o] = fmin(hi[o], below ? INFINITY : c2);
                    }
                }
            }
            __syncthreads();
        };
        for (int t = 0; t < ntiles; t += 2) {
            run_tile(tileA, tileB, t);
            if (t + 1 < ntiles) run_tile(tileB, tileA, t + 1);
        }
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            const double l = wave_max(lo[o]), h = -wave_max(-hi[o]);
            // the larger of the two ln-likes (either side may be empty: -inf / +inf)
            const double ll = (l >= 0.0) ? src.lnl_of_chi2(l) : -INFINITY;
            const double lh = (h < 1e299) ? src.lnl_of_chi2(h) : -INFINITY;
            if (work && lane == 0 && i0 + o < N) cbest[i0 + o] = (ll >= lh) ? ((ll == -INFINITY && lh == -INFINITY) ? (double)NAN : l) : h;
        }
    }
}

// ---- pass 2: weights against the known maximum, straight into the LDS histogram ---------------------
template <class SRC, int NW>
__global__ __launch_bounds__(NW * 64) void k_nl_main(SRC src_, const KdeView* __restrict__ kvp, int acc_stride, int64_t N, int M,
                                                     double wt_thresh, int normalize, const int* __restrict__ omap,
                                                     const double* __restrict__ cbest, double* __restrict__ lmap,
                                                     double* __restrict__ levid, double* __restrict__ pdfs, int* __restrict__ redo) {
    constexpr int TILE = (SRC::RW <= 6) ? 256 : 128, TDR = SRC::RW * TILE, TD = TDR + TILE / 2, NT = NW * 64, OD = SRC::OBJ_DOUBLES;
    constexpr int WP = SRC::WPOW;
    __shared__ __attribute__((aligned(16))) double tileA[TD];
    __shared__ __attribute__((aligned(16))) double tileB[TD];
    __shared__ __attribute__((aligned(16))) double s_tabs[FZ_TABS_DOUBLES];
    __shared__ __attribute__((aligned(16))) double s_objs[NW * OD];
    extern __shared__ double s_rows[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * NW, gw = (int64_t)blockIdx.x * NW + wave;
    const int64_t nrounds = (N + nwaves - 1) / nwaves;
    const int ntiles = (M + TILE - 1) / TILE;
    double* objs = s_objs + wave * OD;
    double* row = s_rows + (size_t)wave * acc_stride;
    SRC src = src_;
    src.tb = stage_tabs(s_tabs, tid, NT);
    const FastTabs tb = src.tb;
    const KdeView kv = *kvp;
    const int32_t* posw = kv.pos;
    const int w0 = kv.w0;
    for (int64_t rnd = 0; rnd < nrounds; ++rnd) {
        const int64_t slot = gw + rnd * nwaves;
        const bool work = slot < N;
        const int64_t i = work ? (omap ? (int64_t)omap[slot] : slot) : 0;
        src.park_obj(i, objs, lane);
        for (int k = lane; k < acc_stride; k += 64) row[k] = 0.0;
        // the best model's chi2 (pass 1): reference of every weight of this object
        const double cref = uniform_d(cbest[work ? slot : 0]);
        const bool live = work && (cref == cref);
        const double lmax = live ? uniform_d(src.lnl_of_chi2(cref)) : (double)NAN;
        const double crs = live ? ((cref > 0.0) ? cref : 1.0) : 1.0;
        const double rcr = uniform_d(1.0 / crs);
        double S = 0.0;
        nl_stage_tile<SRC, TILE, NT, true>(src, posw, 0, tileA, tid, wave);
        __syncthreads();
        auto run_tile = [&](const double* cur, double* nxt, int t) {
            if (t + 1 < ntiles) nl_stage_tile<SRC, TILE, NT, true>(src, posw, t + 1, nxt, tid, wave);
            if (live) {
#pragma unroll 2
                for (int st = 0; st < TILE / 64; ++st) {
                    const int j = t * TILE + st * 64 + lane;
                    typename SRC::MR m;
                    src.template load_model_lds<TILE>(cur, st * 64 + lane, m);
                    const int tag = reinterpret_cast<const int32_t*>(cur + TDR)[st * 64 + lane];
                    typename SRC::OR ob;
                    src.load_obj_lds(objs, ob);
                    double c2 = src.chi2_of(ob, m);
                    c2 = (j < M) ? c2 : 1e30;                     // pad models: weight 0
                    // weight relative to the best, all fp64: (chi2 / cref)^(WP/2) exp(-(chi2 - cref)/2) -- integer powers by
                    // multiplication, the half power by a Newton-refined v_rsq_f64, one exp, no log.  (A fp32 screen in front of
                    // this, as in k_fused, does not pay here: the form only runs when most pairs pass it.)
                    const double dc = c2 - cref;
                    const double r = c2 * rcr;
                    double pw = 1.0;
                    if constexpr ((WP >> 1) >= 1) pw = r;
                    if constexpr ((WP >> 1) >= 2) pw = pw * r;
                    if constexpr ((WP >> 1) >= 3) pw = pw * r;
                    if constexpr (WP & 1) {                       // r^(1/2) = r * rsqrt(r)
                        double y = __builtin_amdgcn_rsq(r);
                        y = y * fma(-0.5 * r, y * y, 1.5);
                        y = y * fma(-0.5 * r, y * y, 1.5);
                        pw = pw * ((r > 0.0) ? r * y : 0.0);      // chi2 == 0 (self match): weight 0, as chi2^(k/2) says
                    }
                    const double w = pw * exp_clamped(-0.5 * dc, tb);      // the argument is > 0 for pairs below the mode; bounded because cref is the arg-max
                    if (w > wt_thresh) unsafeAtomicAdd(&row[tag + w0], w);         // strict; max(w) == 1 (pdf.py:591)
                    S += w;
                }
            }
            __syncthreads();
        };
        for (int t = 0; t < ntiles; t += 2) {
            run_tile(tileA, tileB, t);
            if (t + 1 < ntiles) run_tile(tileB, tileA, t + 1);
        }
        const double stot = wave_sum(S);
        const double le = lmax + log_pos(stot, tb);
        const bool ok = live && (le - le == 0.0) && stot > 0.0;
        if (work && lane == 0) {
            if (lmap) lmap[i] = lmax;
            if (levid) levid[i] = le;
            if (!ok) redo[1 + atomicAdd(redo, 1)] = (int)i;      // nan / empty rows: the exact ln-space sweep decides
        }
        if (work) kde_finalize<true>(kv, row, ok, normalize, pdfs + i * kv.G, lane, ok ? exp_neg(lmax - le, tb) : 1.0, true);
        __syncthreads();
    }
}

}  // namespace fz
