// Shared by the fused kernels: the LDS-DMA copy of one model tile, and the per-launch sample of how broad the likelihoods are.
// (Round 2's list-free two-pass form for broad likelihoods -- k_nl_max / k_nl_main -- lived here; since round 3 no default dispatch
// reached it, k_hist's direct form does its job in one pass, and round 4 removed it.)
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
// two-sided rule -- the likelihood chi2^(k/2) e^(-chi2/2) is unimodal with its mode at chi2 = k, so the best model is the largest
// chi2 at or below the mode or the smallest above it -- then the count).  The launcher picks k_hist's form by it (fz_launch.h).  (The sample: 256 objects x every fourth 64-model
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

}  // namespace fz
