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
    FastTabs tb;
    static constexpr int WPOW = 0;
    struct MR {};
    struct OR { const double* row; };
    __device__ __forceinline__ void load_model(int64_t, MR&) const {}
    __device__ __forceinline__ void load_obj(int64_t i, OR& o) const { o.row = p + i * ld; }
    __device__ __forceinline__ void load_obj_fresh(int64_t i, OR& o) const { o.row = p + i * ld; }
    __device__ __forceinline__ double lnl(const OR& o, const MR&, int64_t j, bool valid) const {
        return valid ? o.row[j] : -INFINITY;
    }
};
template <int BT, int MODE, int VAR, bool PRI = false>
struct PhotSrc : Phot<BT, MODE, VAR> {
    using P = Phot<BT, MODE, VAR>;
    // With the dimensionality prior and no masks the likelihood is chi2^(WPOW/2) e^(-chi2/2)/C
    // with a compile-time half-integer power (B = 5: 3/2 for modes A/Ai, 1 for mode B), so
    // the weight can be formed with a square root instead of a logarithm (k_fused, WM).
    // An additive ln-prior (PRI) keeps to the ln-space body, which tracks nan / +-inf rows.
    static constexpr int WPOW = (VAR == VAR_FAST && !PRI) ? (MODE == 2 ? BT - 3 : BT - 2) : 0;
    static constexpr bool HAS_PRIOR = PRI;
    static constexpr int NB = BT;
    static constexpr int LMODE = MODE;
    // mask-free tame data cannot produce a nan ln-like (chi2 is finite and >= 0, the exponent of the
    // dimensionality prior is positive): the nan bookkeeping of the ln-space body is compiled out
    static constexpr bool NO_NAN = (VAR == VAR_FAST) && !PRI;
    // launch geometry preference of the ln-space body (measured, profiles/README.md)
    static constexpr bool PREF_2x16 = (BT <= 6) && ((MODE == 1) || (MODE == 2) || (MODE == 0 && VAR != VAR_FAST));   // r1_v7 sweep: masked mode B 150 vs 161 ms at (4,8)
    static constexpr bool PREF_2x8 = false;   // r1_v7 sweep: 7 / 8-band unmasked mode A now runs best at (4,8): 97 vs 105 ms, 103 vs 109 ms (was (2,8) before the lean instantiation)
    PriorView pv;                                 // read only when PRI
    struct OR : P::OR { const double* prow; };    // + the object's ln-prior row
    __device__ __forceinline__ void load_obj(int64_t i, OR& o) const {
        P::load_obj(i, o);
        if (PRI) o.prow = pv.tab + pv.row(i) * pv.ld;
    }
    // the row index (< 2^31, checked on the host) rides in the spare high word of the
    // parked object row's mask slot
    __device__ __forceinline__ void park_obj(int64_t i, double* dst, int lane) const {
        P::park_obj(i, dst, lane);
        if (PRI && lane == BT) dst[2 * BT + 1] = __hiloint2double((int)pv.row(i), P::MASKED ? (int)P::ov.bits[i] : -1);
    }
    __device__ __forceinline__ void load_obj_lds(const double* p, OR& o) const {
        P::load_obj_lds(p, o);
        if (PRI) o.prow = pv.tab + (int64_t)__double2hiint(p[2 * BT + 1]) * pv.ld;
    }
    __device__ __forceinline__ double chi2_of(const typename P::OR& o, const typename P::MR& m) const {
        return P::template eval<1>(o, m).chi2;          // the unused ln-like tail is dead code
    }
    __device__ __forceinline__ double lnl_of_chi2(double chi2) const {
        return chi2_logpdf<true>(0.5 * WPOW, chi2, P::lp.lg_full, P::tb);
    }
    // the same with the half power given at run time (band counts padded up to BT; lg_full is that of the real count)
    __device__ __forceinline__ double lnl_of_chi2_k(double chi2, double halfk) const {
        return chi2_logpdf<true>(halfk, chi2, P::lp.lg_full, P::tb);
    }
    __device__ __forceinline__ double lnl(const OR& o, const typename P::MR& m, int64_t j, bool valid) const {
        double l = P::eval(o, m).lnl;             // pad lanes hold benign data; no divergent branch
        if (PRI) l += valid ? o.prow[j] : 0.0;
        return valid ? l : -INFINITY;
    }
    // j / inb: the lane's model and whether it is a real one (only the prior read needs them)
    template <int DPT>
    __device__ __forceinline__ double lnl_t(const OR& o, const typename P::MR& m, int j, bool inb) const {
        double l = P::template eval<DPT>(o, m).lnl;
        if (PRI) l += inb ? o.prow[j] : 0.0;
        return l;
    }

    // ---- LDS-staged model tiles (k_fused) ----
    // A tile is TILE consecutive model RECORDS (y[BT], then ye2[BT] in mode A, padded to
    // RW doubles) followed by the TILE mask words: a contiguous slice of the record
    // array, staged with plain 16-B-per-lane copies.  RW*8 is 16 mod 32 bytes, which
    // makes a wave's ds_read_b128 of 64 consecutive records bank-conflict free.
    static constexpr int NVAL = BT + (MODE == 0 ? BT : 0);
    // models per LDS tile: 256, fewer for wide records so that two tiles stay well inside the LDS
    // (512 measured slower at 5 bands: staging registers)
    static constexpr int TILE = (NVAL > 32) ? 64 : (NVAL > 16 ? 128 : 256);
    static constexpr int RW = NVAL + ((6 - NVAL % 4) % 4);          // smallest width >= NVAL that is 2 mod 4
    static_assert(RW >= NVAL && RW % 4 == 2, "record width must be 2 mod 4 doubles (16 mod 32 bytes)");
    // k_fused may run longer tiles (fewer block barriers) where its LDS allows: FZ_MAX_TILE models, to
    // which the model arrays are padded (Mp).  TL = tile length actually used.
    template <int NWAVES>
    static constexpr int tile_len() { return (NWAVES >= 12 && RW <= 6) ? 1024 : ((NWAVES >= 8 && RW <= 10) ? 512 : TILE); }
    template <int TL> static constexpr int tile_doubles() { return RW * TL + (P::MASKED ? TL / 2 : 0); }
    static constexpr int TILE_DOUBLES = RW * TILE + (P::MASKED ? TILE / 2 : 0);
    static constexpr int NCHUNK = TILE_DOUBLES / 2;                 // 16-byte chunks
    template <int TL = TILE>
    __device__ __forceinline__ const double* tile_chunk_ptr(int64_t tile, int ch) const {
        const double* rec = (MODE == 0) ? P::mv.rec0 : P::mv.rec1;
        return (ch < RW * TL / 2) ? rec + tile * (TL * RW) + 2 * ch
                                  : reinterpret_cast<const double*>(P::mv.bits + tile * TL) + 2 * (ch - RW * TL / 2);
    }
    template <int TL = TILE>
    __device__ __forceinline__ double2 tile_chunk(int64_t tile, int ch) const {
        return *reinterpret_cast<const double2*>(tile_chunk_ptr<TL>(tile, ch));
    }
    // model j from the array-of-records copy, per-lane j, as 16-byte loads (records are 16-B aligned: RW is even): one or two
    // cache lines per model instead of BT (2 BT in mode A) -- for GATHERS (k_knn_subset: 500 scattered models per object; the
    // [band][model] arrays moved 320 KB of cache lines per object, the records move 64 KB)
    __device__ __forceinline__ void load_model_rec16(int64_t j, typename P::MR& m) const {
        const double2* r = reinterpret_cast<const double2*>(((MODE == 0) ? P::mv.rec0 : P::mv.rec1) + j * RW);
        double v[RW];
#pragma unroll
        for (int q = 0; q < RW / 2; ++q) { const double2 w = r[q]; v[2 * q] = w.x; v[2 * q + 1] = w.y; }
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            m.y[b] = v[b];
            if (MODE == 0) m.ye2[b] = v[BT + b];
        }
        m.bits = P::MASKED ? P::mv.bits[j] : 0xffffffffu;
    }
    template <int TL = TILE>
    __device__ __forceinline__ void load_model_lds(const double* t, int k, typename P::MR& m) const {
        const double2* r = reinterpret_cast<const double2*>(t + k * RW);
        double v[RW];
#pragma unroll
        for (int q = 0; q < RW / 2; ++q) { const double2 w = r[q]; v[2 * q] = w.x; v[2 * q + 1] = w.y; }
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            m.y[b] = v[b];
            if (MODE == 0) m.ye2[b] = v[BT + b];
        }
        m.bits = P::MASKED ? reinterpret_cast<const uint32_t*>(t + RW * TL)[k] : 0xffffffffu;
    }
};

// ---- materialising fit ------------------------------------------------------
// A thread owns MPT adjacent models (registers); the block walks TO objects (16 for small
// problems, 256 when that still fills the chip: fewer model loads per plane element), UNR at a
// time so that the independent evaluations interleave; rows of the (N,M) planes are written as
// coalesced streaming stores: 8 B per lane (MPT = 1: 512 B per wave-instruction) or 16 B per lane
// (MPT = 2: 1 KB per wave-instruction; needs M even and 16-B aligned planes).  The planes are
// never read back by this library, so the stores are non-temporal.
typedef double fz_d2 __attribute__((ext_vector_type(2)));
typedef long long fz_l2 __attribute__((ext_vector_type(2)));
template <class PH, int TO, int DPT, int MPT>
__global__ __launch_bounds__(256) void k_planes(PH ph_, int64_t N, int64_t M, double* __restrict__ lnl,
                                                double* __restrict__ chi2, int64_t* __restrict__ ndim,
                                                double* __restrict__ scale, double* __restrict__ serr) {
    constexpr int UNR = 4 / MPT;
    static_assert(TO % UNR == 0, "TO must be a multiple of the unroll");
    PH ph = ph_;
    ph.tb = global_tabs();
    const int64_t j = ((int64_t)blockIdx.y * 256 + threadIdx.x) * MPT;
    const bool valid = j < M;                  // MPT = 2: M is even, so j + 1 < M as well
    typename PH::MR m[MPT];
#pragma unroll
    for (int q = 0; q < MPT; ++q) {
        const int64_t jq = j + q < ph.mv.Mp ? j + q : ph.mv.Mp - 1;     // Mp is a multiple of 256
        ph.load_model(jq, m[q]);
    }
    const int64_t i0 = (int64_t)blockIdx.x * TO;
    for (int o0 = 0; o0 < TO; o0 += UNR) {
        if (i0 + o0 >= N) break;
        PairOut r[UNR][MPT];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + o0 + u < N ? i0 + o0 + u : N - 1;
            typename PH::OR ob;
            ph.load_obj(i, ob);
#pragma unroll
            for (int q = 0; q < MPT; ++q) r[u][q] = ph.template eval<DPT>(ob, m[q]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + o0 + u;
            if (valid && i < N) {
                const int64_t k = i * M + j;
                if (MPT == 1) {
                    if (lnl) __builtin_nontemporal_store(r[u][0].lnl, &lnl[k]);
                    if (chi2) __builtin_nontemporal_store(r[u][0].chi2, &chi2[k]);
                    if (ndim) __builtin_nontemporal_store((int64_t)r[u][0].ndim, &ndim[k]);
                    if (scale) __builtin_nontemporal_store(r[u][0].scale, &scale[k]);
                    if (serr) __builtin_nontemporal_store(sqrt(1.0 / r[u][0].shape), &serr[k]);      // pdf.py:232
                } else {
                    const PairOut &a = r[u][0], &b = r[u][MPT - 1];
                    if (lnl) __builtin_nontemporal_store(fz_d2{a.lnl, b.lnl}, reinterpret_cast<fz_d2*>(&lnl[k]));
                    if (chi2) __builtin_nontemporal_store(fz_d2{a.chi2, b.chi2}, reinterpret_cast<fz_d2*>(&chi2[k]));
                    if (ndim) __builtin_nontemporal_store(fz_l2{a.ndim, b.ndim}, reinterpret_cast<fz_l2*>(&ndim[k]));
                    if (scale) __builtin_nontemporal_store(fz_d2{a.scale, b.scale}, reinterpret_cast<fz_d2*>(&scale[k]));
                    if (serr) __builtin_nontemporal_store(fz_d2{sqrt(1.0 / a.shape), sqrt(1.0 / b.shape)},
                                                          reinterpret_cast<fz_d2*>(&serr[k]));
                }
            }
        }
    }
}

// ---- KDE tables ------------------------------------------------------------------
typedef double fz_d2l __attribute__((ext_vector_type(2)));
struct KdeView {
    int64_t G;
    int kmode;             // KDE_HIST / KDE_DICT / KDE_GRID
    // dictionary path (pdf.py:599-620), per model (padded to Mp)
    const int32_t* pos;    // y_idx
    const int32_t* cls;    // y_std_idx
    const double* norm;    // edge-truncated kernel mass (pdf.py:613-617) / in-window sum (pdf.py:521)
    const double* normtab; // KDE_HIST: the same mass as a function of the (padded) histogram index, [G + 2 w0]
    const int64_t* widths; const int64_t* offsets; const double* kern;
    int32_t w0; int64_t koff0;            // single sigma class: width and table offset
    // direct path (pdf.py:499-502, 519-524), per model
    const double* ly; const double* lstd; const int32_t* lo; const int32_t* hi;
    const double* grid;
    double gstep;                         // direct path: the grid's spacing when it is evenly spaced (to 1e-9 of a step), else 0
    const double* lrec;                   // direct path, even grid: 48-byte record per model (k_prep_grid_labels)
    int acc_stride;                       // doubles of LDS per object
    int lane_window;                      // DICT / GRID: windows up to this many grid points are added by their own lane
    // class-sorted dictionary stack (k_fused<..., MC>): the kernel's model records are ordered by
    // dictionary class; mc_tag[j'] = rank << 10 | (y_idx + mc_w0) of the model in place j'; per rank:
    // half-width, kernel-table offset, and the edge-truncated mass per padded index [rank][mc_gp]
    const int32_t* mc_tag; const int32_t* mc_width; const int64_t* mc_off; const double* mc_norm;
    int32_t mc_gp, mc_w0;
    // segmented model layout (k_hist<..., SEG>, fz_hist.h): the kernel's copy of the model records is sorted by (dictionary class,
    // mask pattern) and every segment is padded to whole 64-model groups; mc_tag[j'] = (y_idx + mc_w0) | segment << 16 |
    // (group holds pad slots) << 30 | (pad slot) << 31; per segment: the models' mask word and the rank of their dictionary class
    const uint32_t* seg_mask; const int32_t* seg_rank; const int32_t* seg_start; int32_t seg_n, seg_nrank;      // seg_start[s .. s + 1]: the segment's slots
};
// HIST: every label shares one dictionary kernel -> accumulate w/norm at the label's
// grid index (one LDS atomic per selected model), convolve once at the end.
// DICT / GRID: add each selected model's window, a wave per model.
enum { KDE_HIST = 0, KDE_DICT = 1, KDE_GRID = 2 };

// add the selected lanes' kernels into `row`.  w: the lane's weight; jm: its model.
// DICT / GRID: a lane adds its own model's window (LDS float atomics; up to FZ_LANE_WINDOW grid
// points, so that 64 windows go on in parallel instead of one after the other, each behind its own
// chain of label gathers); wider windows are added one model at a time by the whole wave.
#define FZ_LANE_WINDOW 160               // default of KdeView::lane_window (FZ_LANE_WINDOW=n overrides, 0: always the whole wave)
__device__ __forceinline__ void kde_scatter(const KdeView& kv, double* row, bool sel, double w, int64_t jm,
                                            int lane) {
    if (kv.kmode == KDE_HIST) {
        if (sel) unsafeAtomicAdd(&row[kv.pos[jm] + kv.w0], w / kv.norm[jm]);
        return;
    }
    bool wide = false;
    if (kv.kmode == KDE_DICT) {
        if (sel) {
            const int p = kv.pos[jm], c = kv.cls[jm];
            const int wd = (int)kv.widths[c];
            const int lo = max(p - wd, 0), hi = min(p + wd + 1, (int)kv.G);
            wide = hi - lo > kv.lane_window;
            if (!wide) {
                const double wn = w / kv.norm[jm];
                const double* kr = kv.kern + kv.offsets[c] + (lo - (p - wd)) - lo;
                for (int t = lo; t < hi; ++t) unsafeAtomicAdd(&row[t], wn * kr[t]);
            }
        }
    } else if (kv.gstep > 0.0) {
        // evenly spaced grid: with z_t = (x_t - mu) / sd and h = step / sd the Gaussian obeys G_{t+1} = G_t R_t, R_{t+1} = R_t e^{-h^2},
        // R_t = exp(-z_t h - h^2 / 2): two exponentials per window and two multiplications per point instead of a table exponential
        // (and a grid load) per point; the products are re-seeded every 64 points, so the accumulated rounding stays below ~3e-13
        // relative.  What the window needs comes from the model's 48-byte record.
        if (sel) {
            const fz_d2l* rec = reinterpret_cast<const fz_d2l*>(kv.lrec + 6 * jm);
            const fz_d2l ra = rec[0], rb = rec[1], rc = rec[2];
            const int lo = __double2loint(rc.x), hi = __double2hiint(rc.x);
            wide = hi - lo > kv.lane_window;
            if (!wide && rb.x != 0.0) {                          // pdf.py:523: kernels with zero sum are skipped
                const double mu = ra.x, isd = ra.y, q = rb.y;
                const double wg = w * rb.x, h = kv.gstep * isd;
                const FastTabs tb = global_tabs();
                for (int t0 = lo; t0 < hi; t0 += 64) {
                    const double z = (kv.grid[t0] - mu) * isd;
                    double g = wg * exp_neg(-0.5 * (z * z), tb);
                    double r = exp_clamped(-fma(z, h, 0.5 * (h * h)), tb);
                    const int t1 = min(t0 + 64, hi);
                    for (int t = t0; t < t1; ++t) { unsafeAtomicAdd(&row[t], g); g *= r; r *= q; }
                }
            }
        }
    } else {
        if (sel) {
            const int lo = kv.lo[jm], hi = kv.hi[jm];
            const double nrm = kv.norm[jm];
            wide = hi - lo > kv.lane_window;
            if (!wide && nrm != 0.0) {                          // pdf.py:523
                const double mu = kv.ly[jm], sd = kv.lstd[jm];
                const double isd = 1.0 / sd;
                const double wg = (w / nrm) / (2.5066282746310002 * sd);      // weight / (sqrt(2 pi) * std)
                for (int t = lo; t < hi; ++t) {
                    const double z = (kv.grid[t] - mu) * isd;
                    unsafeAtomicAdd(&row[t], wg * exp_neg(-0.5 * (z * z)));
                }
            }
        }
    }
    // Windows left to the whole wave, lanes along the grid (no atomics).  The labels of ALL such
    // lanes are gathered first, lane-parallel (one round of dependent loads for the step instead of
    // one per model); the serial loop then only moves them with v_readlane, four models per trip so
    // that their kernel / grid loads are in flight together.  Adds of one trip go in model order:
    // LDS operations of a wave are executed in order, so overlapping windows are safe.
    int lo = 0, hi = 0, kidx = 0;
    double pa = 0.0, pb = 0.0;           // DICT: weight / norm, -     GRID: weight / (norm sqrt(2 pi) std), 1 / std
    double pc = 0.0;                     // GRID: label
    if (wide) {
        if (kv.kmode == KDE_DICT) {
            const int p = kv.pos[jm], c = kv.cls[jm];
            const int wd = (int)kv.widths[c];
            lo = max(p - wd, 0); hi = min(p + wd + 1, (int)kv.G);
            kidx = (int)kv.offsets[c] + (lo - (p - wd)) - lo;          // kernel entry of grid point t: kern[kidx + t]
            pa = w / kv.norm[jm];
        } else {
            const double nrm = kv.norm[jm], sd = kv.lstd[jm];
            lo = kv.lo[jm]; hi = (nrm != 0.0) ? kv.hi[jm] : lo;       // pdf.py:523: kernels with zero sum are skipped
            pa = (w / nrm) / (2.5066282746310002 * sd);
            pb = 1.0 / sd;
            pc = kv.ly[jm];
        }
    }
    auto rl = [](double v, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };
    unsigned long long mask = __ballot(wide);
    constexpr int UM = 4;
    while (mask) {
        int lo_u[UM], hi_u[UM], k_u[UM]; double a_u[UM], b_u[UM], c_u[UM], v_u[UM];
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            if (mask) {
                const int sl = __builtin_ctzll(mask);
                mask &= mask - 1;
                lo_u[u] = __builtin_amdgcn_readlane(lo, sl); hi_u[u] = __builtin_amdgcn_readlane(hi, sl);
                k_u[u] = __builtin_amdgcn_readlane(kidx, sl);
                a_u[u] = rl(pa, sl); b_u[u] = rl(pb, sl); c_u[u] = rl(pc, sl);
            } else { lo_u[u] = 0; hi_u[u] = 0; k_u[u] = 0; a_u[u] = b_u[u] = c_u[u] = 0.0; }
        }
#pragma unroll
        for (int u = 0; u < UM; ++u) {                                 // first 64 points of each window: loads together
            const int t = lo_u[u] + lane;
            v_u[u] = 0.0;
            if (t < hi_u[u]) v_u[u] = (kv.kmode == KDE_DICT) ? kv.kern[k_u[u] + t] : kv.grid[t];
        }
#pragma unroll
        for (int u = 0; u < UM; ++u) {
            const int t = lo_u[u] + lane;
            if (t < hi_u[u]) {
                if (kv.kmode == KDE_DICT) row[t] += a_u[u] * v_u[u];
                else { const double z = (v_u[u] - c_u[u]) * b_u[u]; row[t] += a_u[u] * exp_neg(-0.5 * (z * z)); }
            }
            for (int t2 = t + 64; t2 < hi_u[u]; t2 += 64) {            // the rest of a window wider than the wave
                if (kv.kmode == KDE_DICT) row[t2] += a_u[u] * kv.kern[k_u[u] + t2];
                else { const double z = (kv.grid[t2] - c_u[u]) * b_u[u]; row[t2] += a_u[u] * exp_neg(-0.5 * (z * z)); }
            }
        }
    }
}

// (convolve,) normalise, write one PDF row.  With `normalize` the un-normalised values are parked
// back in the wave's LDS row (each 128-point pass only overwrites inputs that no later pass reads)
// and divided on their way out: the PDF goes to HBM once and is never read back.
// HO: the caller's instantiation only ever sees the single-kernel (histogram) form.
// scale: factor on the un-normalised output (the weight-space body stacks weights relative to its
// own reference instead of the evidence; a normalised PDF does not see the difference)
// bymass: the histogram holds plain weight sums per index; the division by the kernel mass of the
// index (kv.normtab) happens here, once per index instead of once per stacked model.
template <bool HO = false>
__device__ __forceinline__ void kde_finalize(const KdeView& kv, double* row, bool ok, int normalize,
                                             double* out, int lane, double scale = 1.0, bool bymass = false) {
    const int G = (int)kv.G;
    if (!ok) { for (int t = lane; t < G; t += 64) out[t] = NAN; return; }
    double tot = 0.0;
    if (HO || kv.kmode == KDE_HIST) {
        const int w2 = 2 * kv.w0;
        if (bymass) { for (int k = lane; k < G + w2; k += 64) row[k] = row[k] / kv.normtab[k]; }
        const double* kr = kv.kern + kv.koff0;
        if (w2 < 128) {
            // the dictionary kernel (w2 + 1 taps) sits in two registers across the wave -- one coalesced
            // load per object instead of one load per tap and output -- and tap h reaches the FMA as a
            // scalar operand (v_readlane); the same sums in the same order as the plain loop below
            const double ka = (lane <= w2) ? kr[lane] : 0.0;
            const double kb = (lane + 64 <= w2) ? kr[lane + 64] : 0.0;
            const int kal = __double2loint(ka), kah = __double2hiint(ka), kbl = __double2loint(kb), kbh = __double2hiint(kb);
            for (int t = lane; t < G; t += 128) {
                const bool two = t + 64 < G;
                const double* r0 = row + t;
                const double* r1 = row + (two ? t + 64 : t);
                double v0 = 0.0, v1 = 0.0;
                const int hs = w2 < 64 ? 0 : w2 - 63;              // taps w2-h >= 64 (register kb): h < hs
                for (int h = 0; h < hs; ++h) {
                    const int q = w2 - h - 64;
                    const double tap = __hiloint2double(__builtin_amdgcn_readlane(kbh, q), __builtin_amdgcn_readlane(kbl, q));
                    v0 = fma(r0[h], tap, v0); v1 = fma(r1[h], tap, v1);
                }
                for (int h = hs; h <= w2; ++h) {
                    const int q = w2 - h;
                    const double tap = __hiloint2double(__builtin_amdgcn_readlane(kah, q), __builtin_amdgcn_readlane(kal, q));
                    v0 = fma(r0[h], tap, v0); v1 = fma(r1[h], tap, v1);
                }
                tot += v0;
                if (two) tot += v1;
                if (normalize) { row[t] = v0; if (two) row[t + 64] = v1; }
                else { out[t] = v0 * scale; if (two) out[t + 64] = v1 * scale; }
            }
        } else {
            // wide kernels: outputs go straight out (a row entry is still an input of later outputs)
            for (int t = lane; t < G; t += 64) {
                double v = 0.0;
                for (int h = 0; h <= w2; ++h) v = fma(row[t + h], kr[w2 - h], v);
                out[t] = normalize ? v : v * scale;
                tot += v;
            }
            if (normalize) {
                tot = wave_sum(tot);
                for (int t = lane; t < G; t += 64) out[t] = out[t] / tot;
            }
            return;
        }
    } else {
        for (int t = lane; t < G; t += 64) { const double v = row[t]; tot += v; if (!normalize) out[t] = v * scale; }
    }
    if (normalize) {
        tot = wave_sum(tot);
        for (int t = lane; t < G; t += 64) out[t] = row[t] / tot;       // pdf /= pdf.sum()
    }
}

// ---- pass 1: per-object max and logsumexp ------------------------------------
// linear=1: rows are linear weights; only the max is produced (np.max: NaN wins).
template <class SRC, int TW>
__global__ __launch_bounds__(256) void k_stats(SRC src_, int64_t N, int64_t M, int linear,
                                               double* __restrict__ lmap, double* __restrict__ levid) {
    SRC src = src_;
    src.tb = global_tabs();
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
            else ms_push(st[o], l, src.tb);
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
__global__ __launch_bounds__(256) void k_kde(SRC src_, KdeView kv, int64_t N, int64_t M, int linear,
                                             const double* __restrict__ lmap,
                                             const double* __restrict__ levid, double wt_thresh,
                                             int normalize, double* __restrict__ pdfs) {
    extern __shared__ double smem[];
    SRC src = src_;
    src.tb = global_tabs();
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
            thr[o] = wt_thresh * exp_neg(lm - le[o], src.tb); // wt_thresh * max(wt)
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
            const bool cand = valid && (linear ? true : (l >= lthr[o]));   // >=: survives absorption of the offset at huge |lnl|
            if (!__any(cand)) continue;                             // wave-uniform
            const double w = linear ? l : exp_neg(l - le[o], src.tb);
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

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// ---- single pass: likelihood + softmax statistics + candidates -> PDF -----------
struct Cand { double lnl; int32_t j; int32_t pad; };      // 16 B, one dwordx4 store (lnl: the ln-like, or chi2 in the weight-space body)
#define FZ_RES 4                                            // doubles of per-object results a wave parks in LDS

// Block = NW waves that stream the model set TOGETHER through double-buffered LDS
// tiles (each tile is fetched from L2/HBM once per block, not once per wave), while
// every wave keeps its own TW objects: per-object reductions stay private to a wave.
// Waves walk their object groups with a grid stride; each owns a candidate buffer of
// TW x cap entries (cap = M: it can never overflow).
//
// A pair is recorded when its lnl is within the weight threshold of the best lnl
// seen SO FAR (per lane, tightened every 64 steps with the wave-wide best): a
// superset of the pairs with wt > wt_thresh * max(wt), because the running best only
// grows.  The exact test (pdf.py:510 / 591, strict >) is applied afterwards with the
// final max and evidence, then the kernels are stacked and the PDF normalised.

// per-wave running state of the single-pass kernel
template <int TW>
struct FusedState {
    MS st[TW];
    int cnt[TW];
    unsigned firstnan, anynan;
    int tick;
};

// One LDS tile (TILE models, 64 per step) against the wave's TW objects.  The work of
// a step is laid out stage by stage over the objects -- likelihoods, then softmax
// updates, then candidate appends -- so that the independent per-object chains sit in
// one basic block and their LDS / transcendental latencies overlap.  DPT pins
// dim_prior; TAIL = the last, possibly partial tile (only there are lanes masked).
template <class SRC, int TW, int DPT, bool TAIL, int TL>
__device__ __forceinline__ void fused_tile(const SRC& src, const FastTabs& tb, const double* cur, const double* objs,
                                           int jt0, int M, int lane, double lt, Cand* buf, int64_t cap,
                                           FusedState<TW>& fs) {
    constexpr int OD = SRC::OBJ_DOUBLES;
#pragma unroll 1
    for (int s = 0; s < TL / 64; ++s) {
        const int j = jt0 + s * 64 + lane;
        typename SRC::MR m;
        src.template load_model_lds<TL>(cur, s * 64 + lane, m);
        double l[TW];
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            typename SRC::OR ob;
            src.load_obj_lds(objs + o * OD, ob);
            l[o] = src.template lnl_t<DPT>(ob, m, j, TAIL ? (j < M) : true);
            if (TAIL) l[o] = (j < M) ? l[o] : -INFINITY;
        }
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            if constexpr (!SRC::NO_NAN) {
                if (l[o] != l[o]) { fs.anynan |= 1u << o; if (j == 0) fs.firstnan |= 1u << o; }
            }
            ms_push(fs.st[o], l[o], tb);
        }
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            bool c = l[o] >= fs.st[o].m + lt;                             // >=: lt may be absorbed when |lnl| is huge; false for nan
            if (TAIL) c = c && (j < M);      // a pad lane that never saw a real model has m = -inf and would pass (-inf >= -inf):
                                             // only real models may be recorded, the list holds exactly M entries
            const unsigned long long mask = __ballot(c);
            if (mask) {                                                   // wave-uniform
                const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                    __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (c) { Cand e; e.lnl = l[o]; e.j = j; e.pad = 0; buf[(size_t)o * cap + fs.cnt[o] + pre] = e; }
                fs.cnt[o] += __builtin_popcountll(mask);
            }
        }
        // every 64 steps re-reference each lane's (max, sum) to the wave-wide best, so
        // that the candidate filter works against the best lnl any lane has seen (the
        // sum is rescaled accordingly: exact)
        if ((++fs.tick & 63) == 0) {
#pragma unroll
            for (int o = 0; o < TW; ++o) {
                const double mx = wave_max(fs.st[o].m);
                fs.st[o].s *= exp_neg(fs.st[o].m - mx, tb);
                fs.st[o].m = mx;
            }
        }
    }
}

// ---- weight-space variant of the tile step (SRC::WPOW != 0, dim_prior on) ----------
// The likelihood is chi2^(WPOW/2) e^(-chi2/2) / C.  chi2 is formed in fp64 exactly as the other
// bodies form it; what follows it -- the weight of the pair relative to a per-object reference, the
// running sum and the candidate test -- runs in fp32 (v_exp_f32 / v_sqrt_f32: ~60 cycles per 64 pairs
// instead of ~170 for the fp64 log-free form it replaces; tools/ubench_mix.hip has the issue rates).
// That is sound because the hot loop only has to (a) find a SUPERSET of the models that pass the
// weight threshold and (b) sum the weights of the rest: every candidate carries its fp64 chi2 and
// is re-evaluated in fp64 by the PDF stage (exact ln-like, exact maximum, exact threshold, exact
// weight in the PDF and in the evidence); only the sum over the NON-candidates (each below
// wt_thresh of the best) enters the evidence in fp32, ~1e-7 relative on that part.
//
// Reference.  Each object carries (ref, cref): a reference ln-like and a reference chi2, and
//     w = exp(lnl - ref) = 2^( (k/2) log2(chi2 / cref) - (chi2 - cref) log2(e) / 2 + kp ),   kp = (lnl(cref) - ref) log2(e)
// holds for ANY such pair; a good pair has ref near the best ln-like (weights stay inside the fp32
// range) and cref near the chi2 of the models that matter, because the difference chi2 - cref is
// taken in fp64 BEFORE the conversion and the logarithm is that of a ratio near 1: the fp32
// exponent is then small, and accurate, wherever the weight matters (its error is relative to the
// distance from the best fit, not to chi2 itself).  ref follows the largest weight seen; cref is
// the chi2 >= k at which the likelihood chi2^(k/2) e^(-chi2/2) takes the value e^ref (k is its
// mode: above it every ln-like is taken once).  Re-bases: on an object's first step (from the exact
// ln-likes of its first 64 models), every 64 steps, and at once when a pair's exponent passes 60 or
// is not a number (a new best by > e^40; chi2 beyond the fp32 range) -- that re-base takes the exact
// ln-likes of the current step into account, so afterwards every weight of the step is <= ~1; lanes
// whose exponent still is out of range then are recorded as candidates on the spot (exact
// treatment) and carry no weight in the loop.
template <int TW>
struct WState {
    double ref[TW], cref[TW];      // reference ln-like and chi2 (wave-uniform)
    float rcr[TW], kp[TW];         // 1 / cref, (lnl(cref) - ref) log2(e)  (wave-uniform)
    double S[TW];                  // per-lane: non-candidate weight sum (flushed from s)
    float s[TW], wmax[TW];         // per-lane: sum since the last flush, largest weight
    int cnt[TW];
    int tick;
};
template <int TW>
__device__ __forceinline__ void w_init(WState<TW>& ws) {
#pragma unroll
    for (int o = 0; o < TW; ++o) {
        ws.ref[o] = -INFINITY; ws.cref[o] = 1e300; ws.rcr[o] = 0.f; ws.kp[o] = 0.f;
        ws.S[o] = 0.0; ws.s[o] = 0.f; ws.wmax[o] = 0.f; ws.cnt[o] = 0;
    }
    ws.tick = 0;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Move object o's reference to the best ln-like its wave has seen: the largest recorded weight,
// or -- c2 != nullptr: the re-bases inside a step -- a model of the current step if its exact ln-like
// beats it.
template <class SRC, int TW>
__device__ __forceinline__ void w_rebase(const SRC& src, WState<TW>& ws, int o, const double* c2) {
    constexpr double K = (double)SRC::WPOW;
    ws.S[o] += (double)ws.s[o]; ws.s[o] = 0.f;
    const float wm = wave_maxf(ws.wmax[o]);
    double nr = (wm > 0.f) ? ws.ref[o] + log_pos((double)wm, src.tb) : -INFINITY;       // ~1e-7: any reference near the best will do
    if (c2) nr = fmax(nr, wave_max(src.lnl_of_chi2(*c2)));       // pad lanes: chi2 = 1e30 -> ~ -5e29; chi2 == 0 -> -inf
    nr = uniform_d(nr);
    if (nr > ws.ref[o]) {                                        // wave-uniform; false while every ln-like so far is -inf
        const double f = uniform_d(exp_neg(ws.ref[o] - nr, src.tb));       // ref = -inf: S and wmax are still 0
        ws.S[o] *= f; ws.wmax[o] = wm * (float)f;
        // centre: c >= K with (K/2) ln c - c/2 - lg = nr, by the fixed point c = 2 y + K ln c (y = -(nr + lg) >= K/2 - (K/2) ln K);
        // a rough solution is enough (kp below makes the weight formula exact for whatever centre is used)
        const float y2 = (float)(-2.0 * (nr + src.lp.lg_full));
        float cc = fmaxf(y2, (float)K);
#pragma unroll
        for (int it = 0; it < 4; ++it) cc = fmaxf(fmaf((float)(K * 0.6931471805599453), __builtin_amdgcn_logf(cc), y2), (float)K);
        const double nc = uniform_d((double)cc);
        ws.ref[o] = nr; ws.cref[o] = nc; ws.rcr[o] = uniform_f(__builtin_amdgcn_rcpf(cc));
        ws.kp[o] = uniform_f((float)((uniform_d(src.lnl_of_chi2(nc)) - nr) * 1.4426950408889634));
    } else {
        ws.wmax[o] = wm;                                         // every lane filters against the wave-wide best
    }
}

// POSW: the tile carries the models' histogram indices (int32 words behind the records) and a candidate
// records ITS index instead of its model number: the PDF stage then needs no label gathers.
template <class SRC, int TW, bool TAIL, int TL, bool POSW>
__device__ __forceinline__ void fused_tile_w(const SRC& src, const FastTabs& tb, const double* cur, const double* objs,
                                             int jt0, int M, int lane, float thrf, Cand* buf, int64_t cap,
                                             WState<TW>& ws) {
    constexpr int OD = SRC::OBJ_DOUBLES;
    constexpr int WP = SRC::WPOW;
    static_assert(WP >= 1 && WP <= 6, "weight-space path: chi2^(1/2) ... chi2^3 (4-8 exact bands)");
    // append the lanes in `c` to object o's candidate list: ballot + mbcnt compaction, one 16-B store each
    auto append = [&](int o, bool c, double chi2, int j) {
        const unsigned long long mask = __ballot(c);
        if (mask) {                                                   // wave-uniform
            const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (c) { Cand e; e.lnl = chi2; e.j = j; e.pad = 0; buf[(size_t)o * cap + ws.cnt[o] + pre] = e; }
            ws.cnt[o] += __builtin_popcountll(mask);
        }
    };
#pragma unroll 1
    for (int s = 0; s < TL / 64; ++s) {
        const int j = jt0 + s * 64 + lane;
        typename SRC::MR m;
        src.template load_model_lds<TL>(cur, s * 64 + lane, m);
        const int tag = POSW ? reinterpret_cast<const int32_t*>(cur + SRC::RW * TL)[s * 64 + lane] : j;     // what a candidate records
        double c2[TW];
        float t[TW];
        bool over = false;
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            typename SRC::OR ob;
            src.load_obj_lds(objs + o * OD, ob);
            c2[o] = src.chi2_of(ob, m);
            if (TAIL) c2[o] = (j < M) ? c2[o] : 1e30;           // pad lanes: weight 0 (1e30 converts to a finite float)
            const float dcf = (float)(c2[o] - ws.cref[o]);      // fp64 difference, then fp32
            const float l2 = __builtin_amdgcn_logf((float)c2[o] * ws.rcr[o]);     // log2(chi2 / cref); chi2 == 0 (self match): -inf, weight 0
            t[o] = fmaf(l2, 0.5f * WP, fmaf(dcf, -0.72134752f, ws.kp[o]));
            over |= !(t[o] <= 60.f);                            // too large or not a number
        }
        if (__any(over)) {                    // an object's first step; a new best by > e^40; chi2 beyond the fp32 range
#pragma unroll
            for (int o = 0; o < TW; ++o) {
                if (__any(!(t[o] <= 60.f))) { // per object: an object's arithmetic never depends on its wave-mates
                    w_rebase(src, ws, o, &c2[o]);
                    const float dcf = (float)(c2[o] - ws.cref[o]);
                    const float l2 = __builtin_amdgcn_logf((float)c2[o] * ws.rcr[o]);
                    t[o] = fmaf(l2, 0.5f * WP, fmaf(dcf, -0.72134752f, ws.kp[o]));
                    const bool big = !(t[o] <= 60.f);         // still out of range (every weight of the step is <= ~1 now): exact treatment
                    append(o, big, c2[o], tag);
                    t[o] = big ? -INFINITY : t[o];            // ... and no weight in the loop
                }
            }
        }
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            const float w = __builtin_amdgcn_exp2f(t[o]);
            ws.wmax[o] = fmaxf(ws.wmax[o], w);
            const bool c = w > ws.wmax[o] * thrf;
            if (!c) ws.s[o] += w;                                    // candidates are summed exactly by the PDF stage
            append(o, c, c2[o], tag);
        }
        ++ws.tick;
        if ((ws.tick & 15) == 0) {
#pragma unroll
            for (int o = 0; o < TW; ++o) { ws.S[o] += (double)ws.s[o]; ws.s[o] = 0.f; }
            if ((ws.tick & 63) == 0) {
#pragma unroll
                for (int o = 0; o < TW; ++o) w_rebase(src, ws, o, nullptr);
            }
        }
    }
}

// ---- PDF stage of the class-sorted dictionary stack (k_fused<..., MC>) ----------------------------
// The model records this kernel walked are ordered by dictionary class, so the object's candidate list
// is grouped by class.  One LDS histogram row (mc_gp = G + 2 W0 entries, W0 = the widest half-width
// present): selected weights go in at their label index -- ONE ds_add_f64 per stacked model, as in the
// single-kernel form -- and when the class of the entries changes the row is divided by the class's
// edge-truncated kernel mass per index, convolved with the class's kernel into the lane's registers
// (12 outputs per lane: G <= 768) and cleared.  pdf.py:599-620 adds each model's whole window instead
// (~56 LDS adds per model at the benchmark's label errors); the sums agree to rounding order.
// Weights, threshold rule, evidence and the hand-back of failed objects are those of the weight-space
// stage in k_fused (candidates carry chi2; selection against the loop's estimate of the maximum,
// entries within +-DEL of the threshold settled by a second walk with the exact evidence).
template <class SRC, int U>
__device__ __forceinline__ void pdf_stage_mc(const SRC& src, const KdeView& kv, const FastTabs& tb, double* row, const Cand* cb, int n,
                                             const double* res, double wt_thresh, double lthr, int normalize, double* lmap,
                                             double* levid, double* pdfs, int* redo, int64_t i, int lane) {
    constexpr int NACC = 12;
    constexpr double DEL = 1e-4;
    const int G = (int)kv.G, Gp = kv.mc_gp, W0 = kv.mc_w0;
    const double ref = uniform_d(res[0]), sn = uniform_d(res[1]), mxa = uniform_d(res[2]);
    double out[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) out[a] = 0.0;
    for (int k = lane; k < Gp; k += 64) row[k] = 0.0;
    int ccur = -1;                                                // class rank whose weights sit in the row (wave-uniform)
    // A lane owns NACC CONSECUTIVE outputs (t = NACC lane + a), so that the row values of its window are
    // reused across its outputs: tap h needs row[base + a + h] for a = 0..NACC-1, a window that slides by one
    // entry per tap -- ONE ds_read per lane and tap (the 12-slot window rotates through registers, the tap
    // loop is unrolled by 12 so that the rotation is a renaming) instead of one per output and tap; with
    // outputs strided across lanes the convolution of ~23 classes per object was bound by the CU's LDS pipe.
    auto flush = [&]() {
        const int wc = kv.mc_width[ccur], w2 = 2 * wc, sh = W0 - wc;
        const double* nt = kv.mc_norm + (size_t)ccur * Gp;
        const double* kr = kv.kern + kv.mc_off[ccur];
        const double ka = (lane <= w2) ? kr[lane] : 0.0;          // w2 + 1 <= 127 taps in two registers across the wave
        const double kb = (lane + 64 <= w2) ? kr[lane + 64] : 0.0;
        for (int k = lane; k < Gp; k += 64) row[k] = row[k] / nt[k];
        const int kal = __double2loint(ka), kah = __double2hiint(ka), kbl = __double2loint(kb), kbh = __double2hiint(kb);
        const double* r0 = row + sh + NACC * min(lane, (G - 1) / NACC);      // lanes past G repeat the last lane's reads (unused sums); reads end < Gp + 12
        double win[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) win[a] = r0[a];
        for (int h = 0; h <= w2; h += NACC) {
#pragma unroll
            for (int u = 0; u < NACC; ++u) {
                if (h + u <= w2) {                                // wave-uniform
                    const int q = w2 - (h + u);
                    const double tap = (q < 64) ? __hiloint2double(__builtin_amdgcn_readlane(kah, q), __builtin_amdgcn_readlane(kal, q))
                                                : __hiloint2double(__builtin_amdgcn_readlane(kbh, q - 64), __builtin_amdgcn_readlane(kbl, q - 64));
#pragma unroll
                    for (int a = 0; a < NACC; ++a) out[a] = fma(win[(a + u) % NACC], tap, out[a]);
                    win[u] = r0[NACC + h + u];                    // the entry that left the window makes room for the next one
                }
            }
        }
        for (int k = lane; k < Gp; k += 64) row[k] = 0.0;
    };
    // add the selected entries of one 64-entry block; entries are in class order, so the classes of a
    // block are visited in lane order and each one is complete when the next begins
    auto stack = [&](bool sel, int tag, double w) {
        unsigned long long rem = __ballot(sel);
        while (rem) {
            const int r = __builtin_amdgcn_readlane(tag, __builtin_ctzll(rem)) >> 10;
            if (r != ccur) { if (ccur >= 0) flush(); ccur = r; }
            const bool mine = sel && (tag >> 10) == r;
            if (mine) unsafeAtomicAdd(&row[tag & 1023], w);
            rem &= ~__ballot(mine);
        }
    };
    double lbest = -INFINITY, sc = 0.0;
    bool anyamb = false;
    for (int c0 = 0; c0 < n; c0 += 64 * U) {
        Cand e[U]; bool in[U], sel[U]; double w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int k = c0 + u * 64 + lane; in[u] = k < n; e[u] = cb[in[u] ? k : 0]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double l = in[u] ? src.lnl_of_chi2(e[u].lnl) : -INFINITY;
            lbest = fmax(lbest, l);
            w[u] = exp_neg(l - ref, tb);
            sc += in[u] ? w[u] : 0.0;
            const double d = l - mxa;
            sel[u] = in[u] && (d > lthr + DEL);
            anyamb |= in[u] && !sel[u] && (d >= lthr - DEL);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) stack(sel[u], e[u].j, w[u]);
    }
    lbest = wave_max(lbest);
    const double stot = sn + wave_sum(sc);
    const double le = ref + log_pos(stot, tb);
    if (__any(anyamb)) {                                          // wave-uniform, rare: a second pass over the classes
        if (ccur >= 0) { flush(); ccur = -1; }
        const double thr = wt_thresh * exp_neg(lbest - le, tb);   // wt_thresh * max(wt)
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int k = c0 + lane;
            const bool in1 = k < n;
            const Cand e1 = cb[in1 ? k : 0];
            const double l = in1 ? src.lnl_of_chi2(e1.lnl) : -INFINITY;
            const double d = l - mxa;
            const bool amb = in1 && !(d > lthr + DEL) && (d >= lthr - DEL);
            const bool s1 = amb && (exp_neg(l - le, tb) > thr);   // strict, pdf.py:591
            stack(s1, e1.j, exp_neg(l - ref, tb));
        }
    }
    if (ccur >= 0) flush();
    const bool ok = (le - le == 0.0) && n > 0;
    if (lane == 0) {
        if (lmap) lmap[i] = lbest;
        if (levid) levid[i] = le;
        if (!ok) redo[1 + atomicAdd(redo, 1)] = (int)i;
    }
    double* o = pdfs + i * kv.G;
    double tot = 0.0;
#pragma unroll
    for (int a = 0; a < NACC; ++a) if (NACC * lane + a < G) tot += out[a];
    tot = wave_sum(tot);
    const double f = normalize ? 1.0 / tot : exp_neg(ref - le, tb);
#pragma unroll
    for (int a = 0; a < NACC; ++a)
        if (NACC * lane + a < G) o[NACC * lane + a] = !ok ? (double)NAN : (normalize ? out[a] / tot : out[a] * f);
}

// HO: instantiation for label sets with ONE dictionary kernel (histogram + one convolution: every
// demo of the reference) -- the window-scatter code of the other KDE forms is not compiled in, so
// that it cannot cost the hot kernel registers.
// MC (with WM and HO): many dictionary widths through class-sorted records -- see the PDF stage below.
template <class SRC, int TW, int NW, bool WM, bool HO, bool MC = false>
__global__ __launch_bounds__(NW * 64) void k_fused(SRC src_, const KdeView* __restrict__ kvp, int acc_stride, int64_t N_,
                                                    int M, double wt_thresh, int normalize,
                                                    Cand* __restrict__ cand, int64_t cap,
                                                    double* __restrict__ lmap, double* __restrict__ levid,
                                                    double* __restrict__ pdfs, const int* __restrict__ omap,
                                                    int* __restrict__ redo, const int* __restrict__ ndev) {
    // ndev: the object count lives on the device (the sweep over the objects a weight-space launch
    // handed back: usually none, and the host never waits to find out)
    const int64_t N = ndev ? (int64_t)*ndev : N_;
    // LDS.  Static: the two model tile buffers (separate arrays: the compiler then knows that the
    // LDS-DMA filling one does not touch the other, and does not make the reads of the current tile
    // wait for the copy of the next), the per-object results a wave parks, the log / exp tables, the
    // parked object rows.  Outside the model loop the tile buffers hold the waves' PDF rows (half
    // of the waves in each); grids too long for that get dynamic LDS for the rows instead.
    constexpr bool POSW = WM && HO;                               // weight-space body + one dictionary kernel: index words ride in the tile
    constexpr int TILE = SRC::template tile_len<NW>(), TDR = SRC::template tile_doubles<TILE>(), TD = TDR + (POSW ? TILE / 2 : 0);
    constexpr int NCH = TD / 2, NT = NW * 64;
    constexpr int CPT = (NCH + NT - 1) / NT;                      // staging chunks per thread
    constexpr int OD = SRC::OBJ_DOUBLES;
    constexpr int HW = (NW + 1) / 2;                              // waves whose rows share a tile buffer
    __shared__ __attribute__((aligned(16))) double tileA[TD];
    __shared__ __attribute__((aligned(16))) double tileB[TD];
    __shared__ __attribute__((aligned(16))) double s_res[NW * TW * FZ_RES];
    __shared__ __attribute__((aligned(16))) double s_tabs[FZ_TABS_DOUBLES];
    __shared__ __attribute__((aligned(16))) double s_objs[NW * TW * OD];
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    const int64_t gw = (int64_t)blockIdx.x * NW + wave;
    const int64_t ngroups = (N + TW - 1) / TW;
    const int64_t nrounds = (ngroups + nwaves - 1) / nwaves;      // same for every wave: barriers stay aligned
    const int ntiles = (M + TILE - 1) / TILE;
    double* row = (HW * acc_stride <= TD) ? ((wave < HW ? tileA : tileB) + (size_t)(wave < HW ? wave : wave - HW) * acc_stride)
                                          : smem + (size_t)wave * acc_stride;      // valid only outside the model loop
    double* res = s_res + wave * (TW * FZ_RES);                   // {lmap, levid, max, count} per object (WM: {ref, sum, ~max, count})
    double* tabs = s_tabs;
    double* objs = s_objs + wave * (TW * OD);
    SRC src = src_;
    src.tb = stage_tabs(tabs, tid, NT);
    const FastTabs tb = src.tb;
    Cand* buf = cand + (size_t)gw * TW * cap;
    const int32_t* posw = POSW ? (MC ? kvp->mc_tag : kvp->pos) : nullptr;
    const double lt = (wt_thresh > 0.0) ? log(wt_thresh) - 1e-3 : -INFINITY;
    const float thrf = uniform_f((wt_thresh > 0.0) ? (float)(wt_thresh * 0.99) : 0.f);          // fp32 screen: a 1 % margin below the exact threshold
    const double lthr = (wt_thresh > 0.0) ? log(wt_thresh) : -INFINITY;
    const bool dp = src.lp.dim_prior != 0;

    for (int64_t rnd = 0; rnd < nrounds; ++rnd) {
        const int64_t g = gw + rnd * nwaves;
        const bool work = g < ngroups;                            // wave-uniform
        const int64_t i0 = work ? g * TW : 0;
        FusedState<TW> fs;
        WState<TW> ws;
        fs.firstnan = 0; fs.anynan = 0; fs.tick = 0;
        if (WM) w_init(ws);
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            if (!WM) { ms_init(fs.st[o]); fs.cnt[o] = 0; }
            // omap: the launch covers a subset of the chunk's objects (N of them), listed by index
            const int64_t oslot = i0 + o < N ? i0 + o : N - 1;
            src.park_obj(omap ? (int64_t)omap[oslot] : oslot, objs + o * OD, lane);
        }
        // tile 0 -> LDS.  Two staging forms, chosen per kernel body from measurements
        // (profiles/README.md): the ln-space bodies copy tiles with the LDS-DMA form of the load
        // (global_load_lds_dwordx4: 16 B per lane, destination = wave-uniform base + lane * 16,
        // which is exactly this contiguous copy) -- no staging registers in kernels that sit at
        // their VGPR cap, no ds_write pass, drained by __syncthreads() (vmcnt(0)); the
        // weight-space body keeps register staging (loads issued before, parked after the
        // compute of the current tile), which measured 2-4 % faster there.
        constexpr bool GLDS = true;       // (register staging measured 2-4 % faster for the old fp64 weight-space body; the fp32-tail body spills it)
        // LDS-DMA copy of one tile: its segments (records | mask words | index words) are contiguous in
        // HBM and in LDS, so each is a run of 1-KB wave-instructions; the source address is a wave-uniform
        // base plus the thread's 16-byte slot, which keeps one 32-bit offset register live instead of a
        // 64-bit pointer per chunk (those spilled, and every reload made the copies wait for each other)
        auto stage_seg = [&](const void* base, int nchunks, double* dst) {
            const uint32_t slot = (uint32_t)tid * 16u;
#pragma unroll
            for (int q = 0; q < (nchunks + NT - 1) / NT; ++q) {
                if (tid + q * NT < nchunks)
                    __builtin_amdgcn_global_load_lds((gbl_cvoid*)(reinterpret_cast<const char*>(base) + (size_t)q * NT * 16 + slot),
                                                     (lds_void*)(dst + 2 * (q * NT + wave * 64)), 16, 0, 0);
            }
        };
        auto stage_tile = [&](int tile, double* dstbuf) {
            constexpr int NREC = SRC::RW * TILE / 2;                 // 16-byte chunks of the records
            const double* rec = (SRC::LMODE == 0) ? src.mv.rec0 : src.mv.rec1;
            stage_seg(rec + (int64_t)tile * (TILE * SRC::RW), NREC, dstbuf);
            if constexpr (TDR > SRC::RW * TILE) stage_seg(src.mv.bits + (int64_t)tile * TILE, TILE / 4, dstbuf + SRC::RW * TILE);
            if constexpr (POSW) stage_seg(posw + (int64_t)tile * TILE, TILE / 4, dstbuf + TDR);
        };
        double2 stage[CPT];
        if (GLDS) {
            stage_tile(0, tileA);
        } else {
#pragma unroll
            for (int q = 0; q < CPT; ++q) { const int ch = tid + q * NT; if (ch < NCH) stage[q] = src.template tile_chunk<TILE>(0, ch); }
#pragma unroll
            for (int q = 0; q < CPT; ++q) { const int ch = tid + q * NT; if (ch < NCH) reinterpret_cast<double2*>(tileA)[ch] = stage[q]; }
        }
        __syncthreads();
        // one tile: start the copy of the next one into the other buffer, evaluate this one
        auto run_tile = [&](const double* cur, double* nxt, int t) {
            const bool more = t + 1 < ntiles;
            if (more) {                                           // next tile: in flight while the current one is used
                if (GLDS) stage_tile(t + 1, nxt);
                else {
#pragma unroll
                    for (int q = 0; q < CPT; ++q) { const int ch = tid + q * NT; if (ch < NCH) stage[q] = src.template tile_chunk<TILE>(t + 1, ch); }
                }
            }
            if (work && WM) {
                if constexpr (WM) {
                    if (more) fused_tile_w<SRC, TW, false, TILE, POSW>(src, tb, cur, objs, t * TILE, M, lane, thrf, buf, cap, ws);
                    else fused_tile_w<SRC, TW, true, TILE, POSW>(src, tb, cur, objs, t * TILE, M, lane, thrf, buf, cap, ws);
                }
            } else if (work) {                                    // unswitched on (dim_prior, last tile)
                if (more) {
                    if (dp) fused_tile<SRC, TW, 1, false, TILE>(src, tb, cur, objs, t * TILE, M, lane, lt, buf, cap, fs);
                    else fused_tile<SRC, TW, 0, false, TILE>(src, tb, cur, objs, t * TILE, M, lane, lt, buf, cap, fs);
                } else {
                    if (dp) fused_tile<SRC, TW, 1, true, TILE>(src, tb, cur, objs, t * TILE, M, lane, lt, buf, cap, fs);
                    else fused_tile<SRC, TW, 0, true, TILE>(src, tb, cur, objs, t * TILE, M, lane, lt, buf, cap, fs);
                }
            }
            if (more && !GLDS) {                                  // ... and parked in the other buffer late
#pragma unroll
                for (int q = 0; q < CPT; ++q) { const int ch = tid + q * NT; if (ch < NCH) reinterpret_cast<double2*>(nxt)[ch] = stage[q]; }
            }
            __syncthreads();
        };
        if constexpr (WM) {
            // two tiles per trip, so that each call names its buffers statically (see the LDS note above)
            for (int t = 0; t < ntiles; t += 2) {
                run_tile(tileA, tileB, t);
                if (t + 1 < ntiles) run_tile(tileB, tileA, t + 1);
            }
        } else {
            for (int t = 0; t < ntiles; ++t) run_tile((t & 1) ? tileB : tileA, (t & 1) ? tileA : tileB, t);
        }
        // per-object max / evidence (wave reductions), parked in LDS so that the PDF
        // stage below can be ONE loop body instead of TW inlined copies
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            if (WM) {
                if constexpr (WM) w_rebase(src, ws, o, nullptr);     // the steps since the last re-base
                const double mxa = ws.ref[o] + log_pos((double)wave_maxf(ws.wmax[o]), tb);   // ~1e-6: only classifies candidates
                const double sn = wave_sum(ws.S[o] + (double)ws.s[o]);
                if (lane == 0) { res[o * FZ_RES + 0] = ws.ref[o]; res[o * FZ_RES + 1] = sn; res[o * FZ_RES + 2] = mxa; res[o * FZ_RES + 3] = (double)ws.cnt[o]; }
                continue;
            }
            const bool fn = __any((fs.firstnan >> o) & 1u);
            const bool an = __any((fs.anynan >> o) & 1u);
            const double mx = wave_max(fs.st[o].m);
            const double ss = wave_sum(fs.st[o].s * exp_neg(fs.st[o].m - mx, tb));
            if (lane == 0) {
                res[o * FZ_RES + 0] = fn ? (double)NAN : mx;
                res[o * FZ_RES + 1] = an ? (double)NAN : (mx == INFINITY ? (double)INFINITY : mx + log(ss));
                res[o * FZ_RES + 2] = mx;
                res[o * FZ_RES + 3] = (double)fs.cnt[o];
            }
        }
        // Tiles are dead from here on; each wave reuses its slice of the LDS as a PDF
        // row.  The candidate entries were written by other lanes of this wave: make
        // them visible to this CU's loads (stores drained, vector L1 invalidated).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (work) {
            const KdeView kv = *kvp;
#pragma unroll 1
            for (int o = 0; o < TW; ++o) {
                if (i0 + o >= N) break;
                const int64_t i = omap ? (int64_t)omap[i0 + o] : i0 + o;
                const int n = __builtin_amdgcn_readfirstlane((int)res[o * FZ_RES + 3]);
                const Cand* cb = buf + (size_t)o * cap;
                // A wave walks its object's list alone, so each trip is two dependent memory
                // round trips (the entries, then the labels of the selected ones): U 64-entry
                // blocks are kept in flight per trip to overlap them.
                constexpr int U = (HO && !MC) ? 8 : 4;
                if constexpr (MC) {
                    pdf_stage_mc<SRC, U>(src, kv, tb, row, cb, n, res + o * FZ_RES, wt_thresh, lthr, normalize, lmap, levid, pdfs, redo, i, lane);
                } else if constexpr (WM) {
                    // Candidates carry their fp64 chi2.  One walk: exact ln-like, exact maximum, the
                    // candidates' exact share of the evidence, and the stack -- weights relative to the
                    // loop's reference (an upper bound of every ln-like), selection against the loop's
                    // fp32 estimate of the maximum where that is decisive (outside +-DEL of the
                    // threshold); the few entries inside the band are settled by a second walk with
                    // the exact maximum and evidence, by the reference's own rule (pdf.py:591).
                    const double ref = uniform_d(res[o * FZ_RES + 0]), sn = uniform_d(res[o * FZ_RES + 1]), mxa = uniform_d(res[o * FZ_RES + 2]);
                    constexpr double DEL = 1e-4;
                    for (int k = lane; k < acc_stride; k += 64) row[k] = 0.0;
                    double lbest = -INFINITY, sc = 0.0;
                    bool anyamb = false;
                    for (int c0 = 0; c0 < n; c0 += 64 * U) {
                        Cand e[U]; bool in[U], sel[U]; double w[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int k = c0 + u * 64 + lane; in[u] = k < n; e[u] = cb[in[u] ? k : 0]; }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const double l = in[u] ? src.lnl_of_chi2(e[u].lnl) : -INFINITY;
                            lbest = fmax(lbest, l);
                            w[u] = exp_neg(l - ref, tb);
                            sc += in[u] ? w[u] : 0.0;
                            const double d = l - mxa;
                            sel[u] = in[u] && (d > lthr + DEL);
                            anyamb |= in[u] && !sel[u] && (d >= lthr - DEL);
                        }
                        if (HO) {                 // entries carry the histogram index; the division by the kernel mass waits for the finalize
#pragma unroll
                            for (int u = 0; u < U; ++u) if (sel[u]) unsafeAtomicAdd(&row[e[u].j + kv.w0], w[u]);
                        } else if (kv.kmode == KDE_HIST) {
                            int p[U]; double nr[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) { const int j = sel[u] ? e[u].j : 0; p[u] = kv.pos[j]; nr[u] = kv.norm[j]; }
#pragma unroll
                            for (int u = 0; u < U; ++u) if (sel[u]) unsafeAtomicAdd(&row[p[u] + kv.w0], w[u] / nr[u]);
                        } else {
#pragma unroll
                            for (int u = 0; u < U; ++u) kde_scatter(kv, row, sel[u], w[u], e[u].j, lane);
                        }
                    }
                    lbest = wave_max(lbest);
                    const double stot = sn + wave_sum(sc);
                    const double le = ref + log_pos(stot, tb);
                    if (__any(anyamb)) {                          // wave-uniform, rare
                        const double thr = wt_thresh * exp_neg(lbest - le, tb);          // wt_thresh * max(wt)
                        for (int c0 = 0; c0 < n; c0 += 64) {
                            const int k = c0 + lane;
                            const bool in1 = k < n;
                            const Cand e1 = cb[in1 ? k : 0];
                            const double l = in1 ? src.lnl_of_chi2(e1.lnl) : -INFINITY;
                            const double d = l - mxa;
                            const bool amb = in1 && !(d > lthr + DEL) && (d >= lthr - DEL);
                            const bool s1 = amb && (exp_neg(l - le, tb) > thr);          // strict, pdf.py:510/591
                            const double w1 = exp_neg(l - ref, tb);
                            if (HO) {
                                if (s1) unsafeAtomicAdd(&row[e1.j + kv.w0], w1);
                            } else if (kv.kmode == KDE_HIST) {
                                if (s1) unsafeAtomicAdd(&row[kv.pos[e1.j] + kv.w0], w1 / kv.norm[e1.j]);
                            } else kde_scatter(kv, row, s1, w1, e1.j, lane);
                        }
                    }
                    // nothing recorded / no weight at all (every chi2 so far below the mode that the fp32
                    // weights underflow): the object is handed to the fp64 ln-space body by the host
                    const bool ok = (le - le == 0.0) && n > 0;
                    if (lane == 0) {
                        if (lmap) lmap[i] = lbest;
                        if (levid) levid[i] = le;
                        if (!ok) redo[1 + atomicAdd(redo, 1)] = (int)i;
                    }
                    kde_finalize<HO>(kv, row, ok, normalize, pdfs + i * kv.G, lane, exp_neg(ref - le, tb), HO);
                } else {
                const double lm = res[o * FZ_RES + 0], le = res[o * FZ_RES + 1], mx = res[o * FZ_RES + 2];
                if (lane == 0) { if (lmap) lmap[i] = lm; if (levid) levid[i] = le; }
                const bool ok = (le - le == 0.0);
                if (ok) {
                    for (int k = lane; k < acc_stride; k += 64) row[k] = 0.0;
                    const double thr = wt_thresh * exp_neg(mx - le, tb);
                    for (int c0 = 0; c0 < n; c0 += 64 * U) {
                        Cand e[U]; bool in[U], sel[U]; double w[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { const int k = c0 + u * 64 + lane; in[u] = k < n; e[u] = cb[in[u] ? k : 0]; }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            w[u] = exp_neg(e[u].lnl - le, tb);
                            sel[u] = in[u] && (w[u] > thr);
                        }
                        if (HO || kv.kmode == KDE_HIST) {
                            int p[U]; double nr[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) { const int j = sel[u] ? e[u].j : 0; p[u] = kv.pos[j]; nr[u] = kv.norm[j]; }
#pragma unroll
                            for (int u = 0; u < U; ++u) if (sel[u]) unsafeAtomicAdd(&row[p[u] + kv.w0], w[u] / nr[u]);
                        } else {
#pragma unroll
                            for (int u = 0; u < U; ++u) kde_scatter(kv, row, sel[u], w[u], e[u].j, lane);
                        }
                    }
                }
                kde_finalize<HO>(kv, row, ok, normalize, pdfs + i * kv.G, lane);
                }
            }
        }
        __syncthreads();                                          // rows -> tiles again
    }
}

// ---- single pass over a stored (N,M) ln-weight plane -> PDFs -------------------
// BruteForce._predict (bruteforce.py:303-372) with logwt = a materialised plane, and the tail
// of mode C.  The two-pass form (k_stats, then k_kde) reads every row twice; here a wave reads
// its object's row ONCE (VEC = 2: 16 B per lane; VEC = 1 for rows that do not start 16-B aligned;
// four loads per trip, the next trip's loads in flight while the current one is used), sums the
// weights exp(l - ref) against a wave-uniform reference that is re-based (exactly: the partial
// sums are rescaled) before any entry could leave the exponent range, tracks the per-lane best and
// appends the entries within the weight threshold of it -- a superset of the finally selected
// ones -- to its private list (k_fused's hand-off: ballot + mbcnt compaction, one 16-B store
// each).  The list is then walked with the exact evidence and the exact strict threshold, as in
// k_fused.  Waves are independent: no block barrier after the table staging.
template <int VEC>
struct PlaneState {
    double ref, s, m;          // wave-uniform reference; per-lane sum of exp(l - ref) and best l
    float s32;                 // X32: the lane's fp32 partial sum of the non-candidates since the last flush into s
    bool anynan;
    int cnt;
};
template <int VEC, int U, bool TAIL>
__device__ __forceinline__ void plane_load(const double* r, int jb, int M, int lane, double (&l)[U][VEC]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = jb + (u * 64 + lane) * VEC;
        if (VEC == 2) {
            fz_d2 v = {-INFINITY, -INFINITY};
            if (!TAIL || j < M) v = __builtin_nontemporal_load(reinterpret_cast<const fz_d2*>(r + j));   // M is even
            l[u][0] = v.x; l[u][VEC - 1] = v.y;
        } else {
            l[u][0] = (!TAIL || j < M) ? __builtin_nontemporal_load(r + j) : -INFINITY;
        }
    }
}
// X32: the weight of an entry that cannot be stacked -- below wt_thresh of the best seen so far -- is formed and summed in
// fp32 (difference l - ref in fp64 first, then one v_exp_f32: 5 instructions instead of the 14 of the fp64 exp); the
// candidates' weights are left to the list walk, which forms them in fp64.  What reaches lmap, the PDFs and the stacked
// weights is fp64 either way; the ln-evidence carries the fp32 remainder (~1e-9), as in the fused kernels.  X32 = false
// (kde_opts.exact_evidence): every weight in fp64.
template <int VEC, int U, bool TAIL, bool X32>
__device__ __forceinline__ void plane_trip(const double (&l)[U][VEC], int jb, int M, int lane, double lt,
                                           const FastTabs& tb, Cand* buf, PlaneState<VEC>& ps) {
    double mm = ps.m;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < VEC; ++q) { mm = vmax_raw(mm, l[u][q]); ps.anynan |= (l[u][q] != l[u][q]); }   // nan never becomes the best
    if (__any(mm - ps.ref > (X32 ? 40.0 : 500.0))) {   // rare (first trip; the best jumps): re-base on the wave-wide best
        const double nr = wave_max(mm);
        if (X32) { ps.s += (double)ps.s32; ps.s32 = 0.f; }
        ps.s *= exp_neg(ps.ref - nr, tb);    // ref = -inf: s is still 0
        ps.ref = nr;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int j = jb + (u * 64 + lane) * VEC + q;
            const double lv = l[u][q];
            ps.m = vmax_raw(ps.m, lv);
            bool c = lv >= ps.m + lt;                                 // >=: lt may be absorbed at huge |lnl|; false for nan
            if (TAIL) c = c && (j < M);
            if (X32) {
                // fp64 difference, then fp32: exp2((l - ref) log2 e); nan -> 0 (nans are flagged above), -inf -> 0
                float t = (float)(lv - ps.ref) * 1.4426950408889634f;
                asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(-150.f));
                const float w = __builtin_amdgcn_exp2f(t);
                ps.s32 += c ? 0.f : w;                                // the candidates are summed exactly by the list walk
            } else {
                ps.s += exp_clamped(lv - ps.ref, tb);                 // nan, -inf and far tails -> ~1e-304
            }
            const unsigned long long mask = __ballot(c);
            if (mask) {                                               // wave-uniform
                const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                    __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (c) { Cand e; e.lnl = lv; e.j = j; e.pad = 0; buf[ps.cnt + pre] = e; }
                ps.cnt += __builtin_popcountll(mask);
            }
        }
    }
    if (X32) { ps.s += (double)ps.s32; ps.s32 = 0.f; }               // U * VEC terms per flush
}

template <int NW, int VEC, bool HO, bool X32>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_plane_fused(const double* __restrict__ plane, int64_t ld,
                                                          const KdeView* __restrict__ kvp, int acc_stride, int64_t N,
                                                          int M, double wt_thresh, int normalize,
                                                          Cand* __restrict__ cand, int64_t cap,
                                                          double* __restrict__ lmap, double* __restrict__ levid,
                                                          double* __restrict__ pdfs) {
    extern __shared__ double smem[];
    constexpr int U = 4, STEP = 64 * VEC * U;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const FastTabs tb = stage_tabs(smem, tid, NW * 64);
    double* row = smem + FZ_TABS_DOUBLES + (size_t)wave * acc_stride;
    __syncthreads();
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    const int64_t gw = (int64_t)blockIdx.x * NW + wave;
    Cand* buf = cand + (size_t)gw * cap;
    const double lt = (wt_thresh > 0.0) ? log(wt_thresh) - 1e-3 : -INFINITY;
    const KdeView kv = *kvp;
    for (int64_t i = gw; i < N; i += nwaves) {
        const double* r = plane + i * ld;
        PlaneState<VEC> ps;
        ps.ref = -INFINITY; ps.s = 0.0; ps.m = -INFINITY; ps.anynan = false; ps.cnt = 0; ps.s32 = 0.f;
        double l[U][VEC], ln[U][VEC];
        if (STEP <= M) plane_load<VEC, U, false>(r, 0, M, lane, l); else plane_load<VEC, U, true>(r, 0, M, lane, l);
        const bool firstnan = (lane == 0) && (l[0][0] != l[0][0]);
        int tick = 0;
        for (int jb = 0; jb < M; jb += STEP) {
            const int jn = jb + STEP;
            if (jn + STEP <= M) plane_load<VEC, U, false>(r, jn, M, lane, ln);
            else if (jn < M) plane_load<VEC, U, true>(r, jn, M, lane, ln);
            if (jn <= M) plane_trip<VEC, U, false, X32>(l, jb, M, lane, lt, tb, buf, ps);
            else plane_trip<VEC, U, true, X32>(l, jb, M, lane, lt, tb, buf, ps);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < VEC; ++q) l[u][q] = ln[u][q];
            // every 8 trips the lanes take the wave-wide best, so that the filter works against the
            // best any lane has seen
            if ((++tick & 7) == 0) ps.m = wave_max(ps.m);
        }
        const int cnt = ps.cnt;
        const bool anynan = ps.anynan;
        const double mx = wave_max(ps.m);
        const double ss = wave_sum(ps.s);
        const bool fn = __any(firstnan), an = __any(anynan);
        const double lm = fn ? (double)NAN : mx;                       // builtin max: NaN only if first
        // the entries were written by other lanes of this wave: stores drained, vector L1 invalidated
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double sx = ss;
        if (X32) {                                                     // + the candidates' exact weights (a first, short walk)
            double sc = 0.0;
            const int n0 = __builtin_amdgcn_readfirstlane(cnt);
            for (int c0 = lane; c0 < n0; c0 += 64) sc += exp_clamped(buf[c0].lnl - ps.ref, tb);
            sx += wave_sum(sc);
        }
        const double le = an ? (double)NAN : (mx == INFINITY ? (double)INFINITY : ps.ref + log(sx));
        if (lane == 0) { if (lmap) lmap[i] = lm; if (levid) levid[i] = le; }
        const bool ok = (le - le == 0.0);                              // finite evidence
        if (ok) {
            for (int k = lane; k < acc_stride; k += 64) row[k] = 0.0;
            const double thr = wt_thresh * exp_neg(mx - le, tb);       // wt_thresh * max(wt)
            const int n = __builtin_amdgcn_readfirstlane(cnt);
            for (int c0 = 0; c0 < n; c0 += 64 * U) {
                Cand e[U]; bool in[U], sel[U]; double w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int k = c0 + u * 64 + lane; in[u] = k < n; e[u] = buf[in[u] ? k : 0]; }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    w[u] = exp_neg(e[u].lnl - le, tb);
                    sel[u] = in[u] && (w[u] > thr);                    // strict, pdf.py:510/591
                }
                if (HO || kv.kmode == KDE_HIST) {
                    int p[U]; double nr[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { const int j = sel[u] ? e[u].j : 0; p[u] = kv.pos[j]; nr[u] = kv.norm[j]; }
#pragma unroll
                    for (int u = 0; u < U; ++u) if (sel[u]) unsafeAtomicAdd(&row[p[u] + kv.w0], w[u] / nr[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) kde_scatter(kv, row, sel[u], w[u], e[u].j, lane);
                }
            }
        }
        kde_finalize<HO>(kv, row, ok, normalize, pdfs + i * kv.G, lane);
    }
}

}  // namespace fz
