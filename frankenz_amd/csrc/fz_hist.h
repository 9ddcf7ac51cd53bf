// k_hist: fused fit_predict in ONE pass over the models with nothing handed through HBM.
// (single dictionary kernel, dimensionality prior on, mask-free exact band counts: the weight-space
//  conditions of k_fused -- every demo configuration of the reference; bruteforce.py:602-631 ->
//  pdf.py:27-100 / 171-235, 585-622)
//
// k_fused records every model within the weight threshold of the RUNNING best in per-object lists in
// HBM (16 B out, 16 B back: 3.9 B per pair at SDSS depth against 0.06 B algorithmic) and stacks them
// after the model loop, when every wave of the CU sits in the same latency-bound list walk.  Two facts
// remove the lists:
//   (1) the dimensionality-prior likelihood chi2^(k/2) e^(-chi2/2) / C has its maximum at chi2 = k
//       whatever the data are, so ln L(k) is an upper bound of every ln-like of every object and can
//       serve as THE reference of every weight: w = L(chi2) / L(k) <= 1.  No re-basing, no overflow
//       case; objects whose best weight lies below 2^-400 of the mode's are handed to the exact ln-space
//       sweep (k_fused<.., false>).
//   (2) the reference stacks the models with w > wt_thresh * max(w) (pdf.py:591).  max(w) <= 1, so
//       w > wt_thresh is SUFFICIENT to be stacked and can be decided the moment the pair is seen: its
//       weight goes straight into the object's LDS histogram (one ds_add_f64 at the label index, the
//       single-kernel form of the stack).  Only the pairs with wt_thresh * (running best) < w <=
//       wt_thresh -- a thin band when the best model fits well -- wait in a short per-object list
//       ("ambiguous") for the exact maximum.
// The exact (fp64) weight of a pair costs ~28 instructions (integer powers by multiplication, the half power by a Newton-refined
// v_rsq_f64, one table exponential).  A lane that computes it on the spot makes its whole wave pay, so the pairs that can matter
// (chi2, label index) are first compacted into a per-object LDS buffer and the buffer is drained 64 entries at a time by ALL
// lanes: exact weight, exact evidence share, the best weight seen (ln-max = ln L(mode) + its ln), histogram add or the ambiguous
// list.  The drain's latency overlaps the model loop of the other waves instead of forming a phase of its own.
// What is fp32: ONLY the classifier.  t ~ log2 of the pair's weight (four fp32 instructions, no transcendental: log2 chi2 read off
// the float's bits, good to +-0.045 K / 2; the margins cover it) decides whether the pair can matter at all: a weight below 2^-(55 + ceil log2 M) of the best weight seen SO FAR (hence of the final best) is dropped --
// all M of them together change the fp64 sum of the evidence by less than 2^-55 of it, a quarter of its last bit -- and every
// other pair (41 % on the SDSS-depth benchmark, 7 % of them above wt_thresh) goes through the LDS buffer and gets its weight in
// fp64, 64 at a time.  No fp32 term enters any sum: ln-evidence, ln-max, stacked weights and PDFs are the fp64 numbers of the
// reference's logsumexp / exp / KDE (bruteforce.py:619-629) to fp64 rounding.  (Rounds 2-3 summed the pairs below wt_thresh of
// the running best in fp32 -- 1e-9 on the ln-evidence, 6.3e11 evals/s against 4.9e11: gone, the reference is fp64 throughout.)
// EXACT = true computes every pair's weight in fp64 without classifying (the form for broad likelihoods, where most pairs matter).
//
// Mode B (free scale) runs the EXACT form: its chi2 needs the scale first (two passes over the bands, pdf.py:181-189), the closed
// form A - inter^2 / shape cancels at S/N^2 ~ 1e9 and could only screen -- with 41 % of the pairs then re-evaluated from a gathered
// record that was slower (2.9e11 evals/s) than weighing every pair from the residual form directly (3.4e11).
#pragma once
#include <type_traits>
#include "fz_kernels.h"
#include "fz_nolist.h"

namespace fz {

// exp(x) for |x| <= 700 with a 256-entry table (2 KB: what the LDS has left beside 16 histogram rows, the rings and the
// model tiles): n = round(x 256 / ln 2) read off the low mantissa word as in exp_core, r = x - n ln2 / 256 (|r| <= 1.4e-3),
// exp(r) by the degree-5 Taylor polynomial (remainder 9e-21), 2^(n mod 256 / 256) from the table, 2^(n div 256) into the
// exponent field.  Same error bound as exp_core (tests/test_hip_fastmath.py covers both).
#define FZ_HEXP_K 256
#ifndef FZ_HIST_MP
#define FZ_HIST_MP 1             // 64-model groups per trip of the model loop
#endif
#ifndef FZ_HIST_REFRESH
#define FZ_HIST_REFRESH 32     // steps between two updates of the candidate bar (and flushes of the fp32 partial sums)
#endif
__device__ __forceinline__ double exp_small_tab(double x, const double* __restrict__ tab) {
    const double MAGIC = 6755399441055744.0;                     // 1.5 * 2^52
    x = vmin_raw(vmax_raw(x, -700.0), 700.0);                     // also maps NaN -> -700
    const double d = fma(x, 369.3299304675746, MAGIC);           // 256 / ln 2
    const double r = fma(d - MAGIC, -0.0027076061740622863, x);  // ln 2 / 256
    const int n = __double2loint(d);
    const double t = tab[n & (FZ_HEXP_K - 1)];
    double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double v = t * p;                                      // in [1,2)
    return __hiloint2double(__double2hiint(v) + ((n >> 8) << 20), __double2loint(v));
}

// sqrt(r) for r > 0 from v_rsq_f64 (seed ~2^-24) with two coupled Newton steps on the ROOT itself (s += (r - s^2) y / 2: two
// instructions each, quadratic: 2^-24 -> 2^-47 -> rounding); r == 0 (chi2 == 0, a self match) gives 0
__device__ __forceinline__ double sqrt_nr(double r) {
    const double y = __builtin_amdgcn_rsq(vmax_raw(r, 1e-300));
    const double hy = 0.5 * y;
    double s = r * y;
    s = fma(fma(-s, s, r), hy, s);
    s = fma(fma(-s, s, r), hy, s);
    return s;
}
// w = L(chi2) / L(K) = (chi2 / K)^(K/2) exp(-(chi2 - K) / 2), all fp64: integer powers by multiplication,
// the half power by a Newton-refined v_rsq_f64, one exp, no log
template <int WP, bool SMALL = false>
__device__ __forceinline__ double hist_exactw(double c2, const FastTabs& tb) {
    constexpr double K = (double)WP;
    const double r = c2 * (1.0 / K);
    double pw = 1.0;
    if constexpr ((WP >> 1) <= 3) {
        if constexpr ((WP >> 1) >= 1) pw = r;
        if constexpr ((WP >> 1) >= 2) pw = pw * r;
        if constexpr ((WP >> 1) >= 3) pw = pw * r;
    } else {
        // wide band sets (k up to 30): r^(k/2) by squaring
        double base = r;
#pragma unroll
        for (int e = WP >> 1; e > 0; e >>= 1) { if (e & 1) pw = pw * base; base = base * base; }
    }
    if constexpr (WP & 1) pw = pw * sqrt_nr(r);           // chi2 == 0 (self match): weight 0
    const double w = pw * (SMALL ? exp_small_tab(fma(c2, -0.5, 0.5 * K), tb.expt) : exp_clamped(fma(c2, -0.5, 0.5 * K), tb));
    // the exponential is clamped at e^-700, which a high power of a huge chi2 would lift back into range (r^15 reaches 1e300):
    // beyond the clamp the weight is zero, as in the reference's exp
    if constexpr ((WP >> 1) > 3) return (c2 < K + 1400.0) ? w : 0.0;
    else return w;
}

// the same for a band count known at run time (wide sets padded up to 16 / 32 bands: the pad bands add nothing to chi2, the
// power is that of the REAL band count): r^(k div 2) by squaring over the bits of a wave-uniform k
template <bool SMALL = false>
__device__ __forceinline__ double hist_exactw_rt(double c2, int wp, const FastTabs& tb) {
    const double K = (double)wp;
    if (wp == 0) return SMALL ? exp_small_tab(-0.5 * c2, tb.expt) : exp_clamped(-0.5 * c2, tb);      // wave-uniform branch
    const double r = c2 * rcp_nr<2>(K);
    double pw = 1.0, base = r;
    for (int e = wp >> 1; e > 0; e >>= 1) { if (e & 1) pw = pw * base; base = base * base; }     // scalar loop
    if (wp & 1) pw = pw * sqrt_nr(r);
    const double w = pw * (SMALL ? exp_small_tab(fma(c2, -0.5, 0.5 * K), tb.expt) : exp_clamped(fma(c2, -0.5, 0.5 * K), tb));
    return (c2 < K + 1400.0) ? w : 0.0;
}

// models per LDS tile by record width: two tiles, the histograms and the candidate buffers share 160 KB
template <class SRC>
constexpr int hist_tile() {
    // 16 waves per block (up to 8 bands): 256-model tiles up to 10 doubles per record, 128 beyond (6-8 bands with per-model errors:
    // 14-18 doubles); 8 waves per block (the wide instantiations): 256 up to 18 doubles, 128 up to 34, 64 beyond
    if (SRC::NB <= 8) return (SRC::RW <= 10 && !(SRC::LMODE == 2 && SRC::RW > 6)) ? 256 : 128;   // (free scale at 7 / 8 bands: the model-number buffer is twice the size)
    return SRC::RW <= 18 ? 256 : (SRC::RW <= 34 ? 128 : 64);
}
// wide records: the next step's model record is NOT requested ahead where a second copy of the record does not fit the
// register file beside the object (16 bands with per-model errors or the closed-form screen, 32 bands)
template <class SRC>
constexpr bool hist_prefetch() {
    if (SRC::NB <= 8) return SRC::NB + SRC::NVAL <= 18;              // 16 waves per block, 128 registers: all but 7 / 8 bands with per-model errors
    return 2 * SRC::NB + 2 * SRC::RW <= 96;
}

// wave-wide maximum of a float without the LDS pipe: DPP within rows of 16 lanes, then the four row results through scalar
// registers; the result is wave-uniform (lanes hold the same SGPR-fed value)
__device__ __forceinline__ float wave_maxf_dpp(float v) {
    int x = __float_as_int(v);
    auto mx = [](int a, int b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(__int_as_float(a)), "v"(__int_as_float(b))); return __float_as_int(r); };
    x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));     // row_half_mirror
    x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));     // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(x, 0)), b = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(x, 32)), d = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// wave-wide maximum of a POSITIVE double to 20 mantissa bits: positive doubles order like their high words, so an integer maximum
// of the high words (DPP within rows, the four row results through scalar registers: ~12 instructions, no LDS crossbar) gives
// the maximum rounded DOWN to a multiple of 2^-20 of itself -- a lower bound, which is what its use wants
__device__ __forceinline__ double wave_max_pos_hi(double v) {
    int x = __double2hiint(v);
    auto mx = [](int a, int b) { return a > b ? a : b; };
    x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));     // row_half_mirror
    x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));     // row_mirror
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return __hiloint2double(mx(mx(a, b), mx(c, d)), 0);
}

// chi2 of the band-constant-variance algebra in two instructions per band (k_hist, below); 0 restores the exact-difference form
#ifndef FZ_HIST_CHI2_2OP
#define FZ_HIST_CHI2_2OP 1
#endif
#ifndef FZ_HIST_C2ZERO
#define FZ_HIST_C2ZERO 1e-16
#endif

template <int TW>
struct HistState {
    double S[TW];            // per lane, EXACT: sum of the weights
    double Sc[TW];           // per lane: sum of the settled pairs' weights (EXACT: the best weight seen)
    double wmx[TW];          // per lane: largest exact weight among the settled pairs (ln-max = ln L(mode) + ln of it)
    float tmax[TW];          // per lane: largest log2 weight seen (the classifier's fp32 estimate)
    float tthr[TW];          // wave-uniform: pairs with log2 w at or below this are dropped
    double wamb[TW];         // wave-uniform: a settled weight above this (0.999 wt_thresh x the best settled so far) that is not stacked at once waits in the ambiguous list
    int pend[TW];            // wave-uniform: entries waiting in the object's candidate buffer
    int namb[TW];            // wave-uniform: entries in the ambiguous list
    int tick, next;
};

// OBJK: the band count behind the power of chi2 is the OBJECT's (its observed bands; one object per wave, so it is wave-uniform):
// band sets padded up to 12 / 16 / 24 / 32 bands, and objects with unobserved bands against unmasked models in modes Ai / B, where a
// masked band carries inverse variance 0 (k_prep_objects) and adds exactly nothing to chi2 in the mask-free arithmetic.
template <class SRC, int TW, int NW, bool EXACT, bool OBJK = (SRC::NB > 8)>
__global__ __launch_bounds__(NW * 64) void k_hist(SRC src_, const KdeView* __restrict__ kvp, int acc_stride, int64_t N, int M,
                                                   double wt_thresh, int normalize, Cand* __restrict__ amb, int64_t cap,
                                                   double* __restrict__ lmap, double* __restrict__ levid, double* __restrict__ pdfs,
                                                   const int* __restrict__ omap, int* __restrict__ redo) {
    constexpr int TILE = hist_tile<SRC>(), RW = SRC::RW, TDR = RW * TILE, TD = TDR + TILE / 2, NT = NW * 64, OD = SRC::OBJ_DOUBLES;
    constexpr int WP = SRC::WPOW, BT = SRC::NB;
    static_assert(!OBJK || TW == 1, "per-object band counts: one object per wave");
    constexpr bool KRT = OBJK;                                          // run-time power (set per object below)
    int wpr = WP;
    double K = (double)WP;
    bool kok = true;                                                    // OBJK: the object's power is >= 1 (else: the exact sweep)
    double lgq = src_.lp.lg_full;
    constexpr int NOBJ = NW * TW;
    constexpr int CAP = 128, DTHR = CAP - 64;                             // ring entries per object; drain from DTHR pending entries on
    using tag_t = uint16_t;                                               // label index (< 65536, checked by the launcher)
    static_assert(WP >= 1 && WP <= 30, "chi2^(1/2) ... chi2^15");
    __shared__ __attribute__((aligned(16))) double tileA[TD];
    __shared__ __attribute__((aligned(16))) double tileB[TD];
    __shared__ __attribute__((aligned(16))) double s_c2[EXACT ? 2 : NOBJ * CAP];
    __shared__ tag_t s_tag[EXACT ? 2 : NOBJ * CAP];
    // EXACT has no rings and keeps the full log / exp tables in LDS (every pair takes an exp); the screen form a 256-entry exp table
    __shared__ __attribute__((aligned(16))) double s_tabs[EXACT ? FZ_TABS_DOUBLES : FZ_HEXP_K];
    // the parked object rows are only read once, right after the first barrier of a round (into registers): in the screen form
    // they share the LDS of the wave's candidate buffer, which is empty then (all settled at the end of the previous round)
    __shared__ __attribute__((aligned(16))) double s_objs[EXACT ? NOBJ * OD : 2];
    static_assert(EXACT || TW * OD <= TW * CAP, "an object row must fit its candidate buffer");
    extern __shared__ double s_rows[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * NW, gw = (int64_t)blockIdx.x * NW + wave;
    const int64_t ngroups = (N + TW - 1) / TW, nrounds = (ngroups + nwaves - 1) / nwaves;
    const int ntiles = (M + TILE - 1) / TILE;
    SRC src = src_;
    src.tb = global_tabs();                                       // finish (log of the evidence, the ambiguous band): through the vector L1
    const FastTabs tb = src.tb;
    FastTabs tbx;                                                 // what hist_exactw reads in the model loop
    if constexpr (EXACT) tbx = stage_tabs(s_tabs, tid, NT);
    else {
        for (int k = tid; k < FZ_HEXP_K; k += NT) s_tabs[k] = FZ_EXP_TAB[k * (FZ_EXP_K / FZ_HEXP_K)];
        tbx.logt = nullptr; tbx.expt = s_tabs;
    }
    const KdeView kv = *kvp;
    const int32_t* posw = kv.pos;
    const int w0 = kv.w0;
    double* objs = EXACT ? s_objs + wave * (TW * OD) : s_c2 + wave * (TW * CAP);
    double* rows = s_rows + (size_t)wave * TW * acc_stride;
    double* rc2 = s_c2 + (EXACT ? 0 : wave * (TW * CAP));
    tag_t* rtag = s_tag + (EXACT ? 0 : wave * (TW * CAP));
    Cand* ambw = amb + (size_t)gw * TW * cap;
    // log2 of the screening weight: t = (K/2) log2(chi2) - (chi2 - K) log2(e) / 2 - (K/2) log2(K)
    // The classifier: t ~ log2 of the pair's weight = (K/2) log2(chi2 / K) - (chi2 - K) log2(e) / 2, in four fp32 instructions and no
    // transcendental -- log2(chi2) is read off the float's bits (exponent + mantissa as a fraction: the piecewise-linear log2, at most
    // 0.0861 low; centred, +-0.043), so  t = hk23 * float(bits) - 0.7213 * chi2 + T0c  with an error below 0.045 K / 2 + 1e-4.  The
    // margins below (tmarg) cover it; nothing that reaches an output is computed from t.
    auto t_consts = [](double Kd, float& hk23, float& T0c, float& tmarg) {
        const double hk = 0.5 * Kd;
        hk23 = (float)(hk / 8388608.0);
        T0c = (float)((Kd > 0.0 ? -hk * log2(Kd) : 0.0) + Kd * 0.7213475204444817 + hk * (-127.0 + 0.043));
        tmarg = (float)(0.5 + 0.045 * hk);
    };
    float hk23, T0c, tmarg;
    t_consts(K, hk23, T0c, tmarg);
    float tzero = K > 0.0 ? -INFINITY : T0c;                      // t of chi2 == 0 (power 0: weight 1)
    const float lthr2 = (wt_thresh > 0.0) ? (float)log2(wt_thresh) : -INFINITY;
    // the drop bar: log2 of the share of the best weight below which a pair cannot matter to an fp64 sum over M of them
    // (2^-55 / M, and one more bit for the classifier's own error: |t - log2 w| < 1e-4 for every pair that could sit at the bar of an
    // object that stays here, see `ok`); never above the stacking threshold
    int mbits = 0;
    while (((int64_t)1 << mbits) < (int64_t)M) ++mbits;
    const float ldrop0 = fminf(lthr2, -(float)(55 + mbits));      // (minus the classifier's margin, per object: tmarg)
    const double thr_def = wt_thresh * (1.0 + 1e-3);              // above this a weight is stacked whatever the maximum turns out to be
    auto lnl_c2 = [&](double c2) {
        if constexpr (KRT) return (wpr == 0) ? fma(-0.5, c2, -lgq) : chi2_logpdf<true>(0.5 * K, c2, lgq, tb);      // (power 0: no x log x term)
        else return src.lnl_of_chi2(c2);
    };
    auto exactw_tab = [&](double c2, const FastTabs& t, auto small) {
        if constexpr (KRT) return hist_exactw_rt<decltype(small)::value>(c2, wpr, t);
        else return hist_exactw<WP, decltype(small)::value>(c2, t);
    };
    double lref = uniform_d(lnl_c2(K));                           // ln L at the mode: the reference of every weight

    for (int64_t rnd = 0; rnd < nrounds; ++rnd) {
        const int64_t g = gw + rnd * nwaves;
        const bool work = g < ngroups;                            // wave-uniform
        const int64_t i0 = work ? g * TW : 0;
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            const int64_t os = i0 + o < N ? i0 + o : N - 1;
            src.park_obj(omap ? (int64_t)omap[os] : os, objs + o * OD, lane);
        }
        for (int k = lane; k < TW * acc_stride; k += 64) rows[k] = 0.0;
        if constexpr (OBJK) {
            const int64_t os = i0 < N ? i0 : N - 1;
            const int nb = __builtin_amdgcn_readfirstlane(__popc(src.ov.bits[omap ? (int64_t)omap[os] : os]));   // observed bands (pad bits are 0)
            const int64_t oi = omap ? (int64_t)omap[os] : os;
            wpr = nb - (SRC::LMODE == 2 ? 3 : 2);
            // power 0 (two observed bands; three with the free scale): L = e^{-chi2/2} / C, largest at chi2 = 0 -- still bounded by its
            // value at the mode, so the same scheme holds with K = 0.  Below that the likelihood is unbounded at chi2 -> 0: the sweep.
            kok = wpr >= 0;
            if (!kok) wpr = 1;                                          // (arithmetic stays finite; the object goes to the sweep)
            // without the dimensionality prior (pdf.py:94-98, 230-235) the likelihood IS the power-0 form for every band count:
            // ln L = -chi2 / 2 - (N_dim ln 2 pi + sum ln var) / 2, the second term a constant of the object in modes Ai / B
            const bool dp = src.lp.dim_prior != 0;
            if (!dp) { wpr = 0; kok = true; }
            K = uniform_d((double)wpr);                                 // (wave-uniform: scalar registers)
            t_consts(K, hk23, T0c, tmarg);
            hk23 = uniform_f(hk23); T0c = uniform_f(T0c); tmarg = uniform_f(tmarg);
            tzero = wpr > 0 ? -INFINITY : T0c;
            lgq = uniform_d(dp ? src.lp.lgtab[nb] : 0.5 * ((double)nb * FZ_LN2PI + src.ov.slv[oi]));
            lref = uniform_d(lnl_c2(K));
        }
        nl_stage_tile<SRC, TILE, NT, true>(src, posw, 0, tileA, tid, wave);
        __syncthreads();
        // the wave's objects stay in VGPRs for the whole model loop (an LDS broadcast read: the compiler
        // cannot tell that they are wave-uniform and keeps them out of the scalar file)
        typename SRC::OR ob[TW];
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            src.load_obj_lds(objs + o * OD, ob[o]);
            if constexpr (BT > 16) {
                // 32-band records: the object's fluxes live in SCALAR registers (they are wave-uniform; every use has room for one
                // scalar operand), which leaves the vector file to the inverse variances and the model record
#pragma unroll
                for (int b = 0; b < BT; ++b) ob[o].x[b] = uniform_d(ob[o].x[b]);
            }
        }
        // chi2 in TWO instructions per band where the variance does not depend on the model (the mode-Ai algebra, every band count): with
        // s = sqrt(1 / var) and xs = x s, formed here once per object,  (x - y)^2 / var = d^2,  d = fma(-y, s, xs)  -- one rounding of
        // x s instead of the exact difference: |d - (x - y) s| <= 2^-53 |x| s = 1.1e-16 S/N.  The one place where the exact difference
        // matters is a model IDENTICAL to the object (a training-set self match): the reference gets chi2 = 0 -- weight 0 for every
        // power > 0 -- and a chi2 of 1e-28 in its place would make that pair the best fit of an object nothing else fits.  So this form
        // treats chi2 <= 1e-16 AS zero (C2ZERO below: the classifier's zero test, at no cost; one compare + select in the direct form):
        // rounding cannot lift a self match above that for S/N < 4e7, and a genuine pair below it weighs < (1e-16 / k)^(k/2) of the
        // mode (power 0: exp(-chi2 / 2) = 1 to the last bit either way).
        constexpr bool C2OP = (SRC::LMODE == 1) && FZ_HIST_CHI2_2OP;
        // The free scale with model errors ignored (LMODE == 2) in the same units: ys = y s per pair and band, then
        // inter = sum xs ys, shape = sum ys ys, scale = inter / shape, d = fma(-scale, ys, xs), chi2 = fma(d, d, chi2) -- five
        // instructions per band instead of six.  A model identical to the object still has xs == ys bit for bit, hence inter ==
        // shape, scale == 1 and chi2 == 0 EXACTLY (no zero threshold here); the residual x - scale y cancels to rounding in either
        // form, so the accuracy is that of the six-instruction form.
        constexpr bool C2OPB = (SRC::LMODE == 2) && !SRC::SAFE && FZ_HIST_CHI2_2OP;
        if constexpr (C2OP || C2OPB) {
#pragma unroll
            for (int o = 0; o < TW; ++o)
#pragma unroll
                for (int b = 0; b < BT; ++b) {
                    ob[o].v[b] = sqrt(ob[o].v[b]);
                    ob[o].x[b] = ob[o].x[b] * ob[o].v[b];
                    if constexpr (BT > 16) ob[o].x[b] = uniform_d(ob[o].x[b]);        // (32-band records: back to the scalar file, see above)
                }
        }
        HistState<TW> hs;
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            hs.S[o] = 0.0; hs.Sc[o] = 0.0; hs.wmx[o] = 0.0;
            hs.tmax[o] = -INFINITY;
            hs.tthr[o] = -INFINITY;                          // (nothing is dropped before the first look at the best weight, after step 1)
            hs.wamb[o] = 0.0; hs.pend[o] = 0; hs.namb[o] = 0;
        }
        hs.tick = 0; hs.next = 1;

        // one candidate (exact chi2, label index, all lanes of `act`): evidence share, best chi2 on either
        // side of the mode, histogram add or ambiguous list
        auto settle = [&](int o, bool act, double c2, int tag) {
            const double w = act ? exactw_tab(c2, tbx, std::integral_constant<bool, !EXACT>{}) : 0.0;
            hs.Sc[o] += w;
            hs.wmx[o] = vmax_raw(hs.wmx[o], w);
            if (w > thr_def) unsafeAtomicAdd(&rows[o * acc_stride + tag + w0], w);
            // (every pair that can matter to the evidence comes through here, not only the few near the threshold: the ambiguous list
            //  takes those within wt_thresh of the best weight seen so far -- a superset of what the exact maximum will admit)
            const bool am = act && !(w > thr_def) && w > hs.wamb[o];
            const unsigned long long mask = __ballot(am);
            if (mask) {                                           // wave-uniform
                const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                const int np = __builtin_popcountll(mask);
                // the list holds `cap` entries (the launcher sizes it well above what a well-fitted object needs, not at M): an
                // object that would overflow it -- every model within the band of a poor best fit -- is handed to the exact sweep
                if (hs.namb[o] >= 0 && hs.namb[o] + np <= cap) {
                    if (am) { Cand e; e.lnl = c2; e.j = tag; e.pad = 0; ambw[(size_t)o * cap + hs.namb[o] + pre] = e; }
                    hs.namb[o] += np;
                } else hs.namb[o] = -1;                           // overflowed: nothing more is stored, the object goes to the sweep
            }
        };
        // settle the LAST (up to) 64 entries of object o's buffer with all lanes (their order does not matter: nothing has to move)
        auto drain = [&](int o) {
            const int n = hs.pend[o] < 64 ? hs.pend[o] : 64;
            const int rest = hs.pend[o] - n;                      // < 64
            double c2 = rc2[o * CAP + rest + lane];
            int tag = (int)rtag[o * CAP + rest + lane];
            const bool act = lane < n;
            settle(o, act, c2, tag);
            hs.pend[o] = rest;
        };

        auto run_tile = [&](const double* cur, double* nxt, int t, auto tailc) {
            constexpr bool TAIL = decltype(tailc)::value;
            if (t + 1 < ntiles) nl_stage_tile<SRC, TILE, NT, true>(src, posw, t + 1, nxt, tid, wave);
            if (work) {
                const int32_t* tags = reinterpret_cast<const int32_t*>(cur + TDR);
                // MP 64-model groups per trip: their chi2 chains and weights sit in one basic block (the appends, which
                // branch, follow), and the next trip's records are requested before the current ones are used
                constexpr int MP = EXACT ? 1 : FZ_HIST_MP;
                static_assert((TILE / 64) % MP == 0, "groups per trip must divide the tile");
                constexpr bool PF = hist_prefetch<SRC>();
                typename SRC::MR mn[PF ? MP : 1];
                int tagn[MP];
                if constexpr (PF) {
#pragma unroll
                    for (int q = 0; q < MP; ++q) {
                        src.template load_model_lds<TILE>(cur, q * 64 + lane, mn[q]);
                        tagn[q] = tags[q * 64 + lane];
                    }
                }
#pragma unroll
                for (int st = 0; st < TILE / 64; st += MP) {
                    typename SRC::MR m[MP];
                    int ptag[MP];
#pragma unroll
                    for (int q = 0; q < MP; ++q) {
                        if constexpr (PF) { m[q] = mn[q]; ptag[q] = tagn[q]; }
                        else { src.template load_model_lds<TILE>(cur, (st + q) * 64 + lane, m[q]); ptag[q] = tags[(st + q) * 64 + lane]; }
                        asm volatile("" : "+v"(ptag[q]));         // keeps the index read up here, beside the record's (sunk into the append, it made every step wait for the LDS there)
                    }
                    if (PF && st + MP < TILE / 64) {
#pragma unroll
                        for (int q = 0; q < MP; ++q) {
                            src.template load_model_lds<TILE>(cur, (st + MP + q) * 64 + lane, mn[q]);
                            tagn[q] = tags[(st + MP + q) * 64 + lane];
                        }
                    }
                    double c2[MP][TW];
#pragma unroll
                    for (int q = 0; q < MP; ++q)
#pragma unroll
                        for (int o = 0; o < TW; ++o) {
                            if constexpr (C2OP) {
                                double c = 0.0;
#pragma unroll
                                for (int b = 0; b < BT; ++b) { const double d = fma(-m[q].y[b], ob[o].v[b], ob[o].x[b]); c = fma(d, d, c); }
                                if constexpr (EXACT) c = (c <= FZ_HIST_C2ZERO) ? 0.0 : c;      // (the screen form makes this decision in its classifier)
                                c2[q][o] = c;
                            } else if constexpr (C2OPB) {
                                double ys[BT], inter = 0.0, shape = 0.0;
#pragma unroll
                                for (int b = 0; b < BT; ++b) {
                                    ys[b] = m[q].y[b] * ob[o].v[b];
                                    inter = fma(ys[b], ob[o].x[b], inter);
                                    shape = fma(ys[b], ys[b], shape);
                                }
                                double sc;
                                if (!(shape > 1e-280 && shape < 1e280)) sc = inter / shape;       // shape == 0 (no usable band) -> nan / inf like NumPy
                                else {
                                    const double rc = rcp_nr<1>(shape);
                                    sc = inter * rc;
                                    sc = fma(fma(-sc, shape, inter), rc, sc);                     // residual correction: n / n == 1 exactly
                                }
                                double c = 0.0;
#pragma unroll
                                for (int b = 0; b < BT; ++b) { const double d = fma(-sc, ys[b], ob[o].x[b]); c = fma(d, d, c); }
                                c2[q][o] = c;
                            } else
                                c2[q][o] = src.chi2_of(ob[o], m[q]);
                            if (TAIL) c2[q][o] = (t * TILE + (st + q) * 64 + lane < M) ? c2[q][o] : 1e30;   // pad lanes: weight 0
                        }
                    if constexpr (EXACT) {
                        const int j = t * TILE + st * 64 + lane;
#pragma unroll
                        for (int o = 0; o < TW; ++o) {
                            // every pair in fp64; the running best (of the exact weights) bounds what can still be stacked
                            const bool valid = !TAIL || j < M;
                            const double w = valid ? exactw_tab(c2[0][o], tbx, std::false_type{}) : 0.0;
                            hs.S[o] += w;
                            hs.Sc[o] = vmax_raw(hs.Sc[o], w);       // (EXACT: Sc holds the best weight seen: the bar of the ambiguous band, and ln-max at the end)
                            if (w > thr_def) unsafeAtomicAdd(&rows[o * acc_stride + ptag[0] + w0], w);
                            const bool am = valid && !(w > thr_def) && (w >= wt_thresh * 0.999 * hs.Sc[o]);
                            const unsigned long long mask = __ballot(am);
                            if (mask) {
                                const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                                const int np = __builtin_popcountll(mask);
                                if (hs.namb[o] >= 0 && hs.namb[o] + np <= cap) {
                                    if (am) { Cand e; e.lnl = c2[0][o]; e.j = ptag[0]; e.pad = 0; ambw[(size_t)o * cap + hs.namb[o] + pre] = e; }
                                    hs.namb[o] += np;
                                } else hs.namb[o] = -1;
                            }
                        }
                        if ((++hs.tick & 15) == 0) {
#pragma unroll
                            for (int o = 0; o < TW; ++o) hs.Sc[o] = wave_max(hs.Sc[o]);
                        }
                    } else {
                        float tl[MP][TW];
#pragma unroll
                        for (int q = 0; q < MP; ++q)
#pragma unroll
                            for (int o = 0; o < TW; ++o) {
                                const float cf = (float)c2[q][o];                         // chi2 >= 0 (inf: t = -inf; nan: nan, settled and found weightless)
                                tl[q][o] = fmaf((float)__float_as_int(cf), hk23, fmaf(cf, -0.72134752f, T0c));
                                // chi2 == 0 (a training-set self match) has weight 0 for every power K > 0: its t must be -inf, not the -127 K / 2 the bit
                                // trick gives -- a finite t would pass for the object's best weight when everything else lies far below it
                                // (two-instruction chi2: everything up to 1e-16 IS the zero, see C2OP; a nan stays a nan)
                                tl[q][o] = (C2OP ? !(cf <= (float)FZ_HIST_C2ZERO) : (cf != 0.f)) ? tl[q][o] : tzero;
                                if (TAIL) tl[q][o] = (t * TILE + (st + q) * 64 + lane < M) ? tl[q][o] : -INFINITY;   // pad lanes: dropped whatever the bar
                            }
#pragma unroll
                        for (int q = 0; q < MP; ++q) {
#pragma unroll
                            for (int o = 0; o < TW; ++o) {
                                // ONE compare: at or below the bar -> dropped (it cannot matter to an fp64 sum);
                                // above it (or not a number: chi2 beyond fp32's range, settled like any other and found weightless) -> the buffer
                                const bool le = tl[q][o] <= hs.tthr[o];
                                const bool c = !le;
                                asm("v_max_f32 %0, %1, %2" : "=v"(hs.tmax[o]) : "v"(hs.tmax[o]), "v"(tl[q][o]));   // (a nan operand yields the other one)
                                // -> the object's buffer (ballot + mbcnt compaction); its fill level is a wave-uniform scalar
                                const unsigned long long mask = __ballot(c);
                                // (v_mbcnt adds its count to a base: the fill level)  < 64 wait when a group begins, a group adds at most 64: never past CAP
                                const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, hs.pend[o]));
#if !defined(FZ_DIAG_NOAPPEND)
                                if (c) { rc2[o * CAP + slot] = c2[q][o]; rtag[o * CAP + slot] = (tag_t)ptag[q]; }
#endif
                                hs.pend[o] += __builtin_popcountll(mask);
                            }
                            if (++hs.tick == hs.next) {                               // steps 1, 2, 4, 8, 16, then every FZ_HIST_REFRESH-th
                                hs.next = hs.tick < 16 ? 2 * hs.tick : hs.tick + FZ_HIST_REFRESH;
#pragma unroll
                                for (int o = 0; o < TW; ++o) {
                                    const float mx = wave_maxf_dpp(hs.tmax[o]);       // the bars follow the wave-wide best weight seen
                                    hs.tmax[o] = mx;
                                    hs.tthr[o] = mx + (ldrop0 - 2.f * tmarg);       // (the best's t and the pair's t each carry the margin)
                                    // the ambiguous band starts at wt_thresh x the best EXACT weight settled so far (a lower bound of the final best:
                                    // what still waits in the buffer only raises it)
                                    hs.wamb[o] = wave_max_pos_hi(hs.wmx[o]) * (wt_thresh * 0.999);
                                }
                            }
#pragma unroll
                            for (int o = 0; o < TW; ++o) {
#if defined(FZ_DIAG_NODRAIN)
                                if (hs.pend[o] >= DTHR) { hs.pend[o] -= 64; }
#else
                                while (hs.pend[o] >= DTHR) drain(o);
#endif
                            }
                        }
                    }
                }
            }
#if !defined(FZ_DIAG_NOBARRIER)
            __syncthreads();
#endif
        };
        for (int t = 0; t < ntiles; t += 2) {
            if (t + 1 < ntiles) run_tile(tileA, tileB, t, std::false_type{}); else run_tile(tileA, tileB, t, std::true_type{});
            if (t + 1 < ntiles) {
                if (t + 2 < ntiles) run_tile(tileB, tileA, t + 1, std::false_type{}); else run_tile(tileB, tileA, t + 1, std::true_type{});
            }
        }

        // finish: what is left in the rings, the exact maximum, the evidence, the ambiguous entries, the PDF
        if (work) {
#pragma unroll 1
            for (int o = 0; o < TW; ++o) {
                if (i0 + o >= N) break;
                const int64_t i = omap ? (int64_t)omap[i0 + o] : i0 + o;
                double* row = rows + o * acc_stride;
                double wbest_run = 0.0;
                if constexpr (!EXACT) {
                    while (hs.pend[o] > 0) drain(o);
                } else {
                    wbest_run = wave_max(hs.Sc[o]);
                }
                // ln L of the best model = ln L(mode) + ln of its exact relative weight (4e-16 relative on the weight: 1e-16 on ln-max)
                if constexpr (!EXACT) wbest_run = wave_max(hs.wmx[o]);
                const double lbest = (wbest_run > 0.0) ? uniform_d(lref + log_pos(wbest_run, tb)) : -INFINITY;
                const double stot = wave_sum(EXACT ? hs.S[o] : hs.Sc[o]);
                const double le = lref + log_pos(stot, tb);
                const float tm = EXACT ? 0.f : wave_maxf(hs.tmax[o]);
                // no candidate at all, an evidence that is not a number, or a best weight so far below the mode that weights relative to
                // the MODE leave the comfortable range (2^-400: every pair within the drop bar of it is still a normal number well above
                // the exponential's clamp, and chi2 stays small enough -- < ~600 -- for the classifier's error bound): the exact ln-space sweep decides
                const bool ok = (le - le == 0.0) && (lbest > -INFINITY) && (EXACT ? (wbest_run > 1e-24) : (tm >= -400.f)) && hs.namb[o] >= 0 && kok;
                // the ambiguous band, by the reference's own rule (pdf.py:591) with the exact maximum and evidence
                const int na = __builtin_amdgcn_readfirstlane(hs.namb[o]);
                if (na > 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // entries were written by other lanes of this wave
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    const double thr = wt_thresh * exp_neg(lbest - le, tb);     // wt_thresh * max(wt)
                    const Cand* cb = ambw + (size_t)o * cap;
                    for (int c0 = 0; c0 < na; c0 += 64) {
                        const int k = c0 + lane;
                        const bool in1 = k < na;
                        const Cand e1 = cb[in1 ? k : 0];
                        const double l1 = in1 ? lnl_c2(e1.lnl) : -INFINITY;
                        const bool s1 = in1 && (exp_neg(l1 - le, tb) > thr);   // strict
                        if (s1) unsafeAtomicAdd(&row[e1.j + w0], exactw_tab(e1.lnl, tb, std::false_type{}));
                    }
                }
                if (lane == 0) {
                    if (lmap) lmap[i] = lbest;
                    if (levid) levid[i] = le;
                    if (!ok) redo[1 + atomicAdd(redo, 1)] = (int)i;
                }
                kde_finalize<true>(kv, row, ok, normalize, pdfs + i * kv.G, lane, ok ? 1.0 / stot : 1.0, true);
            }
        }
        __syncthreads();
    }
}

}  // namespace fz
