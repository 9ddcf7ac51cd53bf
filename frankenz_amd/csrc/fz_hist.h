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
#ifndef FZ_HIST_TILE384
#define FZ_HIST_TILE384 1        // narrow records (up to 6 doubles): 384-model tiles -- every tile costs ~12 us of prologue per 2.6e10 pairs
#endif                           // whatever its length (128 / 256 / 384 models: 54.8 / 49.9 / 48.8 ms), and 384 is what fits beside 16 rows
#ifndef FZ_HIST_REFRESH_TILE
#define FZ_HIST_REFRESH_TILE 1   // the candidate bar is refreshed at tile ends (tiles 1, 2, 3, then every 5th) instead of by a test in every step
#endif
#ifndef FZ_HIST_STAGE_LATE
#define FZ_HIST_STAGE_LATE 1     // the copy of the next tile is issued AFTER the first group's record reads of the current one (+0.6 %)
#endif
#ifndef FZ_HIST_SETTLE2
#define FZ_HIST_SETTLE2 0        // 1: the screen form settles TWO entries per lane and drain (128 at a time: two independent chains of the
                                 // exact weight in one basic block), on 192-entry buffers and 128-model tiles (narrow records only)
#endif
#ifndef FZ_HIST_SEG_NW0
#define FZ_HIST_SEG_NW0 16       // segmented form with per-model errors: waves per block
#endif
#ifndef FZ_HIST_SEG_PFMAX
#define FZ_HIST_SEG_PFMAX 12     // segmented form: band count + record values up to which the next record is requested ahead
#endif
#ifndef FZ_HIST_REFRESH
#define FZ_HIST_REFRESH 32     // steps between two updates of the candidate bar (and flushes of the fp32 partial sums)
#endif
// (NEGONLY: the caller's argument cannot exceed a few units -- the settle's -chi2 / 2 + const -- so only the lower clamp is applied)
template <bool NEGONLY = false>
__device__ __forceinline__ double exp_small_tab(double x, const double* __restrict__ tab) {
    const double MAGIC = 6755399441055744.0;                     // 1.5 * 2^52
    x = NEGONLY ? vmax_raw(x, -700.0) : vmin_raw(vmax_raw(x, -700.0), 700.0);      // also maps NaN -> -700
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
// ln of the small integers the band counts give (constant expressions: the compiler has no constexpr log)
constexpr double hist_ln(double k) {
    return k == 1.0 ? 0.0 : k == 2.0 ? 0.6931471805599453 : k == 3.0 ? 1.0986122886681098 : k == 4.0 ? 1.3862943611198906
         : k == 5.0 ? 1.6094379124341003 : k == 6.0 ? 1.791759469228055 : k == 7.0 ? 1.9459101090932196 : 2.0794415416798357;
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
    if constexpr (SMALL && (WP >> 1) <= 3) {
        // the settle of the screen form (chi2 > 0 always: a zero never reaches the buffer): the constant K^(-K/2) e^(K/2) rides in the
        // exponential's argument, so the powers are those of chi2 itself -- c2^(K div 2), times sqrt(c2) without the zero guard --
        // one multiplication, one maximum and one minimum fewer per settled pair than the general form below
        double q = 1.0;
        if constexpr ((WP >> 1) >= 1) q = c2;
        if constexpr ((WP >> 1) >= 2) q = q * c2;
        if constexpr ((WP >> 1) >= 3) q = q * c2;
        if constexpr (WP & 1) {
            const double y = __builtin_amdgcn_rsq(c2), hy = 0.5 * y;
            double sq = c2 * y;
            sq = fma(fma(-sq, sq, c2), hy, sq);
            sq = fma(fma(-sq, sq, c2), hy, sq);
            q = (WP >> 1) ? q * sq : sq;
        }
        constexpr double EK = 0.5 * K - 0.5 * K * hist_ln(K);      // ln of K^(-K/2) e^(K/2)
        return q * exp_small_tab<true>(fma(c2, -0.5, EK), tb.expt);
    }
    if constexpr (!SMALL && (WP >> 1) <= 3) {
        // the direct form: the same folding (so that the two forms build a pair's weight from the same factors: an object's ln-max then
        // does not depend on which form its launch took beyond the exponential's own rounding); the root keeps its zero guard here --
        // this form weighs EVERY pair, a chi2 of exactly 0 (a self match) included, and must give exactly 0 for it
        double q = 1.0;
        if constexpr ((WP >> 1) >= 1) q = c2;
        if constexpr ((WP >> 1) >= 2) q = q * c2;
        if constexpr ((WP >> 1) >= 3) q = q * c2;
        if constexpr (WP & 1) q = (WP >> 1) ? q * sqrt_nr(c2) : sqrt_nr(c2);
        constexpr double EK = 0.5 * K - 0.5 * K * hist_ln(K);
        return q * exp_neg(fma(c2, -0.5, EK), tb);
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

// ... and for the exact band counts (4-8 bands: powers 0 ... 3 and a half) with the reciprocal of k formed once per pattern: no loop,
// no branch but the wave-uniform one around the square root; r^3.5 e^-700 cannot overflow, so the clamp's guard is not needed
template <bool SMALL = false>
__device__ __forceinline__ double hist_exactw_rt8(double c2, int wp, double K, double rK, const FastTabs& tb) {
    if (wp == 0) return SMALL ? exp_small_tab<true>(-0.5 * c2, tb.expt) : exp_clamped(-0.5 * c2, tb);      // wave-uniform branch
    const double r = c2 * rK, r2 = r * r;
    const int h = wp >> 1;
    double pw = ((h & 1) ? r : 1.0) * ((h & 2) ? r2 : 1.0);
    if (wp & 1) pw = pw * sqrt_nr(r);
    return pw * (SMALL ? exp_small_tab<true>(fma(c2, -0.5, 0.5 * K), tb.expt) : exp_clamped(fma(c2, -0.5, 0.5 * K), tb));
}

// models per LDS tile by record width: two tiles, the histograms and the candidate buffers share 160 KB
template <class SRC>
constexpr int hist_tile() {
    // 16 waves per block (up to 8 bands): 256-model tiles up to 10 doubles per record, 128 beyond (6-8 bands with per-model errors:
    // 14-18 doubles); 8 waves per block (the wide instantiations): 256 up to 18 doubles, 128 up to 34, 64 beyond
    if (FZ_HIST_SETTLE2 && SRC::NB <= 8 && SRC::RW <= 6 && SRC::LMODE == 1) return 128;      // (room for the 192-entry buffers; the tile barrier costs nothing)
    if (FZ_HIST_TILE384 && SRC::NB <= 8 && SRC::RW <= 6) return 384;                          // narrow records: the longest tile the LDS holds beside 16 rows
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
//
// SEG: the SEGMENTED model layout -- masked MODELS, and objects with unobserved bands against per-model errors (pdf.py:76-87 with
// models_mask / per-model models_err: the shape of a real training catalogue).  N_dim = sum_b m_obj m_model is then a property of
// the PAIR.  The kernel's copy of the model records is sorted by mask pattern (fz_build_segments: a handful of patterns at a few
// per cent of missing bands), every pattern padded to whole 64-model groups, so that the pattern -- hence N_dim, the power of chi2,
// the normalisation and which bands count -- is WAVE-UNIFORM per step and changes a few dozen times per pass.  At a change the
// wave settles what waits in its buffer and re-derives its "effective object row" for the new pattern: a band that either side
// masks gets inverse variance 0 (modes Ai / B: it then adds exactly nothing to chi2, inter and shape) or, with per-model
// errors, the multiplier 0 in d = fma(-y, mk, x mk) (mk = 1: the exact difference x - y, as before; mk = 0: d = 0).  The reference
// of every weight is the largest mode value over the patterns, LREF = max_s ln L_s(k_s); a pattern's weights carry the factor
// f_s = L_s(k_s) / e^LREF <= 1 (in the classifier: an offset of t), so every bound of the scheme (w <= 1, "w > wt_thresh is
// sufficient") holds across patterns.  An ambiguous entry records its N_dim (Cand::pad) and is settled by the reference's rule
// with its own ln-like.  Objects for which some pattern leaves too few bands for a bounded likelihood go to the exact sweep.
template <class SRC, int TW, int NW, bool EXACT, bool OBJK = (SRC::NB > 8), bool SEG = false>
__global__ __launch_bounds__(NW * 64) void k_hist(SRC src_, const KdeView* __restrict__ kvp, int acc_stride, int64_t N, int M,
                                                   double wt_thresh, int normalize, Cand* __restrict__ amb, int64_t cap,
                                                   double* __restrict__ lmap, double* __restrict__ levid, double* __restrict__ pdfs,
                                                   const int* __restrict__ omap, int* __restrict__ redo) {
    constexpr int TILE = hist_tile<SRC>(), RW = SRC::RW, TDR = RW * TILE, TD = TDR + TILE / 2, NT = NW * 64, OD = SRC::OBJ_DOUBLES;
    constexpr int WP = SRC::WPOW, BT = SRC::NB;
    static_assert(!OBJK || TW == 1, "per-object band counts: one object per wave");
    static_assert(!SEG || (OBJK && SRC::NB <= 8), "segments: run-time band counts, 4-8 bands");
    constexpr double SEG_VBIG = BT <= 5 ? 0x1p192 : (BT == 6 ? 0x1p160 : (BT == 7 ? 0x1p137 : 0x1p120));      // 2^(960 / BT): BT such terms multiply without overflow
    constexpr int KOFF = SRC::LMODE == 2 ? 3 : 2;                       // power of chi2 = N_dim - KOFF (pdf.py:91-93 / 227-229)
    constexpr bool KRT = OBJK;                                          // run-time power (set per object below)
    int wpr = WP;
    double K = (double)WP;
    bool kok = true;                                                    // OBJK: the object's power is >= 1 (else: the exact sweep)
    double lgq = src_.lp.lg_full;
    constexpr int NOBJ = NW * TW;
    constexpr bool S2 = (FZ_HIST_SETTLE2 == 1) && !EXACT && SRC::NB <= 8 && SRC::RW <= 6 && SRC::LMODE == 1;      // two entries per lane and drain
    constexpr int CAP = S2 ? 192 : 128, DTHR = CAP - 64;                  // ring entries per object; drain from DTHR pending entries on
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
    // SEG: the object's row as prepared (the effective row of a mask pattern is derived from it at every change of pattern)
    __shared__ __attribute__((aligned(16))) double s_obj0[SEG ? NOBJ * OD : 2];
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
    const int32_t* posw = SEG ? kv.mc_tag : kv.pos;
    const int w0 = SEG ? 0 : kv.w0;                                      // (a segment tag carries its histogram offset)
    double* objs = EXACT ? s_objs + wave * (TW * OD) : s_c2 + wave * (TW * CAP);
    double* obj0 = s_obj0 + (SEG ? wave * (TW * OD) : 0);
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
    double rK = 1.0;                                                    // SEG: 1 / K of the pattern in force
    auto exactw_tab = [&](double c2, const FastTabs& t, auto small) {
        if constexpr (SEG) {
            // (the common pattern -- every band observed on both sides -- has the compile-time power: the cheaper form, behind a wave-uniform
            //  branch.  The free scale only, where EVERY pair is weighed: 72 -> 67 ms per 2.6e10 pairs; the screen forms lose 0-3 ms to it.)
            if constexpr (SRC::LMODE == 2) { if (wpr == WP) return hist_exactw<WP, decltype(small)::value>(c2, t); }
            return hist_exactw_rt8<decltype(small)::value>(c2, wpr, K, rK, t);
        }
        else if constexpr (KRT) return hist_exactw_rt<decltype(small)::value>(c2, wpr, t);
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
        // SEG: state of the pattern in force (set by seg_switch below)
        uint32_t obits = 0; double oslv = 0.0, fk = 1.0; int segcur = -1, ndcur = 0;
        bool segbad = false;                                            // the pattern in force has no mode (power -1/2), see below
        int nbadm = 0;                                                  // models in such patterns
        // ln-like of a pair from its chi2 and its own N_dim (per lane): pdf.py:90-98 / 226-235
        auto seg_lnl = [&](double c2, int nd) {
            const bool dp = src.lp.dim_prior != 0;
            const int wp = dp ? nd - KOFF : 0;
            const double lg = dp ? src.lp.lgtab[nd] : 0.5 * ((double)nd * FZ_LN2PI + oslv);
            const double xl = (wp == 0) ? 0.0 : (0.5 * (double)wp) * log_pos_t<true>(c2, tb);
            return fma(-0.5, c2, xl) - lg;
        };
        if constexpr (SEG) {
            const int64_t os = i0 < N ? i0 : N - 1;
            const int64_t oi = omap ? (int64_t)omap[os] : os;
            src.park_obj(oi, obj0, lane);
            if constexpr ((SRC::LMODE == 1 || SRC::LMODE == 2) && FZ_HIST_CHI2_2OP && !(SRC::LMODE == 2 && SRC::SAFE)) {
                // the two-instruction chi2's units (s = sqrt(1 / var), xs = x s; below) once per object, not at every change of pattern
                // (same lane wrote the entries: no fence needed)
                if (lane < BT) { const double sv = sqrt(obj0[BT + lane]); obj0[BT + lane] = sv; obj0[lane] = obj0[lane] * sv; }
            }
            obits = (uint32_t)__builtin_amdgcn_readfirstlane((int)src.ov.bits[oi]);
            oslv = uniform_d(src.ov.slv[oi]);
            const bool dp = src.lp.dim_prior != 0;
            // the reference of every weight: the largest value any pattern's likelihood takes at its mode (chi2 = k; power 0: chi2 = 0)
            // Power -1/2 (one common band; two with the free scale): chi2^(-1/2) e^(-chi2/2) has no mode, its pairs are bounded by
            // their own best only -- found by a look-ahead over those (few) models before the loop starts (below).  Less than
            // that: the reference's own row is undefined (gammaln(0), pdf.py:92 / 228) -- the sweep reproduces whatever it gives.
            double lr = -INFINITY; bool bad = false;
            for (int sg = lane; sg < kv.seg_n; sg += 64) {
                const int nd = __popc(obits & kv.seg_mask[sg]);
                const int wp = dp ? nd - KOFF : 0;
                bad |= wp < -1;
                nbadm += (wp == -1) ? kv.seg_start[sg + 1] - kv.seg_start[sg] : 0;
                const double lg = dp ? src.lp.lgtab[nd] : 0.5 * ((double)nd * FZ_LN2PI + oslv);
                const double hw = 0.5 * (double)wp;
                if (wp >= 0) lr = fmax(lr, wp > 0 ? fma(hw, log_pos((double)wp, tb), -hw) - lg : -lg);
            }
            nbadm = __builtin_amdgcn_readfirstlane((int)wave_sum((double)nbadm));
            kok = __ballot(bad) == 0ull && nbadm <= (M >> 3);          // (an object whose every pattern is of that kind: the sweep)
            lref = uniform_d(wave_max(lr));
            float h_, t_; t_consts((double)BT, h_, t_, tmarg);       // (the classifier's margin of the largest power covers every pattern)
            tmarg = uniform_f(tmarg);
        }
        if constexpr (OBJK && !SEG) {
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
        if constexpr ((C2OP || C2OPB) && !SEG) {
#pragma unroll
            for (int o = 0; o < TW; ++o)
#pragma unroll
                for (int b = 0; b < BT; ++b) {
                    ob[o].v[b] = sqrt(ob[o].v[b]);
                    ob[o].x[b] = ob[o].x[b] * ob[o].v[b];
                    if constexpr (BT > 16) ob[o].x[b] = uniform_d(ob[o].x[b]);        // (32-band records: back to the scalar file, see above)
                }
        }
        [[maybe_unused]] const bool mcw = SEG && kv.seg_nrank > 1;      // many dictionary widths (class_flush below)
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
        // (badc: the entries may belong to a pattern without a mode -- only the group-by-group path of the segmented form and the finish
        //  see those, so the model loop's copies of this code do not carry the general ln-like)
        auto settle = [&](int o, bool act, double c2, int tag, auto badc) {
            double w;
            if (decltype(badc)::value && segbad) w = act ? exp_clamped(seg_lnl(c2, ndcur) - lref, tb) : 0.0;      // (wave-uniform, rare)
            else w = act ? exactw_tab(c2, tbx, std::integral_constant<bool, !EXACT>{}) : 0.0;
            if constexpr (SEG) w *= fk;                           // (the pattern's mode value relative to the reference)
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
                    if (am) { Cand e; e.lnl = c2; e.j = tag; e.pad = SEG ? (ndcur | (mcw ? kv.seg_rank[segcur] << 8 : 0)) : 0; ambw[(size_t)o * cap + hs.namb[o] + pre] = e; }
                    hs.namb[o] += np;
                    if constexpr (SEG) hs.namb[o] = __builtin_amdgcn_readfirstlane(hs.namb[o]);
                } else hs.namb[o] = -1;                           // overflowed: nothing more is stored, the object goes to the sweep
            }
        };
        // settle the LAST (up to) 64 entries of object o's buffer with all lanes (their order does not matter: nothing has to move)
        auto drain = [&](int o, auto badc) {
            if constexpr (S2) {
                // the last (up to) 128 entries, two per lane: the two exact weights are independent chains in one basic block (a single
                // chain of ~35 dependent fp64 operations left the vector pipe idle a third of the time at four waves per SIMD)
                const int n = hs.pend[o] < 128 ? hs.pend[o] : 128;
                const int rest = hs.pend[o] - n;
                const double ca = rc2[o * CAP + rest + lane], cb = rc2[o * CAP + rest + 64 + lane];
                const int ta = (int)rtag[o * CAP + rest + lane], tb2 = (int)rtag[o * CAP + rest + 64 + lane];
                const bool aa = lane < n, ab = lane + 64 < n;
                double wa, wb;
                if (decltype(badc)::value && segbad) {
                    wa = aa ? exp_clamped(seg_lnl(ca, ndcur) - lref, tb) : 0.0; wb = ab ? exp_clamped(seg_lnl(cb, ndcur) - lref, tb) : 0.0;
                } else {
                    wa = exactw_tab(ca, tbx, std::true_type{}); wb = exactw_tab(cb, tbx, std::true_type{});
                    wa = aa ? wa : 0.0; wb = ab ? wb : 0.0;
                }
                if constexpr (SEG) { wa *= fk; wb *= fk; }
                hs.Sc[o] += wa + wb;
                hs.wmx[o] = vmax_raw(hs.wmx[o], vmax_raw(wa, wb));
                if (wa > thr_def) unsafeAtomicAdd(&rows[o * acc_stride + ta + w0], wa);
                if (wb > thr_def) unsafeAtomicAdd(&rows[o * acc_stride + tb2 + w0], wb);
                const bool ama = aa && !(wa > thr_def) && wa > hs.wamb[o], amb = ab && !(wb > thr_def) && wb > hs.wamb[o];
                const unsigned long long ma = __ballot(ama), mb = __ballot(amb);
                if (ma | mb) {                                      // wave-uniform, rare
                    const int na = __builtin_popcountll(ma), nb = __builtin_popcountll(mb);
                    if (hs.namb[o] >= 0 && hs.namb[o] + na + nb <= cap) {
                        const int pa = __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
                        const int pb = __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
                        if (ama) { Cand e; e.lnl = ca; e.j = ta; e.pad = SEG ? (ndcur | (mcw ? kv.seg_rank[segcur] << 8 : 0)) : 0; ambw[(size_t)o * cap + hs.namb[o] + pa] = e; }
                        if (amb) { Cand e; e.lnl = cb; e.j = tb2; e.pad = SEG ? (ndcur | (mcw ? kv.seg_rank[segcur] << 8 : 0)) : 0; ambw[(size_t)o * cap + hs.namb[o] + na + pb] = e; }
                        hs.namb[o] += na + nb;
                    } else hs.namb[o] = -1;
                }
                hs.pend[o] = rest;
                return;
            }
            const int n = hs.pend[o] < 64 ? hs.pend[o] : 64;
            const int rest = hs.pend[o] - n;                      // < 64
            double c2 = rc2[o * CAP + rest + lane];
            int tag = (int)rtag[o * CAP + rest + lane];
            const bool act = lane < n;
            settle(o, act, c2, tag, badc);
            hs.pend[o] = SEG ? __builtin_amdgcn_readfirstlane(rest) : rest;
        };
        // SEG with many dictionary widths (kv.seg_nrank > 1; the segments are ordered by width class first): the histogram holds the
        // weights of ONE class.  When the class of the model stream changes -- and at the finish -- the row is divided by the class's
        // edge-truncated kernel mass per index (pdf.py:613-617; the reciprocals are tabulated), convolved with the class's kernel and
        // ADDED to the object's PDF row in global memory (L2: 5.6 KB per object, touched ~30 times), then cleared.  The convolution is
        // that of k_fused's class-sorted stack (pdf_stage_mc, fz_kernels.h): a lane owns 12 consecutive outputs, so one LDS read per
        // lane and tap feeds 12 FMAs.  Class 0 comes first in the stream (ranks are dense and ascending): its flush WRITES the row.
        [[maybe_unused]] auto class_flush = [&](int rk, double* row, double* gout) {
            constexpr int NACC = 12;
#if defined(FZ_DIAG_NOFLUSH)
            return;
#endif
            const int G = (int)kv.G, Gp = kv.mc_gp, W0 = kv.mc_w0;
            const int wc = kv.mc_width[rk], w2 = 2 * wc, sh = W0 - wc;
            const double* nt = kv.mc_norm + (size_t)rk * Gp;                // 1 / mass
            const double* kr = kv.kern + kv.mc_off[rk];
            const double ka = (lane <= w2) ? kr[lane] : 0.0;                // w2 + 1 <= 127 taps in two registers across the wave
            const double kb = (lane + 64 <= w2) ? kr[lane + 64] : 0.0;
            for (int k = lane; k < Gp; k += 64) row[k] = row[k] * nt[k];
            const int kal = __double2loint(ka), kah = __double2hiint(ka), kbl = __double2loint(kb), kbh = __double2hiint(kb);
            const double* r0 = row + sh + NACC * min(lane, (G - 1) / NACC);  // lanes past G repeat the last lane's reads (unused sums); reads end < Gp + 12
            double out[NACC], win[NACC];
#pragma unroll
            for (int a = 0; a < NACC; ++a) { out[a] = 0.0; win[a] = r0[a]; }
            for (int h = 0; h <= w2; h += NACC) {
#pragma unroll
                for (int u = 0; u < NACC; ++u) {
                    if (h + u <= w2) {                                      // wave-uniform
                        const int q = w2 - (h + u);
                        const double tap = (q < 64) ? __hiloint2double(__builtin_amdgcn_readlane(kah, q), __builtin_amdgcn_readlane(kal, q))
                                                    : __hiloint2double(__builtin_amdgcn_readlane(kbh, q - 64), __builtin_amdgcn_readlane(kbl, q - 64));
#pragma unroll
                        for (int a = 0; a < NACC; ++a) out[a] = fma(win[(a + u) % NACC], tap, out[a]);
                        win[u] = r0[NACC + h + u];                          // the entry that left the window makes room for the next one
                    }
                }
            }
            for (int k = lane; k < Gp; k += 64) row[k] = 0.0;
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const int t = NACC * lane + a;
                if (t < G) gout[t] = (rk == 0) ? out[a] : gout[t] + out[a];
            }
        };
        // SEG: a group of another mask pattern begins: the pattern's power, normalisation, classifier constants and effective object
        // row.  The candidate buffer is EMPTY here -- the step before a change settles everything (its entries' power and factor
        // are the old pattern's; `flush` in step_body: the loop's own drain site, so that this rare code holds no second copy of the
        // settle and leaves the registers of the model loop alone).
        auto seg_switch = [&](int sw) {
            // bands both sides observe (wave-uniform; said to be so where that helps the register allocation: with per-model errors the
            // hint costs 3 ms per 2.6e10 pairs, in the other modes it saves up to 4)
            constexpr bool UH = SRC::LMODE != 0;
            uint32_t jb = obits & kv.seg_mask[sw & 0x3fff];
            if constexpr (UH) jb = (uint32_t)__builtin_amdgcn_readfirstlane((int)jb);
            const int nd = __popc(jb);
            const bool dp = src.lp.dim_prior != 0;
            wpr = dp ? nd - KOFF : 0;
            if constexpr (UH) wpr = __builtin_amdgcn_readfirstlane(wpr);
            segbad = wpr == -1;
            if (wpr < 0) wpr = 1;                                      // (below -1 kok is false: the object goes to the sweep; the arithmetic stays finite)
            ndcur = nd;
            K = uniform_d((double)wpr);
            rK = uniform_d(wpr > 0 ? 1.0 / (double)wpr : 1.0);
            // (the classifier's constants and the mode value without a logarithm: K is a small integer here)
            static constexpr double LOG2K[9] = {0.0, 0.0, 1.0, 1.5849625007211562, 2.0, 2.321928094887362, 2.584962500721156, 2.807354922057604, 3.0};
            const double hk_ = 0.5 * K, l2k = LOG2K[wpr > 8 ? 8 : wpr];      // (wpr >= 0 here)
            hk23 = (float)(hk_ / 8388608.0);
            T0c = (float)(-hk_ * l2k + K * 0.7213475204444817 + hk_ * (-127.0 + 0.043));
            lgq = uniform_d(dp ? src.lp.lgtab[nd] : 0.5 * ((double)nd * FZ_LN2PI + oslv));
            // ln L at the mode chi2 = K: (K / 2) ln K - K / 2 - lgq
            const double dl = uniform_d(fma(hk_, l2k * 0.6931471805599453, -hk_) - lgq) - lref;      // the pattern's mode value against the reference: <= 0
            fk = uniform_d(exp_neg(dl, tb));
            hk23 = uniform_f(hk23);
            T0c = uniform_f(T0c + (float)(dl * 1.4426950408889634));
            tzero = wpr > 0 ? -INFINITY : T0c;
            if (segbad) {
                // no classifier for a likelihood without a mode: every pair of the pattern is settled (t finite and far below anything,
                // the bar at -inf while the pattern lasts), by its own ln-like against the reference
                hk23 = 0.f; T0c = -1e30f; tzero = INFINITY; fk = 1.0;
            }
            if constexpr (!EXACT) hs.tthr[0] = segbad ? -INFINITY : wave_maxf_dpp(hs.tmax[0]) + (ldrop0 - 2.f * tmarg);
            src.load_obj_lds(obj0, ob[0]);
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                const bool on = (jb >> b) & 1u;
                // a band either side masks: inverse variance 0 (modes Ai / B: it adds exactly nothing to chi2, inter and shape); with
                // per-model errors the VARIANCE term 2^(960 / BT) and flux 0 -- a masked MODEL band has y = ye^2 = 0 in the segment-ordered
                // records, so d = 0 and the band adds exactly nothing; an unobserved OBJECT band adds y^2 / 2^192 (5 bands; 2^120 at 8)
                // to a chi2 of order one: below its last bit for |y| < 1e20 (1e9), and the launcher declines model sets beyond that.
                // (No multiplier per band: five more wave-uniform doubles did not fit the scalar file and spilled in the model loop.)
                if constexpr (SRC::LMODE == 0) ob[0].v[b] = on ? ob[0].v[b] : SEG_VBIG;
                else ob[0].v[b] = on ? ob[0].v[b] : 0.0;
                ob[0].x[b] = on ? ob[0].x[b] : 0.0;
            }
            segcur = UH ? __builtin_amdgcn_readfirstlane(sw & 0x3fff) : (sw & 0x3fff);      // (without the pad-slots flag: a flagged group never passes for "the pattern in force")
        };

        // chi2 of one pair from the wave's (effective) object row: the forms of the model loop, also used by the segmented form's
        // look-ahead over the few patterns whose likelihood has no mode (below)
        auto pair_chi2 = [&](const typename SRC::MR& mm, const typename SRC::OR& oo) -> double {
                if constexpr (C2OP) {
                    double c = 0.0;
#pragma unroll
                    for (int b = 0; b < BT; ++b) { const double d = fma(-mm.y[b], oo.v[b], oo.x[b]); c = fma(d, d, c); }
                    if constexpr (EXACT) c = (c <= FZ_HIST_C2ZERO) ? 0.0 : c;      // (the screen form makes this decision in its classifier)
                    return c;
                } else if constexpr (C2OPB) {
                    double ys[BT], inter = 0.0, shape = 0.0;
#pragma unroll
                    for (int b = 0; b < BT; ++b) {
                        ys[b] = mm.y[b] * oo.v[b];
                        inter = fma(ys[b], oo.x[b], inter);
                        shape = fma(ys[b], ys[b], shape);
                    }
                    // shape == 0 (no usable band) -> nan / inf like NumPy: the IEEE division sits behind a WAVE-uniform branch (the empty
                    // asm keeps the compiler from turning it into a select, which puts the division's ~25 instructions into every step:
                    // mode B 64.4 -> 73.4 ms per 2.6e10 pairs when this code moved into a lambda)
                    const bool odd = !(shape > 1e-280 && shape < 1e280);
                    const double rc = rcp_nr<1>(shape);
                    double sc = inter * rc;
                    sc = fma(fma(-sc, shape, inter), rc, sc);                         // residual correction: n / n == 1 exactly
                    if (__ballot(odd) != 0ull) {
                        asm volatile("" ::: "memory");
                        if (odd) sc = inter / shape;
                    }
                    double c = 0.0;
#pragma unroll
                    for (int b = 0; b < BT; ++b) { const double d = fma(-sc, ys[b], oo.x[b]); c = fma(d, d, c); }
                    return c;
                } else {
                    // (SEG with per-model errors: the same mask-free arithmetic -- a masked band sits in the effective row as a huge
                    //  variance term and a zero flux, see seg_switch; chi2 up to C2ZERO is the exact zero of a self match)
                    double c = src.chi2_of(oo, mm);
                    if constexpr (SEG && SRC::LMODE == 0 && EXACT) c = (c <= FZ_HIST_C2ZERO) ? 0.0 : c;
                    return c;
                }
        };
        if constexpr (SEG) {
            // look-ahead: the best ln-like among the pairs of the patterns without a mode (few models: two-band records against an
            // object that misses one of them) joins the reference, so that every weight of the object stays <= 1
            if (kok && nbadm > 0) {                                    // wave-uniform
                const bool dp = src.lp.dim_prior != 0;
                double lb = -INFINITY;
                for (int s0 = 0; s0 < kv.seg_n; s0 += 64) {
                    const int sg = s0 + lane;
                    const bool isb = sg < kv.seg_n && dp && (__popc(obits & kv.seg_mask[sg < kv.seg_n ? sg : 0]) - KOFF == -1);
                    unsigned long long bm = __ballot(isb);
                    while (bm) {
                        const int sb = s0 + __builtin_ctzll(bm);
                        bm &= bm - 1;
                        seg_switch(sb);
                        const int j1 = kv.seg_start[sb + 1];
                        for (int j = kv.seg_start[sb] + lane; j < j1; j += 64) {          // (whole 64-slot groups)
                            typename SRC::MR mm;
                            src.load_model_rec16(j, mm);
                            double c2 = pair_chi2(mm, ob[0]);
                            if constexpr (C2OP || SRC::LMODE == 0) c2 = (c2 <= FZ_HIST_C2ZERO) ? 0.0 : c2;
                            const double l = seg_lnl(c2, ndcur);
                            lb = fmax(lb, posw[j] < 0 ? -INFINITY : l);
                        }
                    }
                }
                lref = uniform_d(fmax(lref, wave_max(lb)));
                kok = lref < INFINITY;                                 // (a chi2 of exactly 0 there: +inf in the reference, pdf.py:92)
                segcur = -1;
            }
        }
        // the bars follow the wave-wide best weight seen
        auto refresh_bars = [&]() {
#pragma unroll
            for (int o = 0; o < TW; ++o) {
                const float mx = wave_maxf_dpp(hs.tmax[o]);
                hs.tmax[o] = mx;
                hs.tthr[o] = mx + (ldrop0 - 2.f * tmarg);       // (the best's t and the pair's t each carry the margin)
                if constexpr (SEG) hs.tthr[o] = segbad ? -INFINITY : hs.tthr[o];
                // the ambiguous band starts at wt_thresh x the best EXACT weight settled so far (a lower bound of the final best:
                // what still waits in the buffer only raises it)
                hs.wamb[o] = wave_max_pos_hi(hs.wmx[o]) * (wt_thresh * 0.999);
            }
        };
        // one 64-model group against the wave's objects, records and label words in registers.  SLOW: the group may hold pad slots of
        // a segment (SEG, tiles in which the mask pattern changes)
        constexpr int MP = EXACT ? 1 : FZ_HIST_MP;
        static_assert(!SEG || MP == 1, "segments: one group per trip");
        auto step_body = [&](typename SRC::MR (&m)[MP], int (&ptag)[MP], int st, int t, auto tailc, auto slowc, bool flush) {
            constexpr bool TAIL = decltype(tailc)::value, SLOW = decltype(slowc)::value;
                double c2[MP][TW];
#pragma unroll
                for (int q = 0; q < MP; ++q)
#pragma unroll
                    for (int o = 0; o < TW; ++o) {
                        c2[q][o] = pair_chi2(m[q], ob[o]);
                        if (TAIL) c2[q][o] = (t * TILE + (st + q) * 64 + lane < M) ? c2[q][o] : 1e30;   // pad lanes: weight 0
                    }
                if constexpr (EXACT) {
                    const int j = t * TILE + st * 64 + lane;
#pragma unroll
                    for (int o = 0; o < TW; ++o) {
                        // every pair in fp64; the running best (of the exact weights) bounds what can still be stacked
                        const bool valid = (!TAIL || j < M) && (!SLOW || ptag[0] >= 0);     // (pad slots of a segment carry the sign bit)
                        double w;
                        if (SLOW && segbad) w = valid ? exp_clamped(seg_lnl(c2[0][o], ndcur) - lref, tb) : 0.0;
                        else w = valid ? exactw_tab(c2[0][o], tbx, std::false_type{}) : 0.0;
                        if constexpr (SEG) { w *= fk; ptag[0] &= 0xffff; }
                        hs.S[o] += w;
                        hs.Sc[o] = vmax_raw(hs.Sc[o], w);       // (EXACT: Sc holds the best weight seen: the bar of the ambiguous band, and ln-max at the end)
                        if (w > thr_def) unsafeAtomicAdd(&rows[o * acc_stride + ptag[0] + w0], w);
                        const bool am = valid && !(w > thr_def) && (w >= wt_thresh * 0.999 * hs.Sc[o]);
                        const unsigned long long mask = __ballot(am);
                        if (mask) {
                            const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                            const int np = __builtin_popcountll(mask);
                            if (hs.namb[o] >= 0 && hs.namb[o] + np <= cap) {
                                if (am) { Cand e; e.lnl = c2[0][o]; e.j = ptag[0]; e.pad = SEG ? (ndcur | (mcw ? kv.seg_rank[segcur] << 8 : 0)) : 0; ambw[(size_t)o * cap + hs.namb[o] + pre] = e; }
                                hs.namb[o] += np;
                    if constexpr (SEG) hs.namb[o] = __builtin_amdgcn_readfirstlane(hs.namb[o]);
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
                            tl[q][o] = ((C2OP || (SEG && SRC::LMODE == 0)) ? !(cf <= (float)FZ_HIST_C2ZERO) : (cf != 0.f)) ? tl[q][o] : tzero;
                            if (TAIL) tl[q][o] = (t * TILE + (st + q) * 64 + lane < M) ? tl[q][o] : -INFINITY;   // pad lanes: dropped whatever the bar
                            if constexpr (SLOW) tl[q][o] = (ptag[q] < 0) ? -INFINITY : tl[q][o];                   // pad slots of a segment: likewise
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
                            // (the segmented kernels run out of scalar registers: without this the fill level lives in a vector register and
                            //  every `while (pend >= DTHR)` below becomes a divergent loop)
                            if constexpr (SEG) hs.pend[o] = __builtin_amdgcn_readfirstlane(hs.pend[o]);
                        }
                        if constexpr (!FZ_HIST_REFRESH_TILE) {
                            if (++hs.tick == hs.next) {                           // steps 1, 2, 4, 8, 16, then every FZ_HIST_REFRESH-th
                                hs.next = hs.tick < 16 ? 2 * hs.tick : hs.tick + FZ_HIST_REFRESH;
                                refresh_bars();
                            }
                        }
#pragma unroll
                        for (int o = 0; o < TW; ++o) {
#if defined(FZ_DIAG_NODRAIN)
                            if (hs.pend[o] >= DTHR) { hs.pend[o] -= 64; }
#else
                    if constexpr (SLOW) { const int lim = flush ? 1 : DTHR; while (hs.pend[o] >= lim) drain(o, std::true_type{}); }     // (flush: a change of pattern follows)
                    else { while (hs.pend[o] >= DTHR) drain(o, std::false_type{}); }
#endif
                        }
                    }
                }
        };
        // slowc (SEG): the tile goes group by group through the code that can change the pattern; else it is taken to be "pure" (below)
        auto run_tile = [&](const double* cur, double* nxt, int t, auto tailc, auto slowc) {
            constexpr bool TAIL = decltype(tailc)::value;
            constexpr bool pure = !decltype(slowc)::value;
            constexpr bool LATE = FZ_HIST_STAGE_LATE && pure && hist_prefetch<SRC>() && !(SEG && SRC::NB + SRC::NVAL > FZ_HIST_SEG_PFMAX);
            if (!LATE || !work) { if (t + 1 < ntiles) nl_stage_tile<SRC, TILE, NT, true>(src, posw, t + 1, nxt, tid, wave); }
            if (work) {
                const int32_t* tags = reinterpret_cast<const int32_t*>(cur + TDR);
                // MP 64-model groups per trip: their chi2 chains and weights sit in one basic block (the appends, which
                // branch, follow), and the next trip's records are requested before the current ones are used
                static_assert((TILE / 64) % MP == 0, "groups per trip must divide the tile");
                // (SEG with per-model errors: the next group's record is not requested ahead -- with the pattern's state beside the object
                //  row, a second copy of a 10-double record does not fit the register file: spills inside the loop, 2.7x slower)
                constexpr bool PF = hist_prefetch<SRC>() && !(SEG && SRC::NB + SRC::NVAL > FZ_HIST_SEG_PFMAX);
                if constexpr (pure) {
                    typename SRC::MR mn[PF ? MP : 1];
                    int tagn[MP];
                    if constexpr (PF) {
#pragma unroll
                        for (int q = 0; q < MP; ++q) {
                            src.template load_model_lds<TILE>(cur, q * 64 + lane, mn[q]);
                            tagn[q] = tags[q * 64 + lane];
                        }
                    }
                    if constexpr (LATE) { if (t + 1 < ntiles) nl_stage_tile<SRC, TILE, NT, true>(src, posw, t + 1, nxt, tid, wave); }
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
                        step_body(m, ptag, st, t, tailc, std::false_type{}, false);
                    }
                } else {
                    {
                        // group by group; a group whose SUCCESSOR belongs to another pattern empties the buffer when it is done (flush),
                        // so that the change finds nothing waiting.  The tile's first group may already be of a new pattern (the previous
                        // tile could not know): an all-pad step in front (st = -1) does the flushing for it.  (The last group's successor
                        // lies in the next tile, which is then not "pure" and starts with that step.)
#pragma unroll 1
                        for (int st = -1; st < TILE / 64; ++st) {
                            typename SRC::MR m[MP];
                            int ptag[MP];
                            const int stc = st < 0 ? 0 : st;
                            src.template load_model_lds<TILE>(cur, stc * 64 + lane, m[0]);
                            ptag[0] = st < 0 ? (int)0x80000000 : tags[stc * 64 + lane];
                            const int nxt = st + 1 < TILE / 64 ? (tags[(st + 1) * 64] >> 16) & 0x3fff : segcur;      // (wave-uniform read)
                            if (st >= 0) {
                                const int sw = (__builtin_amdgcn_readfirstlane(ptag[0]) >> 16) & 0x7fff;
                                if ((sw & 0x3fff) != segcur) {                     // a few dozen times per pass; nothing waits in the buffer
                                    if (mcw && segcur >= 0) {
                                        const int rko = kv.seg_rank[segcur];
                                        if (kv.seg_rank[sw & 0x3fff] != rko) class_flush(rko, rows, pdfs + (omap ? (int64_t)omap[i0] : i0) * kv.G);
                                    }
                                    seg_switch(sw);
                                }
                            }
                            const bool flush = __builtin_amdgcn_readfirstlane(nxt) != segcur;        // (segcur: this group's pattern by now)
                            step_body(m, ptag, stc, t, tailc, std::true_type{}, flush);
                        }
                    }
                }
                if constexpr (FZ_HIST_REFRESH_TILE && !EXACT) {
                    if (++hs.tick == hs.next) { hs.next = hs.tick < 3 ? hs.tick + 1 : hs.tick + 5; refresh_bars(); }
                }
            }
#if !defined(FZ_DIAG_NOBARRIER)
            __syncthreads();
#endif
        };
        if constexpr (!SEG) {
            for (int t = 0; t < ntiles; t += 2) {
                if (t + 1 < ntiles) run_tile(tileA, tileB, t, std::false_type{}, std::false_type{}); else run_tile(tileA, tileB, t, std::true_type{}, std::false_type{});
                if (t + 1 < ntiles) {
                    if (t + 2 < ntiles) run_tile(tileB, tileA, t + 1, std::false_type{}, std::false_type{}); else run_tile(tileB, tileA, t + 1, std::true_type{}, std::false_type{});
                }
            }
        } else {
            // SEG: a tile whose four groups all belong to the pattern in force and hold no pad slot -- all but a few per cent of them -- is
            // "pure" and runs the plain loop.  Runs of pure tiles have a loop of their own (two per trip, A then B), so that the loop-carried
            // registers of the model loop are reconciled with those of the group-by-group code only where a run ends, not after every
            // tile (one loop with a branch per tile: ~60 register moves per tile).  The segment-ordered set is padded to whole tiles:
            // there is no partial tile.  (Every wave stages and meets the barrier once per tile whichever way it takes.)
            auto tile_pure = [&](const double* cur) -> bool {
                if (!work) return true;
                const int wv = (reinterpret_cast<const int32_t*>(cur + TDR)[(lane % (TILE / 64)) * 64] >> 16) & 0x7fff;      // (segment | pad-slots flag << 14) of group lane mod 4
                return __ballot(wv != segcur) == 0ull && !segbad;          // (segcur never carries the flag; a pattern without a mode: group by group)
            };
            int t = 0;
            while (t < ntiles) {
                if (!(t & 1)) {
                    while (t < ntiles) {
                        if (!tile_pure(tileA)) break;
                        run_tile(tileA, tileB, t, std::false_type{}, std::false_type{}); ++t;
                        if (t >= ntiles || !tile_pure(tileB)) break;
                        run_tile(tileB, tileA, t, std::false_type{}, std::false_type{}); ++t;
                    }
                    if (t >= ntiles) break;
                }
                // the tile that ended the run -- or the B tile after it, pure or not: back to the A, B rhythm
                if (t & 1) run_tile(tileB, tileA, t, std::false_type{}, std::true_type{});
                else run_tile(tileA, tileB, t, std::false_type{}, std::true_type{});
                ++t;
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
                    while (hs.pend[o] > 0) drain(o, std::integral_constant<bool, SEG>{});
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
                if constexpr (SEG) { if (mcw && segcur >= 0) class_flush(kv.seg_rank[segcur], row, pdfs + i * kv.G); }      // the last class
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
                        double l1, w1;
                        if constexpr (SEG) {
                            // the entry's own N_dim (Cand::pad, low byte): power, normalisation, and its weight against the reference
                            l1 = in1 ? seg_lnl(e1.lnl, e1.pad & 0xff) : -INFINITY;
                            w1 = exp_neg(l1 - lref, tb);
                        } else {
                            l1 = in1 ? lnl_c2(e1.lnl) : -INFINITY;
                            w1 = exactw_tab(e1.lnl, tb, std::false_type{});
                        }
                        const bool s1 = in1 && (exp_neg(l1 - le, tb) > thr);   // strict
                        if constexpr (SEG) {
                            if (mcw) {
                                // many widths: the row is a PDF row by now (every class flushed, below): the entry's whole window goes in,
                                // pdf.py:599-620 as written (a lane per entry; LDS atomics)
                                if (s1) {
                                    const int rk = e1.pad >> 8, wc = kv.mc_width[rk], p = e1.j - kv.mc_w0;
                                    const double* kr = kv.kern + kv.mc_off[rk];
                                    const double wn = w1 * kv.mc_norm[(size_t)rk * kv.mc_gp + e1.j];
                                    const int G = (int)kv.G;
                                    for (int h = 0; h <= 2 * wc; ++h) {
                                        const int tq = p - wc + h;
                                        if (tq >= 0 && tq < G) unsafeAtomicAdd(&row[tq], wn * kr[h]);
                                    }
                                }
                                continue;
                            }
                        }
                        if (s1) unsafeAtomicAdd(&row[e1.j + w0], w1);
                    }
                }
                if (lane == 0) {
                    if (lmap) lmap[i] = lbest;
                    if (levid) levid[i] = le;
                    if (!ok) redo[1 + atomicAdd(redo, 1)] = (int)i;
                }
                if constexpr (SEG) {
                    if (mcw) {
                        // the flushed classes (global row) + the ambiguous entries' windows (LDS row), normalised
                        double* gout = pdfs + i * kv.G;
                        const int G = (int)kv.G;
                        if (!ok) { for (int tq = lane; tq < G; tq += 64) gout[tq] = NAN; continue; }
                        double tot = 0.0;
                        for (int tq = lane; tq < G; tq += 64) { const double vq = gout[tq] + row[tq]; row[tq] = vq; tot += vq; }
                        tot = wave_sum(tot);
                        const double sc = normalize ? 1.0 / tot : 1.0 / stot;
                        for (int tq = lane; tq < G; tq += 64) gout[tq] = normalize ? row[tq] / tot : row[tq] * sc;
                        continue;
                    }
                }
                kde_finalize<true>(kv, row, ok, normalize, pdfs + i * kv.G, lane, ok ? 1.0 / stot : 1.0, true);
            }
        }
        __syncthreads();
    }
}

}  // namespace fz
