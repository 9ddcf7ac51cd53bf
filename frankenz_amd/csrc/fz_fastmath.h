// fp64 transcendental/division helpers sized for this path (gfx950).
//
// The likelihood path needs log(chi2), exp(lnl - max) and 1/var per object-model
// pair.  The OCML versions are correctly rounded-ish and cost ~100 / ~50 / ~11
// fp64 instructions; these cost ~15 / ~17 / ~3 and keep ~1e-15 relative accuracy
// (absolute for log), seven orders of magnitude inside the 1e-5 parity bar.
// Accuracy is pinned by tests/test_hip_fastmath.py against NumPy.
#pragma once
#include <hip/hip_runtime.h>
#include "fz_tables.h"

namespace fz {

// 1/v: hardware v_rcp_f64 seed + NITER Newton steps (each 2 FMAs).
template <int NITER>
__device__ __forceinline__ double rcp_nr(double v) {
    double r = __builtin_amdgcn_rcp(v);
#pragma unroll
    for (int k = 0; k < NITER; ++k) {
        const double e = fma(-v, r, 1.0);
        r = fma(r, e, r);
    }
    return r;
}

// Reduction tables.  The global copies (fz_tables.h) serve one-off uses; kernels that
// call log/exp per pair keep a copy in LDS (FastTabs) so that a lookup is one
// ds_read instead of a global load sitting in the vector-memory queue.
struct FastTabs {
    const double2* logt;     // [128] {1/c, log c}
    const double* expt;      // [64]  2^(j/64)
};
#define FZ_TABS_DOUBLES (256 + 64)
__device__ __forceinline__ FastTabs global_tabs() {
    FastTabs t; t.logt = reinterpret_cast<const double2*>(FZ_LOG_TAB); t.expt = FZ_EXP_TAB; return t;
}
// copy the tables into LDS at `dst` (FZ_TABS_DOUBLES doubles); caller synchronises
__device__ __forceinline__ FastTabs stage_tabs(double* dst, int tid, int nthreads) {
    for (int k = tid; k < 256; k += nthreads) dst[k] = FZ_LOG_TAB[k];
    for (int k = tid; k < 64; k += nthreads) dst[256 + k] = FZ_EXP_TAB[k];
    FastTabs t; t.logt = reinterpret_cast<const double2*>(dst); t.expt = dst + 256; return t;
}

// natural log for finite x > 0 (denormals included); x == 0 -> -inf, x == +inf -> +inf,
// NaN or x < 0 -> NaN.  |abs error| < 4e-16 + 2e-16*|log x|.
// TAME = true: the caller guarantees x is +0, a positive finite number or NaN (a chi2
// of range-checked data); only the x == 0 fix-up is applied, branch-free, so that two
// independent evaluations stay in one basic block and interleave.
template <bool TAME>
__device__ __forceinline__ double log_pos_t(double x, const FastTabs& tb) {
    const double mant = __builtin_amdgcn_frexp_mant(x);          // [0.5,1)
    const int ex = __builtin_amdgcn_frexp_exp(x);
    const unsigned hi = (unsigned)__double2hiint(mant);
    const unsigned idx = (hi >> 13) & 127u;                      // top 7 fraction bits
    const double2 t = tb.logt[idx];
    const double r = fma(mant, t.x, -1.0);                       // |r| <= 2^-8
    // log1p(r) = r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6
    // r + r^2 (-1/2 + r/3 + r^2 (-1/4 + r/5 - r^2/6)), Estrin form
    const double r2 = r * r;
    const double q01 = fma(r, 1.0 / 3.0, -0.5);
    const double q23 = fma(r, 0.2, -0.25);
    const double q = fma(r2, fma(r2, -1.0 / 6.0, q23), q01);
    const double p = fma(r2, q, r);
    double l = fma((double)ex, 0.6931471805599453, t.y) + p;
    if (TAME) {
        l = (x == 0.0) ? -INFINITY : l;
    } else {
        // +normal | +denormal is the only class the table path is valid for; anything
        // else is rare, so the fix-up sits behind a real wave-uniform branch (the empty
        // asm keeps the compiler from turning it back into selects).
        const bool ok = __builtin_amdgcn_class(x, 0x180);
        if (__ballot(!ok) != 0ull) {
            asm volatile("" ::: "memory");
            if (!ok) l = (x == 0.0) ? -INFINITY : ((x == INFINITY) ? INFINITY : NAN);
        }
    }
    return l;
}
__device__ __forceinline__ double log_pos(double x, const FastTabs& tb) { return log_pos_t<false>(x, tb); }
__device__ __forceinline__ double log_pos(double x) { return log_pos_t<false>(x, global_tabs()); }

// exp(x) for x <= 0 (what the softmax needs).  Arguments below -700 are clamped, i.e.
// return exp(-700) ~ 1e-304 instead of underflowing towards 0: harmless for sums of
// weights (callers never rely on an exact 0) and it keeps 2^q a plain exponent-field
// add instead of an ldexp.  NaN input is NOT propagated (callers track NaNs
// separately).  Relative error < 3e-16.
__device__ __forceinline__ double exp_neg(double x, const FastTabs& tb) {
    x = fmax(x, -700.0);                                         // also maps NaN -> -700
    const double k = rint(x * 92.33248261689366);                // 64/ln2
    double r = fma(k, -0.010830424696249145, x);                 // ln2/64 hi
    r = fma(k, -3.623510646634843e-19, r);                       // ln2/64 lo  (hi+lo good to 1e-35)
    const int ki = (int)k;
    const double t = tb.expt[ki & 63];
    // 1 + r + r^2/2 + r^3/6 + r^4/24 + r^5/120, Estrin form (dependency depth 3)
    const double r2 = r * r;
    const double p01 = 1.0 + r;
    const double p23 = fma(r, 1.0 / 6.0, 0.5);
    const double p45 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    double p = fma(r2, fma(r2, p45, p23), p01);
    const double v = t * p;                                      // in [1,2): exponent field 1023
    // v * 2^q with q = ki >> 6 in [-1010, 0]: add q to the exponent field
    return __hiloint2double(__double2hiint(v) + ((ki >> 6) << 20), __double2loint(v));
}
__device__ __forceinline__ double exp_neg(double x) { return exp_neg(x, global_tabs()); }

// exp(x) with x clamped to [-700, 700] (NaN -> -700): same algorithm, either sign.
__device__ __forceinline__ double exp_clamped(double x, const FastTabs& tb) {
    x = fmin(fmax(x, -700.0), 700.0);
    const double k = rint(x * 92.33248261689366);
    double r = fma(k, -0.010830424696249145, x);
    r = fma(k, -3.623510646634843e-19, r);
    const int ki = (int)k;
    const double t = tb.expt[ki & 63];
    // 1 + r + r^2/2 + r^3/6 + r^4/24 + r^5/120, Estrin form (dependency depth 3)
    const double r2 = r * r;
    const double p01 = 1.0 + r;
    const double p23 = fma(r, 1.0 / 6.0, 0.5);
    const double p45 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    double p = fma(r2, fma(r2, p45, p23), p01);
    const double v = t * p;
    return __hiloint2double(__double2hiint(v) + ((ki >> 6) << 20), __double2loint(v));
}

}  // namespace fz
