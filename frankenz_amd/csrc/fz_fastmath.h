// fp64 transcendental/division helpers sized for this path (gfx950).
//
// The likelihood path needs log(chi2), exp(lnl - max) and 1/var per object-model
// pair.  The OCML versions are correctly rounded-ish and cost ~100 / ~50 / ~11
// fp64 instructions; these cost ~15 / ~12 / ~3 and keep ~1e-15 relative accuracy
// (absolute for log), seven orders of magnitude inside the 1e-5 parity bar.
// Accuracy is pinned by tests/test_hip_fastmath.py against NumPy.
#pragma once
#include <hip/hip_runtime.h>
#include "fz_tables.h"

namespace fz {

// max(a, b) as the bare v_max_f64 (IEEE mode: a quiet NaN operand yields the other one).
// fmax() makes the compiler canonicalise any operand it cannot prove quiet-NaN-free -- an
// extra v_max_f64 x, x per call on loop-carried values; the hot loops use this instead.
__device__ __forceinline__ double vmax_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double vmin_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// (the s_nop is the wait state gfx950 needs between a transcendental result -- v_exp_f32, v_log_f32,
// v_rcp_f32 ... -- and the next vector instruction that reads it: the compiler pads its own
// instructions, it does not look inside an asm statement, and the callers feed exp2 results in)
__device__ __forceinline__ float vmaxf_raw(float a, float b) {
    float r;
    asm("s_nop 0\n\tv_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// DX9-rule multiply: 0 * anything (inf, nan included) = 0
__device__ __forceinline__ float mul_legacy(float a, float b) {
    float r;
    asm("s_nop 0\n\tv_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a wave-uniform value the compiler cannot see is uniform (a reduction result, an LDS read): into SGPRs
__device__ __forceinline__ double uniform_d(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ float uniform_f(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

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
    const double* expt;      // [FZ_EXP_K]  2^(j/FZ_EXP_K)
};
#define FZ_TABS_DOUBLES (256 + FZ_EXP_K)
__device__ __forceinline__ FastTabs global_tabs() {
    FastTabs t; t.logt = reinterpret_cast<const double2*>(FZ_LOG_TAB); t.expt = FZ_EXP_TAB; return t;
}
// copy the tables into LDS at `dst` (FZ_TABS_DOUBLES doubles); caller synchronises
__device__ __forceinline__ FastTabs stage_tabs(double* dst, int tid, int nthreads) {
    for (int k = tid; k < 256; k += nthreads) dst[k] = FZ_LOG_TAB[k];
    for (int k = tid; k < FZ_EXP_K; k += nthreads) dst[256 + k] = FZ_EXP_TAB[k];
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

// exp(x) for |x| <= 700 (the callers clamp).  n = round(x K/ln2), K = 2048, is read off the low
// mantissa word of x K/ln2 + 1.5*2^52 (no rint / cvt); r = x - n ln2/K with ONE constant
// (|r| <= 1.7e-4; the constant is off by 3.4e-17 relative, which puts 3.4e-17 |x| into the
// result -- a third of the rounding error 1.1e-16 |x| that x itself carries); exp(r) is the
// Taylor cubic (remainder 3.4e-17); 2^(n mod K / K) from the table; 2^(n div K) is added to the
// exponent field (the result stays normal for x >= -700).
// Relative error < 4e-16 + 3.4e-17 |x|  (tests/test_hip_fastmath.py).
__device__ __forceinline__ double exp_core(double x, const FastTabs& tb) {
    const double MAGIC = 6755399441055744.0;                     // 1.5 * 2^52
    const double d = fma(x, 2954.639443740597, MAGIC);           // K/ln2
    const double r = fma(d - MAGIC, -0.0003384507717577858, x);  // ln2/K
    const int n = __double2loint(d);
    const double t = tb.expt[n & (FZ_EXP_K - 1)];
    double p = fma(r, 1.0 / 6.0, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double v = t * p;                                      // in [1,2): exponent field 1023
    return __hiloint2double(__double2hiint(v) + ((n >> 11) << 20), __double2loint(v));
}
// exp(x) for x <= 0 (what the softmax needs).  Arguments below -700 are clamped, i.e.
// return exp(-700) ~ 1e-304 instead of underflowing towards 0: harmless for sums of
// weights (callers never rely on an exact 0) and it keeps 2^q a plain exponent-field
// add instead of an ldexp.  NaN input is NOT propagated (callers track NaNs separately).
__device__ __forceinline__ double exp_neg(double x, const FastTabs& tb) {
    return exp_core(vmax_raw(x, -700.0), tb);                    // also maps NaN -> -700
}
__device__ __forceinline__ double exp_neg(double x) { return exp_neg(x, global_tabs()); }

// exp(x) with x clamped to [-700, 700] (NaN -> -700): same algorithm, either sign.
__device__ __forceinline__ double exp_clamped(double x, const FastTabs& tb) {
    return exp_core(vmin_raw(vmax_raw(x, -700.0), 700.0), tb);
}

}  // namespace fz
