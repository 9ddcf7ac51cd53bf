// Device-side arithmetic of the likelihood path, shared by every kernel.
// gfx950 only: wave64, fp64 VALU.  No CUDA/portable paths.
//
// Reference arithmetic restated here (joshspeagle/frankenz v0.3.5):
//   mode 0  "A"   fixed scale, model errors kept      pdf.py:76-98
//   mode 1  "Ai"  fixed scale, model errors ignored   pdf.py:76-77, 82-98
//   mode 2  "B"   free scale,  model errors ignored   pdf.py:171-194, 226-235
//   (mode C, free scale with model errors, iterates:  fz_modec.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fz_fastmath.h"

#define FZ_WAVE 64
#define FZ_LN2PI 1.8378770664093453   // log(2*pi)
#define FZ_LN2 0.6931471805599453

namespace fz {

// ---- views of device-resident data -----------------------------------------
// Models: structure-of-arrays, band-major, padded to Mp = ceil256(M) so that a
// wave's 64 lanes read 512 contiguous bytes per band (coalesced).
struct ModelView {
    const double* y;       // [BT][Mp]  model flux
    const double* ye2;     // [BT][Mp]  model error squared
    const double* ye;      // [BT][Mp]  model error (mode C squares scale*ye, pdf.py:201-202)
    // array-of-records copies for the LDS-tiled kernel: one record per model, a
    // contiguous tile is a straight copy and a lane reads its record with
    // conflict-free ds_read_b128 (record sizes are 16 mod 32 bytes)
    const double* rec0;    // [Mp][rw0]  y[0..BT) ye2[0..BT) pad      (mode A)
    const double* rec1;    // [Mp][rw1]  y[0..BT) pad                 (modes Ai, B)
    const uint32_t* bits;  // [Mp]      bit b = models_mask[j][b] != 0 (pad bands 0)
    int64_t M, Mp;
};
// Objects: row-major (N,BT), cleaned (pdf.py:309-311) and pre-derived once per
// chunk; read wave-uniformly (scalar loads -> SGPR operands).
struct ObjView {
    const double* x;       // flux
    const double* v;       // mode 0 / C: xe^2 ; modes 1,2: 1/xe^2
    const uint32_t* bits;  // data_mask bits
    const double* slv;     // sum over the B real bands of log(xe^2) (UNMASKED, pdf.py:96-98)
};
// Additive ln-prior (extension; SURVEY 8f-1 -- the lnprior a custom lprob_func returns,
// bruteforce.py:193-194): row-major table of ln-prior rows over the M models and the row
// each object of the chunk reads.  lnprob[i][j] = lnlike[i][j] + tab[row(i) * ld + j].
struct PriorView {
    const double* tab;     // nullptr: no prior
    const int64_t* rows;   // [n] row per object (validated against P on the device), or nullptr
    int64_t ident;         // rows == nullptr: 1 -> object i reads row i, 0 -> every object reads row 0
    int64_t ld;            // row length (= M)
    __device__ __forceinline__ int64_t row(int64_t i) const { return rows ? rows[i] : (ident ? i : 0); }
};
struct LikeParams {
    int dim_prior;
    const double* lgtab;   // [BT+1] gammaln(a)+a*ln2, a=n/2 (modes 0,1) or (n-1)/2 (mode 2)
    double lg_full;        // lgtab[B] for the all-unmasked fast path
    int nband;             // real band count B (<= BT)
};

struct PairOut { double lnl, chi2, scale, shape; int ndim; };

// xlogy(a-1, chi2) - chi2/2 - gammaln(a) - a ln2      (pdf.py:92-93, 228-229)
// FASTV: the mask-free tame-data variant, where am1 = B/2 - 1 (or (B-1)/2 - 1) with
// B = 5 is never zero and chi2 is +0, finite or nan.
template <bool FASTV = false>
__device__ __forceinline__ double chi2_logpdf(double am1, double chi2, double lg, const FastTabs& tb) {
    double xl = am1 * log_pos_t<FASTV>(chi2, tb);
    if (!FASTV && am1 == 0.0) xl = (chi2 == chi2) ? 0.0 : chi2;      // xlogy(0, y) = 0 unless y is nan
    return fma(-0.5, chi2, xl) - lg;
}

// VAR selects the arithmetic variant of one (BT, MODE) kernel:
//   0  no masked band anywhere, |flux| < 1e30 and variances in [1e-30,1e30]: mask-free code, ONE
//      reciprocal per pair (the B per-band quotients are summed over a common
//      denominator), Newton-refined v_rcp_f64
//   1  masks present (or B padded up to BT), tame variances: per-band masked terms
//      with Newton-refined reciprocals
//   2  anything else (zero/huge/non-finite variances): IEEE division throughout
enum { VAR_FAST = 0, VAR_MASKED = 1, VAR_SAFE = 2,
       VAR_PAD = 3,      // host-side only: tame, no REAL band masked, but the band count is padded up to 12 / 16 / 24 / 32 -- the one-pass kernel
                         // runs its mask-free form (pad bands are zeros and add nothing), every other kernel its masked variant
       VAR_OBJMASK = 4,  // host-side only, a REQUEST: objects with unobserved bands against unmasked models in modes Ai / B -- try the one-pass
                         // kernel with per-object band counts (masked bands carry inverse variance 0); +1 = not applicable, nothing was launched
       VAR_SEG = 5 };    // host-side only, a REQUEST: masked models, or unobserved object bands against per-model errors -- try the one-pass
                         // kernel on the segmented model layout (fz_hist.h, SEG); +1 = not applicable, nothing was launched

template <int BT, int MODE, int VAR>
struct Phot {
    static constexpr bool MASKED = (VAR != VAR_FAST);
    static constexpr bool SAFE = (VAR == VAR_SAFE);
    ModelView mv;
    ObjView ov;
    LikeParams lp;
    FastTabs tb;           // set by the kernel (LDS or global copy of the log/exp tables)

    struct MR { double y[BT]; double ye2[BT]; uint32_t bits; };
    struct OR { double x[BT]; double v[BT]; uint32_t bits; double slv; };

    __device__ __forceinline__ void load_model(int64_t j, MR& m) const {
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            m.y[b] = mv.y[(int64_t)b * mv.Mp + j];
            if (MODE == 0) m.ye2[b] = mv.ye2[(int64_t)b * mv.Mp + j];
        }
        m.bits = MASKED ? mv.bits[j] : 0xffffffffu;
    }
    // i must be wave-uniform
    __device__ __forceinline__ void load_obj(int64_t i, OR& o) const {
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            o.x[b] = ov.x[i * BT + b];
            o.v[b] = ov.v[i * BT + b];
        }
        o.bits = MASKED ? ov.bits[i] : 0xffffffffu;
        o.slv = ov.slv[i];
    }
    // Object rows parked in LDS by the owning wave ((2 BT + 2) doubles each) and read
    // back with broadcast ds_reads inside the model loop: keeps TW objects' rows out
    // of the register file for the whole loop.
    static constexpr int OBJ_DOUBLES = 2 * BT + 2;
    __device__ __forceinline__ void park_obj(int64_t i, double* dst, int lane) const {
        if (lane < BT) { dst[lane] = ov.x[i * BT + lane]; dst[BT + lane] = ov.v[i * BT + lane]; }
        if (lane == BT) { dst[2 * BT] = ov.slv[i]; dst[2 * BT + 1] = __hiloint2double(0, MASKED ? (int)ov.bits[i] : -1); }
    }
    __device__ __forceinline__ void load_obj_lds(const double* p, OR& o) const {
        // p is 16-B aligned and wave-uniform: broadcast ds_read_b128
        const double2* q = reinterpret_cast<const double2*>(p);
        double v[OBJ_DOUBLES];
#pragma unroll
        for (int k = 0; k < OBJ_DOUBLES / 2; ++k) { const double2 w = q[k]; v[2 * k] = w.x; v[2 * k + 1] = w.y; }
#pragma unroll
        for (int b = 0; b < BT; ++b) { o.x[b] = v[b]; o.v[b] = v[BT + b]; }
        o.slv = v[2 * BT];
        o.bits = MASKED ? (uint32_t)__double2loint(v[2 * BT + 1]) : 0xffffffffu;
    }

    // num/den: one Newton step on v_rcp_f64 (2e-15 relative, tests/test_hip_fastmath.py)
    static __device__ __forceinline__ double quot(double num, double den) {
        return SAFE ? num / den : num * rcp_nr<1>(den);
    }

    // DPT: 1 / 0 pins dim_prior at compile time (hot loops, which are unswitched on
    // it), -1 reads it from lp.
    template <int DPT = -1>
    __device__ __forceinline__ PairOut eval(const OR& o, const MR& m) const {
        const bool dim_prior = DPT < 0 ? (lp.dim_prior != 0) : (DPT != 0);
        PairOut r;
        uint32_t jb = o.bits & m.bits;
        r.ndim = MASKED ? __popc(jb) : lp.nband;
        double tm[BT];
        if (MASKED) {
#pragma unroll
            for (int b = 0; b < BT; ++b) tm[b] = ((jb >> b) & 1u) ? 1.0 : 0.0;
        }
        double chi2 = 0.0, slogv = 0.0;
        r.scale = 1.0; r.shape = 0.0;
        if (MODE == 0) {
            // tot_var = xe^2 + ye^2 ; chi2 = sum_b m (x-y)^2 / tot_var
            double vprod = 1.0; int vexp = 0;
            if (!MASKED) {
                // sum_b t_b/v_b over a common denominator: one reciprocal per pair.  Bands are
                // combined pairwise (a tree, not a chain) to keep the dependency depth low:
                // (n,d)(a) + (n,d)(b) = (n_a d_b + n_b d_a, d_a d_b)
                double tn[BT], tv[BT];
#pragma unroll
                for (int b = 0; b < BT; ++b) {
                    tv[b] = o.v[b] + m.ye2[b];
                    const double d = o.x[b] - m.y[b];
                    tn[b] = d * d;
                }
                int cntb = BT;
#pragma unroll
                for (int lvl = 0; lvl < 5; ++lvl) {
                    if (cntb > 1) {
                        const int half = cntb / 2;
#pragma unroll
                        for (int p = 0; p < half; ++p) {
                            const double na = tn[2 * p], nb = tn[2 * p + 1], da = tv[2 * p], db = tv[2 * p + 1];
                            tn[p] = fma(na, db, nb * da);
                            tv[p] = da * db;
                        }
                        if (cntb & 1) { tn[half] = tn[cntb - 1]; tv[half] = tv[cntb - 1]; }
                        cntb = half + (cntb & 1);
                    }
                }
                const double num = tn[0], den = tv[0];
                chi2 = num * rcp_nr<1>(den);
                if (!dim_prior) {              // uniform branch; log of the product = sum of logs
                    int e; vprod = frexp(den, &e); vexp = e;
                }
            } else {
#pragma unroll
                for (int b = 0; b < BT; ++b) {
                    const double v = o.v[b] + m.ye2[b];
                    const double d = o.x[b] - m.y[b];
                    chi2 = fma(quot(d * d, v), tm[b], chi2);
                    if (b < lp.nband && !dim_prior) {      // uniform branch
                        int e; vprod *= frexp(v, &e); vexp += e;
                    }
                }
            }
            if (!dim_prior) slogv = log_pos(vprod, tb) + (double)vexp * FZ_LN2;
        } else if (MODE == 1) {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                const double d = o.x[b] - m.y[b];
                if (MASKED) chi2 = fma((d * d) * o.v[b], tm[b], chi2);
                else chi2 = fma(d * d, o.v[b], chi2);       // sub, mul, fma: three instructions per band
            }
            slogv = o.slv;
        } else {
            // inter = sum m (y x)/var ; shape = sum m (y y)/var ; s = inter/shape.
            // Both use the SAME y/var factor, so a model identical to the data gives
            // inter == shape bit for bit and s == 1 exactly, as in the reference.
            double inter = 0.0, shape = 0.0;
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                const double w = MASKED ? o.v[b] * tm[b] : o.v[b];
                const double yw = m.y[b] * w;            // shared factor: three instructions per band, and x == y still gives inter == shape
                inter = fma(yw, o.x[b], inter);
                shape = fma(yw, m.y[b], shape);
            }
            double s;
            if (SAFE || !(shape > 1e-280 && shape < 1e280)) {
                s = inter / shape;                 // shape == 0 (no usable band) -> nan/inf like NumPy
            } else {
                const double rc = rcp_nr<1>(shape);
                s = inter * rc;
                s = fma(fma(-s, shape, inter), rc, s);    // residual correction: n/n == 1 exactly
            }
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                const double d = fma(-s, m.y[b], o.x[b]);
                if (MASKED) chi2 = fma((d * d) * o.v[b], tm[b], chi2);
                else chi2 = fma(d * o.v[b], d, chi2);    // three instructions per band; d == 0 (self match) still gives exactly 0
            }
            r.scale = s; r.shape = shape;
            slogv = o.slv;
        }
        r.chi2 = chi2;
        const double nd = (double)r.ndim;
        if (dim_prior) {
            const double a = (MODE == 2) ? 0.5 * (nd - 1.0) : 0.5 * nd;
            const double lg = MASKED ? lp.lgtab[r.ndim] : lp.lg_full;
            r.lnl = chi2_logpdf<VAR == VAR_FAST>(a - 1.0, chi2, lg, tb);
        } else {
            r.lnl = -0.5 * chi2 - 0.5 * (nd * FZ_LN2PI + slogv);
        }
        return r;
    }
};

// ---- wave64 reductions --------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// online (max, sum-exp) state; NaNs never become the max (Python's builtin max
// skips them unless first, bruteforce.py:359/619).  They DO poison logsumexp in the
// reference; the kernels flag them separately and force levid = nan at the end.
struct MS { double m, s; };
__device__ __forceinline__ void ms_init(MS& a) { a.m = -INFINITY; a.s = 0.0; }
// branch-free: d = l - m is nan when l is nan or both are the same infinity, and
// then e ~ 0 and nothing changes; l = -inf gives e ~ 0 as well.
__device__ __forceinline__ void ms_push(MS& a, double l, const FastTabs& tb) {
    const double d = l - a.m;
    const double e = exp_neg(-fabs(d), tb);     // nan d -> ~0: nans are flagged by the caller
    const bool up = d > 0.0;
    const double s_up = fma(a.s, e, 1.0), s_dn = a.s + e;
    a.s = up ? s_up : s_dn;
    a.m = up ? l : a.m;
}
__device__ __forceinline__ MS ms_merge(const MS& a, const MS& b) {
    MS r;
    if (b.m == -INFINITY && b.s == 0.0) return a;
    if (a.m == -INFINITY && a.s == 0.0) return b;
    r.m = fmax(a.m, b.m);
    r.s = a.s * exp_neg(a.m - r.m) + b.s * exp_neg(b.m - r.m);
    return r;
}
__device__ __forceinline__ MS wave_ms(MS a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MS b; b.m = __shfl_xor(a.m, o, 64); b.s = __shfl_xor(a.s, o, 64);
        a = ms_merge(a, b);
    }
    return a;
}

}  // namespace fz
