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

#define FZ_WAVE 64
#define FZ_LN2PI 1.8378770664093453   // log(2*pi)
#define FZ_LN2 0.6931471805599453

namespace fz {

// ---- views of device-resident data -----------------------------------------
// Models: structure-of-arrays, band-major, padded to Mp = ceil64(M) so that a
// wave's 64 lanes read 512 contiguous bytes per band (coalesced).
struct ModelView {
    const double* y;       // [BT][Mp]  model flux
    const double* ye2;     // [BT][Mp]  model error squared
    const double* ye;      // [BT][Mp]  model error (mode C squares scale*ye, pdf.py:201-202)
    const uint32_t* bits;  // [Mp]      bit b = models_mask[j][b] != 0 (pad bands 0)
    int64_t M, Mp;
};
// Objects: row-major (N,BT), cleaned (pdf.py:309-311) and pre-derived once per
// chunk; read wave-uniformly (scalar loads).
struct ObjView {
    const double* x;       // flux
    const double* v;       // mode 0: xe^2 ; modes 1,2: 1/xe^2
    const double* xw;      // mode 2: x/xe^2
    const uint32_t* bits;  // data_mask bits
    const double* slv;     // sum over the B real bands of log(xe^2) (UNMASKED, pdf.py:96-98)
};
struct LikeParams {
    int dim_prior;
    const double* lgtab;   // [BT+1] gammaln(a)+a*ln2, a=n/2 (modes 0,1) or (n-1)/2 (mode 2)
    double lg_full;        // lgtab[B] for the all-unmasked fast path
    int nband;             // real band count B (<= BT)
};

struct PairOut { double lnl, chi2, scale, shape; int ndim; };

// xlogy(a-1, chi2) - chi2/2 - gammaln(a) - a ln2      (pdf.py:92-93, 228-229)
__device__ __forceinline__ double chi2_logpdf(double am1, double chi2, double lg) {
    double xl = (am1 == 0.0 && chi2 == chi2) ? 0.0 : am1 * log(chi2);
    return xl - 0.5 * chi2 - lg;
}

template <int BT, int MODE, bool MASKED>
struct Phot {
    ModelView mv;
    ObjView ov;
    LikeParams lp;

    struct MR { double y[BT]; double ye2[BT]; uint32_t bits; };
    struct OR { double x[BT]; double v[BT]; double xw[BT]; uint32_t bits; double slv; };

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
            if (MODE == 2) o.xw[b] = ov.xw[i * BT + b];
        }
        o.bits = MASKED ? ov.bits[i] : 0xffffffffu;
        o.slv = ov.slv[i];
    }

    __device__ __forceinline__ PairOut eval(const OR& o, const MR& m) const {
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
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                double v = o.v[b] + m.ye2[b];
                double d = o.x[b] - m.y[b];
                double q = (d * d) / v;
                chi2 = MASKED ? fma(q, tm[b], chi2) : chi2 + q;
                if (b < lp.nband && !lp.dim_prior) {   // uniform branch
                    int e; vprod *= frexp(v, &e); vexp += e;
                }
            }
            if (!lp.dim_prior) slogv = log(vprod) + (double)vexp * FZ_LN2;
        } else if (MODE == 1) {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                double d = o.x[b] - m.y[b];
                double q = (d * d) * o.v[b];
                chi2 = MASKED ? fma(q, tm[b], chi2) : chi2 + q;
            }
            slogv = o.slv;
        } else {
            // inter = sum m y x / var ; shape = sum m y^2 / var ; s = inter/shape
            double inter = 0.0, shape = 0.0;
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                double yi = m.y[b] * o.xw[b];
                double ys = (m.y[b] * m.y[b]) * o.v[b];
                inter = MASKED ? fma(yi, tm[b], inter) : inter + yi;
                shape = MASKED ? fma(ys, tm[b], shape) : shape + ys;
            }
            double s = inter / shape;
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                double d = fma(-s, m.y[b], o.x[b]);
                double q = (d * d) * o.v[b];
                chi2 = MASKED ? fma(q, tm[b], chi2) : chi2 + q;
            }
            r.scale = s; r.shape = shape;
            slogv = o.slv;
        }
        r.chi2 = chi2;
        double nd = (double)r.ndim;
        if (lp.dim_prior) {
            double a = (MODE == 2) ? 0.5 * (nd - 1.0) : 0.5 * nd;
            double lg = MASKED ? lp.lgtab[r.ndim] : lp.lg_full;
            r.lnl = chi2_logpdf(a - 1.0, chi2, lg);
        } else {
            r.lnl = -0.5 * chi2 - 0.5 * (nd * FZ_LN2PI + slogv);
        }
        return r;
    }
};

// ---- wave64 reductions (DPP/permute via __shfl_xor) ---------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// online (max, sum-exp) state; NaNs never become the max (Python's builtin max
// skips them unless first, bruteforce.py:359/619) but do poison the sum
// (logsumexp -> nan).
struct MS { double m, s; };
__device__ __forceinline__ void ms_init(MS& a) { a.m = -INFINITY; a.s = 0.0; }
__device__ __forceinline__ void ms_push(MS& a, double l) {
    if (l == -INFINITY) return;                 // exp(-inf - m) = 0 contributes nothing
    double d = l - a.m;                         // nan if l is nan (or both +inf)
    double e = exp(-fabs(d));
    if (d > 0.0) { a.s = fma(a.s, e, 1.0); a.m = l; }
    else a.s += e;                              // d<=0 or nan
}
__device__ __forceinline__ MS ms_merge(const MS& a, const MS& b) {
    MS r;
    if (b.m == -INFINITY && b.s == 0.0) return a;
    if (a.m == -INFINITY && a.s == 0.0) return b;
    r.m = fmax(a.m, b.m);
    r.s = a.s * exp(a.m - r.m) + b.s * exp(b.m - r.m);
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
