// Mode C: free scale WITH model errors -- the fixed-point iteration of
// pdf.py:196-223 and its per-object GLOBAL stop rule
//     while max_j |lnl_new_j - lnl_j| > ltol
// (Python builtin max over an ndarray: NaNs are skipped unless element 0 is NaN).
//
// The rule couples all M models of one object, so the state (scale, lnl, chi2,
// shape) of a chunk of objects is kept in (Nc,M) planes in HBM and one kernel
// launch advances every still-active object by one iteration; a tiny kernel then
// applies the stop rule per object and lists the objects that go on (the next launch
// covers only those: the slowest object of a chunk can take hundreds of iterations
// after the rest have stopped).  Objects that have stopped are frozen, which
// reproduces the reference's per-object iteration count exactly.  The solve keeps
// NumPy's operation order (the library is built with -ffp-contract=off) so that the
// iterates, and therefore the stop decision, track the reference to the last ulps.
#pragma once
#include "fz_device.h"

namespace fz {

struct ModeCState {
    double* s;    // scale           (Nc,M)
    double* l;    // Gaussian lnl    (Nc,M)
    double* c;    // chi2            (Nc,M)
    double* sh;   // shape           (Nc,M)
    unsigned long long* err;   // (Nc) max |dlnl| as ordered bits
    int* firstnan;             // (Nc) |dlnl[0]| is NaN
    const int* list;           // objects this launch advances (nullptr: all Nc), written by the last check
    int* list_next;            // objects that go on (filled by k_modec_check)
    int* nactive;              // (1) length of list_next
    const int* ncur;           // (1) length of `list` on the device (nullptr: Nc): the host launches for an upper bound
    int* last_iter;            // (1) highest iteration after which some object still went on
};

// Optional indirection for the k-NN subset (knn.py:847-849): object i's "model" slot j is
// model nbr[i*W + j], valid while j < nnb[i].
struct SubsetView { const int64_t* nbr; const int64_t* nnb; int W; };

template <int BT, bool MASKED>
struct ModeC {
    ModelView mv;
    ObjView ov;       // v = xe^2
    int nband;
    SubsetView sub;   // sub.nbr == nullptr: all models

    // one solve of (scale, chi2, lnl) for variance var_b = xe2_b + (s_prev*ye_b)^2;
    // s_prev = 1 gives the initial pass of pdf.py:171-194.
    __device__ __forceinline__ void solve(int64_t i, int64_t j, double sprev, double& s, double& lnl,
                                          double& chi2, double& shape, int& ndim) const {
        if (sub.nbr) j = sub.nbr[i * sub.W + j];
        uint32_t jb = MASKED ? (ov.bits[i] & mv.bits[j]) : 0xffffffffu;
        ndim = MASKED ? __popc(jb) : nband;
        double var[BT], y[BT], x[BT], tm[BT];
        double inter = 0.0; shape = 0.0;
        // sum_b log(var_b) = log(prod_b mantissa_b) + (sum_b exponent_b) ln 2: ONE table log per solve
        // instead of B library logs (425 of the solve's ~660 instructions), to ~1e-15 absolute -- the
        // size of the difference between two correctly working log implementations, eleven orders
        // below ltol
        double vprod = 1.0; int vexp = 0;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            y[b] = mv.y[(int64_t)b * mv.Mp + j];
            x[b] = ov.x[i * BT + b];
            tm[b] = MASKED ? (((jb >> b) & 1u) ? 1.0 : 0.0) : 1.0;
            const double sye = sprev * mv.ye[(int64_t)b * mv.Mp + j];
            var[b] = ov.v[i * BT + b] + sye * sye;                  // xe^2 + (s*ye)^2
            inter += tm[b] * y[b] * x[b] / var[b];
            shape += tm[b] * (y[b] * y[b]) / var[b];
            if (b < nband) { int e; vprod *= frexp(var[b], &e); vexp += e; }    // unmasked, pdf.py:193-194
        }
        s = inter / shape;
        chi2 = 0.0;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            double d = x[b] - s * y[b];
            chi2 += tm[b] * (d * d) / var[b];
        }
        const double slog = log_pos(vprod, global_tabs()) + (double)vexp * FZ_LN2;
        lnl = -0.5 * chi2 - 0.5 * ((double)ndim * FZ_LN2PI + slog);
    }
};

template <class MC>
__global__ __launch_bounds__(256) void k_modec_step(MC mc, ModeCState st, int64_t Nc, int64_t M, int init) {
    const int64_t tiles = (M + 255) / 256;
    const int64_t slot = blockIdx.x / tiles;             // Nc = number of objects of this launch (an upper bound when st.ncur is set)
    if (st.ncur && slot >= *st.ncur) return;             // block-uniform: this slot's object stopped since the host last looked
    const int64_t i = st.list ? (int64_t)st.list[slot] : slot;
    const int64_t j = (blockIdx.x % tiles) * 256 + threadIdx.x;
    const bool valid = j < M && (!mc.sub.nnb || j < mc.sub.nnb[i]);
    double e = 0.0;
    if (valid) {
        const int64_t k = i * M + j;
        double s, l, c, sh; int nd;
        if (init) {
            mc.solve(i, j, 1.0, s, l, c, sh, nd);
        } else {
            const double lold = st.l[k];
            mc.solve(i, j, st.s[k], s, l, c, sh, nd);
            e = fabs(l - lold);
            if (j == 0 && e != e) st.firstnan[i] = 1;
        }
        st.s[k] = s; st.l[k] = l; st.c[k] = c; st.sh[k] = sh;
    }
    if (init) return;
    // block max, NaNs dropped (a > b is false for NaN)
    if (!(e == e)) e = 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e = fmax(e, __shfl_xor(e, o, 64));
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) {
        e = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        atomicMax(&st.err[i], (unsigned long long)__double_as_longlong(e));   // e >= 0: bit order == value order
    }
}

static __global__ void k_modec_check(ModeCState st, int64_t Nc, double ltol, int iter) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= Nc || (st.ncur && slot >= *st.ncur)) return;
    const int i = st.list ? st.list[slot] : (int)slot;
    const double e = __longlong_as_double((long long)st.err[i]);
    const bool go = !st.firstnan[i] && (e > ltol);       // `while lerr > ltol`
    st.err[i] = 0ull;
    if (go) { st.list_next[atomicAdd(st.nactive, 1)] = i; atomicMax(st.last_iter, iter); }
}

// after convergence: dim prior (pdf.py:226-229) and the output planes
static __global__ __launch_bounds__(256) void k_modec_final(ModeCState st, ModelView mv, SubsetView sub, const uint32_t* obits,
                                                     int masked, int nband, int dim_prior,
                                                     const double* lgtab, int64_t Nc, int64_t M,
                                                     double* lnl, double* chi2, int64_t* ndim,
                                                     double* scale, double* serr) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= Nc * M) return;
    const int64_t i = k / M, j = k % M;
    if (sub.nnb && j >= sub.nnb[i]) {                     // padding of knn.py:812-821
        if (lnl) lnl[k] = -INFINITY;
        if (chi2) chi2[k] = INFINITY;
        if (ndim) ndim[k] = 0;
        if (scale) scale[k] = 1.0;
        if (serr) serr[k] = 0.0;
        return;
    }
    const int64_t jm = sub.nbr ? sub.nbr[i * sub.W + j] : j;
    const int nd = masked ? __popc(obits[i] & mv.bits[jm]) : nband;
    double l = st.l[k];
    const double c = st.c[k];
    if (dim_prior) l = chi2_logpdf<false>(0.5 * ((double)nd - 1.0) - 1.0, c, lgtab[nd], global_tabs());
    if (lnl) lnl[k] = l;
    if (chi2) chi2[k] = c;
    if (ndim) ndim[k] = nd;
    if (scale) scale[k] = st.s[k];
    if (serr) serr[k] = sqrt(1.0 / st.sh[k]);
}

}  // namespace fz
