// Monte-Carlo k-nearest-neighbour variant (reference frankenz/knn.py).
//
//   k_knn_query  : what the K scipy KDTree.query calls of knn.py:834-837 return -- the
//                  k nearest models (p-norm 2) of every object in each of the K
//                  Monte-Carlo feature sets -- by exact brute force.
//   k_knn_subset : knn.py:840-872 for one object per wave: first-appearance de-dup of
//                  its K*k neighbour row (pandas.unique), likelihood on that subset
//                  (same arithmetic as the brute-force kernels), logsumexp, weights,
//                  threshold, kernel stack, normalise.
#pragma once
#include "fz_device.h"
#include "fz_kernels.h"

namespace fz {

// ---------------------------------------------------------------------------
// exact top-k by streaming: a lane owns one candidate model per step; each of the
// wave's TQ queries keeps its current k best (distance, index) sorted across lanes
// 0..k-1 and the k-th distance tau as the admission bar.  After a short warm-up
// almost no candidate beats tau (expected k*ln(M/k) admissions per query), so the
// steady state is: FT loads, 2*FT fp64 ops, one compare, one ballot per step.
// ---------------------------------------------------------------------------
template <int FT, int TQ>
__global__ __launch_bounds__(256) void k_knn_query(const float* __restrict__ feats, int64_t Mp, int M,
                                                   const double* __restrict__ q, int64_t N, int F, int k,
                                                   double bound2, int64_t* __restrict__ idx, int K, int pnorm) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * TQ;
    if (i0 >= N) return;
    const int tree = blockIdx.y;
    const float* ft = feats + (size_t)tree * FT * Mp;
    double qv[TQ][FT];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int64_t i = i0 + u < N ? i0 + u : N - 1;
#pragma unroll
        for (int f = 0; f < FT; ++f) qv[u][f] = f < F ? q[i * F + f] : 0.0;
    }
    double ld[TQ], tau[TQ];
    int lj[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) { ld[u] = INFINITY; lj[u] = M; tau[u] = bound2; }

    for (int jb = 0; jb < M; jb += 64) {
        const int j = jb + lane;
        double p[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) p[f] = (double)ft[(size_t)f * Mp + j];     // Mp padded: always in range
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            // pnorm 2: squared Euclidean (bound2 is the squared bound); 1: sum |d|; 0: max |d|
            double d2 = 0.0;
            if (pnorm == 2) {
#pragma unroll
                for (int f = 0; f < FT; ++f) { const double d = qv[u][f] - p[f]; d2 = fma(d, d, d2); }
            } else if (pnorm == 1) {
#pragma unroll
                for (int f = 0; f < FT; ++f) d2 += fabs(qv[u][f] - p[f]);
            } else {
#pragma unroll
                for (int f = 0; f < FT; ++f) d2 = fmax(d2, fabs(qv[u][f] - p[f]));
            }
            unsigned long long mask = __ballot(j < M && d2 < tau[u]);
            while (mask) {                                   // rare after warm-up
                const int sl = __builtin_ctzll(mask);
                mask &= mask - 1;
                const double dn = __shfl(d2, sl, 64);
                if (!(dn < tau[u])) continue;                // bar moved while draining this step
                const int jn = jb + sl;
                const int pos = __builtin_popcountll(__ballot(lane < k && ld[u] <= dn));
                const double ud = __shfl_up(ld[u], 1, 64);
                const int uj = __shfl_up(lj[u], 1, 64);
                if (lane > pos) { ld[u] = ud; lj[u] = uj; }
                else if (lane == pos) { ld[u] = dn; lj[u] = jn; }
                const double kth = __shfl(ld[u], k - 1, 64);
                tau[u] = kth < bound2 ? kth : bound2;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int64_t i = i0 + u;
        if (i < N && lane < k) idx[(i * K + tree) * k + lane] = (ld[u] < bound2) ? lj[u] : M;
    }
}

// ---------------------------------------------------------------------------
// The same search for the Euclidean norm with an fp32 SCREEN: the squared distances of two
// queries at a time are formed with packed fp32 (v_pk_add_f32 / v_pk_fma_f32: twice the fp64
// rate), compared with an admission bar that is provably above the exact one, and only the
// rare lanes that pass recompute their distance in fp64 for the exact test -- so the result is
// the exact fp64 top-k of k_knn_query, bit for bit.
//
// Bar: the features are float32 already; the query q is rounded to qf = float(q) with
// |q - qf| <= delta = 2^-24 max|q|.  If D^2 = sum (q-p)^2 < tau then
// sum (qf-p)^2 <= (sqrt(tau) + sqrt(F) delta)^2 = tau + 2 sqrt(F tau) delta + F delta^2, and the
// fp32 evaluation of that sum carries a relative error below (F+2) 2^-24 < 1e-6.
// ---------------------------------------------------------------------------
typedef float fz_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float knn_bar32(double tau, double delta, int F) {
    if (!(tau < 1e300)) return INFINITY;
    const double b = (tau + 2.0 * sqrt((double)F * tau) * delta + (double)F * delta * delta) * (1.0 + 4e-6);
    float f = (float)b;
    if ((double)f < b) f = __uint_as_float(__float_as_uint(f) + 1u);      // next float up (f is positive and finite here)
    return f;
}

template <int FT, int TQ>
__global__ __launch_bounds__(256) void k_knn_query32(const float* __restrict__ feats, int64_t Mp, int M,
                                                     const double* __restrict__ q, int64_t N, int F, int k,
                                                     double bound2, int64_t* __restrict__ idx, int K) {
    static_assert(TQ % 2 == 0, "queries are screened in pairs");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * TQ;
    if (i0 >= N) return;
    const int tree = blockIdx.y;
    const float* ft = feats + (size_t)tree * FT * Mp;
    double qv[TQ][FT], delta[TQ];
    fz_f2 qp[TQ / 2][FT];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int64_t i = i0 + u < N ? i0 + u : N - 1;
        double mx = 0.0;
#pragma unroll
        for (int f = 0; f < FT; ++f) { qv[u][f] = f < F ? q[i * F + f] : 0.0; mx = fmax(mx, fabs(qv[u][f])); }
        delta[u] = mx * 6.0e-8;                                   // >= 2^-24 max|q|
    }
#pragma unroll
    for (int h = 0; h < TQ / 2; ++h)
#pragma unroll
        for (int f = 0; f < FT; ++f) qp[h][f] = fz_f2{(float)qv[2 * h][f], (float)qv[2 * h + 1][f]};
    double ld[TQ], tau[TQ];
    float bar[TQ];
    int lj[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) { ld[u] = INFINITY; lj[u] = M; tau[u] = bound2; bar[u] = knn_bar32(bound2, delta[u], F); }

    for (int jb = 0; jb < M; jb += 64) {
        const int j = jb + lane;
        float p[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) p[f] = ft[(size_t)f * Mp + j];             // Mp padded: always in range
        fz_f2 acc[TQ / 2];
#pragma unroll
        for (int h = 0; h < TQ / 2; ++h) {
            fz_f2 a = {0.f, 0.f};
#pragma unroll
            for (int f = 0; f < FT; ++f) { const fz_f2 d = qp[h][f] - fz_f2{p[f], p[f]}; a = __builtin_elementwise_fma(d, d, a); }
            acc[h] = a;
        }
        bool passu[TQ], anyp = false;
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            const float s32 = (u & 1) ? acc[u >> 1].y : acc[u >> 1].x;
            passu[u] = j < M && s32 < bar[u];                      // nan never passes (as in the exact test)
            anyp |= passu[u];
        }
        if (!__any(anyp)) continue;                                // the common case after warm-up: one test per step
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            const bool pass = passu[u];
            if (!__any(pass)) continue;
            double d2 = INFINITY;
            if (pass) {                                            // exact distance, only in the lanes that passed
                d2 = 0.0;
#pragma unroll
                for (int f = 0; f < FT; ++f) { const double d = qv[u][f] - (double)p[f]; d2 = fma(d, d, d2); }
            }
            unsigned long long mask = __ballot(pass && d2 < tau[u]);
            while (mask) {
                const int sl = __builtin_ctzll(mask);
                mask &= mask - 1;
                const double dn = __shfl(d2, sl, 64);
                if (!(dn < tau[u])) continue;                      // bar moved while draining this step
                const int jn = jb + sl;
                const int pos = __builtin_popcountll(__ballot(lane < k && ld[u] <= dn));
                const double ud = __shfl_up(ld[u], 1, 64);
                const int uj = __shfl_up(lj[u], 1, 64);
                if (lane > pos) { ld[u] = ud; lj[u] = uj; }
                else if (lane == pos) { ld[u] = dn; lj[u] = jn; }
                const double kth = __shfl(ld[u], k - 1, 64);
                tau[u] = kth < bound2 ? kth : bound2;
                bar[u] = knn_bar32(tau[u], delta[u], F);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int64_t i = i0 + u;
        if (i < N && lane < k) idx[(i * K + tree) * k + lane] = (ld[u] < bound2) ? lj[u] : M;
    }
}

// ---------------------------------------------------------------------------
// subset likelihood + PDF, one object per wave
// ---------------------------------------------------------------------------
// capacities per launch: wcap = K*k rounded up to a power of two (>= 64), hash table of 2 wcap slots (open addressing)
#define FZ_KNN_WMAX 4096          // largest K*k handled (the reference takes any; knn.py:190-193)
#define FZ_KNN_KMAX 256           // largest k: the matrix-pipe Euclidean search keeps longer lists in 64-entry segments; the other norms / wide feature sets stop at 64
__host__ __device__ inline int fz_knn_wcap(int W) { int c = 64; while (c < W) c <<= 1; return c; }
// doubles of LDS per wave of k_knn_subset: the list, then max(hash table, ln-likelihoods + accumulation row)
__host__ __device__ inline size_t fz_knn_subset_lds_doubles(int acc_stride, int W) {
    const size_t wcap = (size_t)fz_knn_wcap(W), a = 2 * wcap, b = wcap + (size_t)acc_stride;
    return wcap / 2 + (a > b ? a : b);
}

struct KnnOut {                    // padded outputs of knn.py:812-821 (any may be NULL)
    int64_t* neighbors;            // (N,W) -99 padded
    int64_t* nnbr;                 // (N)
    double* lnlike; double* chi2; int64_t* ndim; double* scale; double* serr;     // (N,W)
    double* pdfs; double* lmap; double* levid;
};

template <class PH>
__global__ __launch_bounds__(256) void k_knn_subset(PH ph_, const KdeView* __restrict__ kvp, int acc_stride,
                                                    int64_t N, int M, const int64_t* __restrict__ idx, int W,
                                                    int free_scale, double wt_thresh, int normalize, KnnOut out,
                                                    int* __restrict__ errflag) {
    // LDS per wave: list[wcap] ints, then ONE region that first holds the de-dup hash table (key[2 wcap], pos[2 wcap] ints) and, once
    // the list is made, the ln-likelihoods lnl[wcap] and the accumulation row[acc_stride] (doubles): 11.8 KB instead of 20 KB at
    // K k = 500 on a 701-point grid -- the kernel waits on gathers 2/3 of its time and LDS bounds its occupancy
    // (fz_knn_subset_lds_doubles sizes the launch)
    extern __shared__ double smem[];
    PH ph = ph_;
    ph.tb = global_tabs();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (i >= N) return;
    const int wcap = fz_knn_wcap(W), hcap = 2 * wcap, hshift = 32 - (31 - __builtin_clz(hcap));
    const size_t per_wave = fz_knn_subset_lds_doubles(acc_stride, W);
    int* list = reinterpret_cast<int*>(smem + wave * per_wave);
    double* lnls = smem + wave * per_wave + wcap / 2;
    double* row = lnls + wcap;
    int* hkey = reinterpret_cast<int*>(lnls);
    int* hpos = hkey + hcap;
    const KdeView kv = *kvp;

    // ---- pandas.unique: keep first appearances, in order (knn.py:840) ----
    for (int s = lane; s < hcap; s += 64) { hkey[s] = -1; hpos[s] = 0x7fffffff; }
    const int64_t* myrow = idx + i * W;
    bool bad = false;
    auto put = [&](int p, long long v) __attribute__((always_inline)) {       // p < W
        if (v < 0 || v >= M) bad = true;              // KDTree's "missing" index: models[M] raises in the reference
        const int key = (int)v;
        unsigned h = ((unsigned)key * 2654435761u) >> hshift;          // log2(hcap) bits
        while (true) {
            const int prev = atomicCAS(&hkey[h], -1, key);
            if (prev == -1 || prev == key) break;
            h = (h + 1) & (hcap - 1);
        }
        atomicMin(&hpos[h], p);
    };
    int nn = 0;                                        // wave-uniform running count
    auto take = [&](int p, long long v) __attribute__((always_inline)) {      // every lane of the wave; p >= W: no entry
        bool first = false; int key = 0;
        if (p < W) {
            key = (int)v;
            unsigned h = ((unsigned)key * 2654435761u) >> hshift;
            while (hkey[h] != key) h = (h + 1) & (hcap - 1);
            first = (hpos[h] == p);
        }
        const unsigned long long mask = __ballot(first);
        const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (first) list[nn + pre] = key;
        nn += __builtin_popcountll(mask);
    };
    if (W <= 512) {                                    // the whole row in registers: ONE gather latency instead of one per 64 entries, twice
        long long vr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vr[u] = (u * 64 + lane < W) ? myrow[u * 64 + lane] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u * 64 + lane < W) put(u * 64 + lane, vr[u]);
        if (__any(bad)) { if (lane == 0) atomicExch(errflag, 1); return; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u * 64 < W) take(u * 64 + lane, vr[u]);
    } else {
        for (int p0 = 0; p0 < W; p0 += 64) if (p0 + lane < W) put(p0 + lane, myrow[p0 + lane]);
        if (__any(bad)) { if (lane == 0) atomicExch(errflag, 1); return; }
        for (int p0 = 0; p0 < W; p0 += 64) take(p0 + lane, p0 + lane < W ? myrow[p0 + lane] : 0);
    }
    if (out.nnbr && lane == 0) out.nnbr[i] = nn;
    if (out.neighbors) for (int s = lane; s < W; s += 64) out.neighbors[i * W + s] = s < nn ? list[s] : -99;

    // ---- likelihood on the subset (knn.py:847-849), running max / sum-exp ----
    typename PH::OR ob;
    ph.load_obj(i, ob);
    MS st; ms_init(st);
    bool isnan0 = false, anynan = false;
    // (records, and the next 64 models' requested before this round's are evaluated: the kernel waits on these gathers)
    typename PH::MR mnext;
    ph.load_model_rec16(lane < nn ? list[lane] : 0, mnext);
    for (int s0 = 0; s0 < W; s0 += 64) {
        const int s = s0 + lane;
        const bool in = s < nn;
        const typename PH::MR m = mnext;
        if (s0 + 64 < W) ph.load_model_rec16(s + 64 < nn ? list[s + 64] : 0, mnext);
        const PairOut r = ph.eval(ob, m);
        const double l = in ? r.lnl : -INFINITY;
        if (s < W) {
            const int64_t o = i * W + s;
            if (out.lnlike) out.lnlike[o] = in ? r.lnl : -INFINITY;            // knn.py:814-816 padding
            if (out.chi2) out.chi2[o] = in ? r.chi2 : INFINITY;
            if (out.ndim) out.ndim[o] = in ? r.ndim : 0;
            if (out.scale) out.scale[o] = (in && free_scale) ? r.scale : 1.0;
            if (out.serr) out.serr[o] = (in && free_scale) ? sqrt(1.0 / r.shape) : 0.0;
            lnls[s] = l;
        }
        if (l != l) { anynan = true; if (s == 0) isnan0 = true; }
        ms_push(st, l, ph.tb);
    }
    if (!out.pdfs) return;
    const bool fn = __any(isnan0), an = __any(anynan);
    const double mx = wave_max(st.m);
    const double ss = wave_sum(st.s * exp_neg(st.m - mx, ph.tb));
    const double lm = fn ? (double)NAN : mx;
    const double le = an ? (double)NAN : (mx == INFINITY ? (double)INFINITY : mx + log(ss));
    if (lane == 0) { if (out.lmap) out.lmap[i] = lm; if (out.levid) out.levid[i] = le; }
    // ---- weights, threshold, kernel stack (knn.py:862-872) ----
    const bool ok = (le - le == 0.0);
    if (ok) {
        for (int t = lane; t < acc_stride; t += 64) row[t] = 0.0;
        const double thr = wt_thresh * exp_neg(mx - le, ph.tb);
        for (int s0 = 0; s0 < nn; s0 += 64) {
            const int s = s0 + lane;
            const bool in = s < nn;
            const double w = exp_neg((in ? lnls[s] : -INFINITY) - le, ph.tb);
            kde_scatter(kv, row, in && (w > thr), w, in ? list[s] : 0, lane);
        }
    }
    kde_finalize(kv, row, ok, normalize, out.pdfs + i * kv.G, lane);
}

// pandas.unique alone (knn.py:840): neighbour table -> first-appearance lists + counts.
// Used when the likelihood itself runs elsewhere (the iterative mode C).
static __global__ __launch_bounds__(256) void k_knn_dedup(int64_t N, int M, const int64_t* __restrict__ idx, int W,
                                                   int64_t* __restrict__ neighbors, int64_t* __restrict__ nnbr,
                                                   int* __restrict__ errflag) {
    extern __shared__ double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (i >= N) return;
    const int wcap = fz_knn_wcap(W), hcap = 2 * wcap, hshift = 32 - (31 - __builtin_clz(hcap));
    int* list = reinterpret_cast<int*>(smem) + (size_t)wave * (wcap + 2 * hcap);
    int* hkey = list + wcap;
    int* hpos = hkey + hcap;
    for (int s = lane; s < hcap; s += 64) { hkey[s] = -1; hpos[s] = 0x7fffffff; }
    const int64_t* myrow = idx + i * W;
    bool bad = false;
    for (int p0 = 0; p0 < W; p0 += 64) {
        const int p = p0 + lane;
        if (p < W) {
            const long long v = myrow[p];
            if (v < 0 || v >= M) bad = true;
            const int key = (int)v;
            unsigned h = ((unsigned)key * 2654435761u) >> hshift;
            while (true) {
                const int prev = atomicCAS(&hkey[h], -1, key);
                if (prev == -1 || prev == key) break;
                h = (h + 1) & (hcap - 1);
            }
            atomicMin(&hpos[h], p);
        }
    }
    if (__any(bad)) { if (lane == 0) atomicExch(errflag, 1); return; }
    int nn = 0;
    for (int p0 = 0; p0 < W; p0 += 64) {
        const int p = p0 + lane;
        bool first = false; int key = 0;
        if (p < W) {
            key = (int)myrow[p];
            unsigned h = ((unsigned)key * 2654435761u) >> hshift;
            while (hkey[h] != key) h = (h + 1) & (hcap - 1);
            first = (hpos[h] == p);
        }
        const unsigned long long mask = __ballot(first);
        const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (first) list[nn + pre] = key;
        nn += __builtin_popcountll(mask);
    }
    if (lane == 0) nnbr[i] = nn;
    for (int s = lane; s < W; s += 64) neighbors[i * W + s] = s < nn ? list[s] : -99;
}

// NearestNeighbors._predict (knn.py:488-558): PDFs from stored (N,W) ln-weights and
// the stored neighbour table; one object per wave.
static __global__ __launch_bounds__(256) void k_knn_predict(const KdeView* __restrict__ kvp, int acc_stride, int64_t N, int M,
                                                     const double* __restrict__ logwt,
                                                     const int64_t* __restrict__ nbr, const int64_t* __restrict__ nnbr,
                                                     int W, double wt_thresh, int normalize, double* __restrict__ pdfs,
                                                     double* __restrict__ lmap, double* __restrict__ levid,
                                                     int* __restrict__ errflag) {
    extern __shared__ double smem[];
    const FastTabs tb = global_tabs();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (i >= N) return;
    double* row = smem + (size_t)wave * acc_stride;
    const KdeView kv = *kvp;
    const int nn = (int)nnbr[i];
    if (nn < 0 || nn > W) { if (lane == 0) atomicExch(errflag, 1); return; }
    MS st; ms_init(st);
    bool isnan0 = false, anynan = false, bad = false;
    for (int s0 = 0; s0 < nn; s0 += 64) {
        const int s = s0 + lane;
        const bool in = s < nn;
        const double l = in ? logwt[i * W + s] : -INFINITY;
        if (in) { const long long j = nbr[i * W + s]; if (j < 0 || j >= M) bad = true; }
        if (l != l) { anynan = true; if (s == 0) isnan0 = true; }
        ms_push(st, l, tb);
    }
    if (__any(bad)) { if (lane == 0) atomicExch(errflag, 1); return; }
    const bool fn = __any(isnan0), an = __any(anynan);
    const double mx = wave_max(st.m);
    const double ss = wave_sum(st.s * exp_neg(st.m - mx, tb));
    const double lm = fn ? (double)NAN : mx;
    const double le = an ? (double)NAN : (mx == INFINITY ? (double)INFINITY : mx + log(ss));
    if (lane == 0) { if (lmap) lmap[i] = lm; if (levid) levid[i] = le; }
    const bool ok = (le - le == 0.0);
    if (ok) {
        for (int t = lane; t < acc_stride; t += 64) row[t] = 0.0;
        const double thr = wt_thresh * exp_neg(mx - le, tb);
        for (int s0 = 0; s0 < nn; s0 += 64) {
            const int s = s0 + lane;
            const bool in = s < nn;
            const double w = exp_neg((in ? logwt[i * W + s] : -INFINITY) - le, tb);
            kde_scatter(kv, row, in && (w > thr), w, in ? nbr[i * W + s] : 0, lane);
        }
    }
    kde_finalize(kv, row, ok, normalize, pdfs + i * kv.G, lane);
}

}  // namespace fz
