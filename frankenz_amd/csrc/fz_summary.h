// PDF summary statistics: pdf.pdfs_summarize (pdf.py:899-1074) and the population overlap
// likelihood samplers.loglike_nz (samplers.py:23-86) -- the consumers of the (N,G) PDF stack
// (SURVEY 8f rows 2 and 3).
//
//   k_rownorm    pdfs /= pdfs.sum(axis=1)                       (pdf.py:984-985)   HBM-bound
//   k_gemm_f64   risk = pdfs @ (1 - kernel)                     (pdf.py:1024)      fp64 MFMA
//   k_summarize  everything per object: mean / mode / CDF quantiles / argmin of the risk row /
//                second moments / windowed CDF mass / interpolated risk (pdf.py:987-1068)
//   k_overlap    overlap = pdfs @ nz (+ pair step), sum of logs (samplers.py:66-76) HBM-bound
//   k_nz_assign  one categorical draw per object from pdf_i * nz (samplers.py:498-499, 519-520)  HBM-bound
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace fz {

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// ---- row normalisation ------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_rownorm(double* __restrict__ p, int64_t N, int G) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    double* row = p + i * G;
    double s = 0.0;
    for (int k = lane; k < G; k += 64) s += row[k];
    s = wsum(s);
    for (int k = lane; k < G; k += 64) row[k] = row[k] / s;
}

// ---- C[M x N] = A[M x K] B[K x N], row-major fp64, v_mfma_f64_16x16x4_f64 -------------------
// 128 x 128 block tile, 4 waves of 64 x 64 (4 x 4 MFMA tiles, 64 accumulator doubles per lane),
// K in steps of 16 through double-buffered LDS tiles stored k-major so that an MFMA operand
// (A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15]) is one ds_read_b64 of
// 16 consecutive doubles per k.  Edges are zero-filled on load and masked on store.
#define FZ_GEMM_BM 128
#define FZ_GEMM_BN 128
#define FZ_GEMM_BK 16
#define FZ_GEMM_LD (FZ_GEMM_BM + 4)
static __global__ __launch_bounds__(256) void k_gemm_f64(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                                         int64_t ldb, double* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    extern __shared__ double smem[];                   // As[2][BK][LD] | Bs[2][BK][LD]
    double* As = smem;
    double* Bs = smem + 2 * FZ_GEMM_BK * FZ_GEMM_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * FZ_GEMM_BM;
    const int n0 = blockIdx.y * FZ_GEMM_BN;
    // global -> register staging: A: thread owns (row = tid >> 1, 8 consecutive k); B: (k = tid >> 4, 8 consecutive cols)
    const int arow = tid >> 1, akseg = (tid & 1) * 8;
    const int bk = tid >> 4, bcseg = (tid & 15) * 8;
    double ra[8], rb[8];
    auto gload = [&](int k0) {
        const int64_t gr = m0 + arow;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + akseg + q;
            ra[q] = (gr < M && k < K) ? A[gr * lda + k] : 0.0;
        }
        const int kk = k0 + bk;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cidx = n0 + bcseg + q;
            rb[q] = (kk < K && cidx < N) ? B[(int64_t)kk * ldb + cidx] : 0.0;
        }
    };
    auto sstore = [&](int buf) {
        double* a = As + buf * FZ_GEMM_BK * FZ_GEMM_LD;
        double* b = Bs + buf * FZ_GEMM_BK * FZ_GEMM_LD;
#pragma unroll
        for (int q = 0; q < 8; ++q) a[(akseg + q) * FZ_GEMM_LD + arow] = ra[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) b[bk * FZ_GEMM_LD + bcseg + q] = rb[q];
    };
    v4f64 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int nk = (K + FZ_GEMM_BK - 1) / FZ_GEMM_BK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) gload((t + 1) * FZ_GEMM_BK);
        const double* a = As + buf * FZ_GEMM_BK * FZ_GEMM_LD + wr * 64 + (lane & 15);
        const double* b = Bs + buf * FZ_GEMM_BK * FZ_GEMM_LD + wc * 64 + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < FZ_GEMM_BK; kk += 4) {
            const int krow = (kk + (lane >> 4)) * FZ_GEMM_LD;
            double av[4], bv[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) av[mi] = a[krow + mi * 16];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bv[ni] = b[krow + ni * 16];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
        if (t + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }
    // D layout (f64 16x16x4): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = n0 + wc * 64 + ni * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wr * 64 + mi * 16 + (lane >> 4) + 4 * r;
                if (row < M && col < N) C[row * ldc + col] = acc[mi][ni][r];
            }
        }
}

// ---- numpy.interp (compiled_base.c arr_interp) on device -----------------------------------
// xp non-decreasing (plateaus allowed); j = last index with xp[j] <= x.
template <class XP, class FP>
__device__ __forceinline__ double interp1(double x, const XP& xp, const FP& fp, int n) {
    if (x != x) return x;
    if (x < xp(0)) return fp(0);
    if (x > xp(n - 1)) return fp(n - 1);
    int lo = 0, hi = n;                                  // first index with xp > x
    while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (x >= xp(mid)) lo = mid + 1; else hi = mid; }
    const int j = lo - 1;
    if (j >= n - 1) return fp(n - 1);
    const double xj = xp(j), fj = fp(j);
    if (xj == x) return fj;
    const double fj1 = fp(j + 1), xj1 = xp(j + 1);
    const double slope = (fj1 - fj) / (xj1 - xj);
    double r = slope * (x - xj) + fj;
    if (r != r) {
        r = slope * (x - xj1) + fj1;
        if (r != r && fj == fj1) r = fj;
    }
    return r;
}

// ---- per-object statistics: one wave per object ----------------------------------------------
// stats rows (each of length N): 0-3 mean{value,std,conf,risk}, 4-7 median, 8-11 mode, 12-15 best,
// 16-19 low95, low68, high68, high95, 20 Monte-Carlo draw.
// One CDF row of G doubles per wave in LDS: four waves per block up to G = 4 800, two up to 9 600, one up to 19 200 (the launcher
// picks; pdf.py:899-1074 takes any grid -- beyond 19 200 points a row no longer fits the 160 KB).
#define FZ_SUM_MAXG 19200
static __global__ __launch_bounds__(256) void k_summarize(const double* __restrict__ pdfs, const double* __restrict__ risk,
                                                          int64_t N, int G, const double* __restrict__ grid,
                                                          const double* __restrict__ urand, const double* __restrict__ widths,
                                                          double wscale, int64_t ostride, double* __restrict__ stats) {
    extern __shared__ double smem[];                    // [waves per block][G] CDF rows
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (i >= N) return;
    double* cdf = smem + (size_t)wave * G;
    const double* p = pdfs + i * G;
    const double* rk = risk + i * G;
    const int CH = (G + 63) / 64;                       // contiguous chunk per lane
    const int k0 = lane * CH, k1 = min(G, k0 + CH);
    // mean (pdf.py:988), mode (pdf.py:991: first maximum), chunk sums for the CDF (pdf.py:994)
    double sp = 0.0, spg = 0.0, best = -INFINITY; int bidx = 0x7fffffff;
    for (int k = k0; k < k1; ++k) {
        const double v = p[k];
        sp += v; spg = fma(v, grid[k], spg);
        if (v > best || (v != v && best == best)) { best = v; bidx = k; }     // np.argmax: first max, nan wins
    }
    const double pmean = wsum(spg);
    {   // wave arg-max with the smallest index on ties
        double bv = best; int bi = (k0 < k1) ? bidx : 0x7fffffff;
        if (!(k0 < k1)) bv = -INFINITY;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const double ov = __shfl_xor(bv, s, 64); const int oi = __shfl_xor(bi, s, 64);
            const bool take = (ov > bv) || (ov == bv && oi < bi) || (ov != ov && bv == bv) || (ov != ov && bv != bv && oi < bi);
            if (take) { bv = ov; bi = oi; }
        }
        bidx = bi;
    }
    const double pmode = grid[bidx];
    // inclusive scan of the chunk sums -> CDF in LDS
    double inc = sp;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const double v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
    double run = inc - sp;
    for (int k = k0; k < k1; ++k) { run += p[k]; cdf[k] = run; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    auto CDF = [&](int k) { return cdf[k]; };
    auto GRD = [&](int k) { return grid[k]; };
    auto RSK = [&](int k) { return rk[k]; };
    // quantiles and the Monte-Carlo draw (pdf.py:999-1001)
    double q = 0.0;
    {
        const double qs[6] = {0.025, 0.16, 0.5, 0.84, 0.975, urand[i]};
        const double x = qs[lane < 6 ? lane : 0];
        q = interp1(x, CDF, GRD, G);
    }
    const double plow2 = __shfl(q, 0, 64), plow1 = __shfl(q, 1, 64), pmed = __shfl(q, 2, 64);
    const double phigh1 = __shfl(q, 3, 64), phigh2 = __shfl(q, 4, 64), pmc = __shfl(q, 5, 64);
    // "best": first minimum of the risk row (pdf.py:1025)
    double rbest = INFINITY; int ridx = 0x7fffffff;
    for (int k = k0; k < k1; ++k) {
        const double v = rk[k];
        if (v < rbest || (v != v && rbest == rbest)) { rbest = v; ridx = k; }
    }
    {
        double bv = (k0 < k1) ? rbest : INFINITY; int bi = (k0 < k1) ? ridx : 0x7fffffff;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const double ov = __shfl_xor(bv, s, 64); const int oi = __shfl_xor(bi, s, 64);
            const bool take = (ov < bv) || (ov == bv && oi < bi) || (ov != ov && bv == bv) || (ov != ov && bv != bv && oi < bi);
            if (take) { bv = ov; bi = oi; }
        }
        ridx = bi;
    }
    const double pbest = grid[ridx];
    const double est[4] = {pmean, pmed, pmode, pbest};
    // second moments around the four estimators (pdf.py:1028-1036)
    double sd[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = k0; k < k1; ++k) {
        const double v = p[k], g = grid[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const double d = g - est[e]; sd[e] = fma(d * d, v, sd[e]); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sd[e] = sqrt(wsum(sd[e]));
    // windowed CDF mass (pdf.py:1041-1062) and interpolated risk (pdf.py:1065-1068)
    double cval = 0.0, rval = 0.0;
    {
        const int e = (lane >> 1) & 3;
        const double pt = est[e];
        const double w = widths ? widths[i * 4 + e] : (1. + pt) * wscale;
        const double x = (lane & 1) ? pt + w : pt - w;
        if (lane < 8) cval = interp1(x, GRD, CDF, G);
        if (lane < 4) rval = interp1(est[lane & 3], GRD, RSK, G);
    }
    double conf[4], rsk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        conf[e] = __shfl(cval, 2 * e + 1, 64) - __shfl(cval, 2 * e, 64);
        rsk[e] = __shfl(rval, e, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            stats[(4 * e + 0) * ostride + i] = est[e];
            stats[(4 * e + 1) * ostride + i] = sd[e];
            stats[(4 * e + 2) * ostride + i] = conf[e];
            stats[(4 * e + 3) * ostride + i] = rsk[e];
        }
        stats[16 * ostride + i] = plow2; stats[17 * ostride + i] = plow1;
        stats[18 * ostride + i] = phigh1; stats[19 * ostride + i] = phigh2;
        stats[20 * ostride + i] = pmc;
    }
}

// ---- pdfs_resample (pdf.py:855-896): numpy.interp of each row onto a new grid ----------------
static __global__ __launch_bounds__(256) void k_resample(const double* __restrict__ pdfs, int64_t N, int G, const double* __restrict__ og,
                                                         int Gn, const double* __restrict__ ng, double left, double right,
                                                         int renormalize, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const double* p = pdfs + i * G;
    double* o = out + i * Gn;
    auto OG = [&](int k) { return og[k]; };
    auto PV = [&](int k) { return p[k]; };
    double s = 0.0;
    for (int k = lane; k < Gn; k += 64) {
        const double x = ng[k];
        double v;
        if (x != x) v = x;
        else if (x < og[0]) v = left;                     // np.interp(left=..., right=...)
        else if (x > og[G - 1]) v = right;
        else v = interp1(x, OG, PV, G);
        o[k] = v; s += v;
    }
    if (renormalize) {
        s = wsum(s);
        for (int k = lane; k < Gn; k += 64) o[k] = o[k] / s;
    }
}

// ---- population overlap (samplers.py:66-76): one wave per object ------------------------------
static __global__ __launch_bounds__(256) void k_overlap(const double* __restrict__ pdfs, int64_t N, int G,
                                                        const double* __restrict__ nz, int pi, int pj, double step,
                                                        double* __restrict__ overlap, double* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    __shared__ double part[4];
    double lg = 0.0;
    if (i < N) {
        const double* p = pdfs + i * G;
        double s = 0.0;
        for (int k = lane; k < G; k += 64) s = fma(p[k], nz[k], s);
        s = wsum(s);
        if (pi >= 0) s = s + step * (p[pi] - p[pj]);
        if (lane == 0) overlap[i] = s;
        lg = log(s);
    }
    if (lane == 0) part[threadIdx.x >> 6] = lg;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
// ---- per-object categorical draw of the hierarchical / population samplers ---------------------
// samplers.py:498-499, 519-520 draw  multinomial(1, p * pos / dot(p, pos))  for every object in
// every Gibbs sweep (N x G work) and sum the one-hot rows.  Here: the draw by inverse CDF with a
// uniform supplied by the caller (u[i] in [0, 1)) -- bin = the number of grid points whose running
// sum of p[g] * nz[g] is <= u * total -- one wave per object, the row read once; counts are summed
// with integer atomics (order-free).  A row without mass (total <= 0 or not finite) gets bin -1.
static __global__ __launch_bounds__(256) void k_nz_assign(const double* __restrict__ pdfs, int64_t N, int G,
                                                          const double* __restrict__ nz, const double* __restrict__ u,
                                                          int64_t* __restrict__ bins, unsigned long long* __restrict__ counts, int staged) {
    extern __shared__ double s_w[];                               // [4][G]: the wave's row of p[g] * nz[g]
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    // the row comes in with lanes ALONG the grid (512 contiguous bytes per load; a lane reading its own run straight from memory
    // strides the wave's 64 addresses by `per` doubles: 2 TB/s) and waits in LDS, weighted, for the lanes' runs
    // (staged == 0: a grid too long for four rows of LDS -- the lanes read their runs from memory)
    double* pl = s_w + (size_t)(threadIdx.x >> 6) * G;
    const double* pg = pdfs + i * G;
    if (staged) for (int g = lane; g < G; g += 64) pl[g] = pg[g] * nz[g];
    auto W = [&](int g) { return staged ? pl[g] : pg[g] * nz[g]; };
    // every lane owns a CONTIGUOUS run of grid points, so that the running sum is monotone across lanes
    const int per = (G + 63) / 64;
    const int g0 = lane * per, g1 = min(G, g0 + per);
    double s = 0.0;
    for (int g = g0; g < g1; ++g) s += W(g);
    // exclusive prefix over the lanes (Hillis-Steele on the lane sums)
    double incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const double t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const double total = __shfl(incl, 63, 64);
    int64_t bin = -1;
    if (total > 0.0 && total - total == 0.0) {
        const double target = u[i] * total;
        // the lane whose run holds the crossing: inclusive sum > target, exclusive sum <= target
        const double excl = incl - s;
        // (the owner must hold mass itself: the tree-ordered lane prefix is monotone only up to rounding)
        const unsigned long long mass = __ballot(s > 0.0), at = __ballot(incl > target) & mass;
        // rounding at the very top (u * total == total): the last lane that holds any mass, its last such point
        const int owner = at ? __builtin_ctzll(at) : 63 - __builtin_clzll(mass);
        if (lane == owner) {
            double c = excl; int g = g0, last = g0;
            for (; g < g1; ++g) { const double w = W(g); if (w > 0.0) last = g; c += w; if (at && c > target) break; }
            bin = (g < g1) ? g : last;                           // (run exhausted: by rounding, or at the top: its last point with mass)
            bins[i] = bin;
            atomicAdd(&counts[bin], 1ull);
        }
    } else if (lane == 0) bins[i] = -1;
}
// fixed-order sum of the block partials (one block): the result does not depend on scheduling
static __global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ partial, int64_t n, double* __restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int64_t k = threadIdx.x; k < n; k += 256) s += partial[k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = sh[0];
}

}  // namespace fz
