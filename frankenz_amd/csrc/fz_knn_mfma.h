// Euclidean k-nearest-neighbour search on the matrix pipe (reference: the K scipy KDTree.query
// calls of knn.py:834-837; SURVEY 7 hard part 5).
//
// With q' = fl32(q - c) for a per-set centre c, the squared distance of a query and a model is one
// 8-slot row-times-column product (F <= 6)
//     |q' - (p - c)|^2 - bar = [-2 q'_0 .. -2 q'_{F-1}, alpha - bar, 1, 0] . [p_0 .. p_{F-1}, 1, beta, j]
//     alpha = |q'|^2 + 2 q'.c      beta = |p - c|^2            (both formed in fp64, rounded once)
// so a wave forms the 16 x 64 products of 16 queries and 64 models with eight v_mfma_f32_16x16x4_f32, and
// the SIGN of a product says whether the pair is under the query's admission bar.  The model side holds the
// ORIGINAL float features (and, in the slot the query side multiplies by 0, the model's original index), so
// the LDS tile that feeds the matrix pipe also feeds the exact re-check; the centring only lives in alpha /
// beta and keeps the big |q|^2 + |p|^2 - 2 q.p cancellation out of the fp32 chain.
// The product is only the SCREEN: the bar is provably above the exact k-th distance (bound below), and the
// pairs under it recompute their distance in fp64 from the original query and features and enter the
// query's sorted list (LDS, ordered by (distance, original index)) -- the neighbour table is the exact fp64
// top-k, bit for bit, whatever the visiting order (tests/test_hip_knn.py
// test_matrix_pipe_search_is_the_exact_search, tests/test_hip_fullsize.py at M = 1e5, K = 25, k = 20).
// Around that: seeds from feature set 0, models in k-d order (tiles = leaves) and queries grouped by leaf with an outward scan from the
// queries' own place, and tiles / groups of tiles skipped by bounding box (comments at k_knn_mfma).
//
// Bar.  u = 2^-24, Q = |q'|, P = max_p |p - c|, C = |c|.  Without rounding the product is
// |q' - (p - c)|^2 - bar + e_a (alpha - bar) + e_b beta (|e| <= u), and | |q' - (p - c)| - |q - p| | <= u Q; the
// 8-term fma chain of the matrix pipe adds at most 8 u sum |a_k b_k| <= 8 u (2 Q (P + C) + |alpha| + bar + beta)
// (16 u is budgeted, in case the pipe truncates).  Hence  |q - p|^2 <= tau  implies  fl(product) < 0  for
//     bar = (tau + 2 sqrt(tau) u Q + u (19 Q^2 + 18 P^2 + 70 Q C + 32 Q P)) (1 + 2e-6),
// slack terms times 1.01, every float operation rounded up (knn_bar_mfma).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fz {

typedef float fz_f4 __attribute__((ext_vector_type(4)));

#define FZ_KM_TILE 64                   // models per LDS tile
#define FZ_KM_TFLOATS (FZ_KM_TILE * 8)  // 8 slots per model
#define FZ_KM_TSTR 544                  // floats per 64-model tile in HBM and LDS: 512 operand floats + bounding box (8 lo, 8 hi) + 16 spare (the stride keeps the LDS-DMA chunking: 2176 B)

// per-feature mean of one feature set (one block per set), in fp64
static __global__ __launch_bounds__(256) void k_knn_center(const float* __restrict__ in, int64_t M, int F, float* __restrict__ cen) {
    const int t = blockIdx.x;
    __shared__ double sh[256];
    for (int f = 0; f < 8; ++f) {
        double s = 0.0;
        if (f < F) for (int64_t j = threadIdx.x; j < M; j += 256) s += (double)in[((size_t)t * M + j) * F + f];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
        if (threadIdx.x == 0) cen[t * 8 + f] = (f < F) ? (float)(sh[0] / (double)M) : 0.f;
        __syncthreads();
    }
}

// B operands: [set][64-model block][half kb][lane][group g] floats -- lane l of the wave reads, for the
// four 16-model groups of a 64-model step, slot kb*4 + (l >> 4) of model 16 g + (l & 15) as ONE 16-byte
// LDS read.  Slots: p_f (f < F, the original float), 1 (slot F), beta (slot F+1), 0.  Pad models: beta = 1e30.
// perm (may be null): position j of set t holds model perm[t][j] (the set's models in k-d order, fz_knn_host.inc);
// with F <= 5 slot 7 is free and carries the model's ORIGINAL index as a float (exact below 2^24; its A-side
// factor is 0), so that the admission path gets it from the tile.
static __global__ __launch_bounds__(256) void k_knn_pack_mfma(const float* __restrict__ in, int64_t M, int F, int64_t Mp,
                                                              const float* __restrict__ cen, float* __restrict__ bmat,
                                                              unsigned* __restrict__ pmax_bits, const int* __restrict__ perm) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    if (j >= Mp) return;
    float slot[8];
    for (int f = 0; f < 8; ++f) slot[f] = 0.f;
    if (j < M) {
        const int64_t oj = perm ? (int64_t)perm[(size_t)t * M + j] : j;
        if (F <= 5) slot[7] = (float)oj;
        double n2 = 0.0;
        for (int f = 0; f < F; ++f) {
            slot[f] = in[((size_t)t * M + oj) * F + f];
            const double pc = (double)slot[f] - (double)cen[t * 8 + f];
            n2 = fma(pc, pc, n2);
        }
        slot[F] = 1.f; slot[F + 1] = (float)n2;
        atomicMax(&pmax_bits[t], __float_as_uint((float)(sqrt(n2) * 1.000001)));     // positive floats order like their bits
    } else {
        slot[F] = 1.f; slot[F + 1] = 1e30f;
    }
    const int64_t blk = j >> 6; const int m = (int)(j & 63), g = m >> 4, col = m & 15;
    float* o = bmat + ((size_t)t * (Mp >> 6) + blk) * FZ_KM_TSTR;
    for (int s = 0; s < 8; ++s) o[(s >> 2) * 256 + ((s & 3) * 16 + col) * 4 + g] = slot[s];
}


// bounding box of every 64-model tile (behind the tile's operands): the search skips a tile when the box is farther from
// each of the wave's 16 queries than that query's admission bar.  Stored slightly enlarged (one float rounding), so that
// the fp32 lower bound formed from it stays below the exact distance.  enable = 0: boxes that never exclude anything.
static __global__ __launch_bounds__(64) void k_knn_boxes(const float* __restrict__ in, int64_t M, int F, int64_t Mp,
                                                         const int* __restrict__ perm, float* __restrict__ bmat, int enable) {
    const int lane = threadIdx.x, t = blockIdx.y;
    const int64_t blk = blockIdx.x, j = blk * 64 + lane;
    float* o = bmat + ((size_t)t * (Mp >> 6) + blk) * FZ_KM_TSTR + 512;
    const int64_t oj = (j < M) ? (perm ? (int64_t)perm[(size_t)t * M + j] : j) : 0;
    for (int f = 0; f < 8; ++f) {
        float lo = INFINITY, hi = -INFINITY;
        if (f < F && j < M) { const float v = in[((size_t)t * M + oj) * F + f]; if (v == v) { lo = v; hi = v; } }
        for (int d = 32; d > 0; d >>= 1) { lo = fminf(lo, __shfl_xor(lo, d, 64)); hi = fmaxf(hi, __shfl_xor(hi, d, 64)); }
        if (lane == 0) {
            const bool on = enable && f < F;
            o[f] = on ? lo - 1.2e-7f * fabsf(lo) : -INFINITY;
            o[8 + f] = on ? hi + 1.2e-7f * fabsf(hi) : INFINITY;
        }
    }
}

// second level of the skipping: boxes of at most 128 groups of 2^gsl consecutive tiles (k-d order: consecutive leaves are subtrees or
// neighbouring subtrees), [set][128][lo 8 | hi 8].  A wave tests them all at once against its 16 queries' bars (k_knn_mfma: two
// lane-parallel rounds) and keeps the result as a 128-bit mask in scalar registers; the scan then steps over unreachable groups
// without touching their tiles' boxes.  Groups past the last tile get an empty box (never reachable).
static __global__ __launch_bounds__(64) void k_knn_maskboxes(int64_t Mp, int M, int gsl, const float* __restrict__ bmat, float* __restrict__ gbox) {
    const int lane = threadIdx.x, t = blockIdx.y, g = blockIdx.x;
    const int64_t ntl = ((int64_t)M + 63) >> 6, t0 = (int64_t)g << gsl, t1 = t0 + ((int64_t)1 << gsl) < ntl ? t0 + ((int64_t)1 << gsl) : ntl;
    if (lane >= 16) return;
    const float* base = bmat + (size_t)t * (Mp >> 6) * FZ_KM_TSTR;
    float v = lane < 8 ? INFINITY : -INFINITY;
    for (int64_t u = t0; u < t1; ++u) {
        const float b = base[u * FZ_KM_TSTR + 512 + lane];
        v = lane < 8 ? fminf(v, b) : fmaxf(v, b);
    }
    gbox[((size_t)t * 128 + g) * 16 + lane] = v;          // (no tile: lo = +inf, hi = -inf)
}

// leaf (= 64-model tile) of a point in a set's implicit k-d tree (k-d order of the models, fz_knn_host.inc): the node over tiles [a, b)
// splits at mid = a + (b - a) / 2 along feature sp[2 mid] at value sp[2 mid + 1] (float bits)
__device__ inline int knn_kd_leaf(const double* v, const int* __restrict__ sp, int ntiles) {
    int a = 0, b = ntiles;
    while (b - a > 1) {
        const int mid = a + (b - a) / 2;
        const int d = sp[2 * mid];
        const float sv = __int_as_float(sp[2 * mid + 1]);
        if ((float)v[d] < sv) b = mid; else a = mid;
    }
    return a;
}
// bucket of a query for the counting sort: its leaf in set 0's k-d tree (scaled into 4096 buckets)
__device__ inline unsigned knn_qbucket(const double* v, const int* __restrict__ sp, int kdn) {
    int sh = 0;
    while ((kdn >> sh) > 4096) ++sh;
    return (unsigned)(knn_kd_leaf(v, sp, kdn) >> sh);
}
// counting sort of the queries by prefix (the order inside a bucket is whatever the atomics give: every query's
// result is independent of the wave that computes it)
static __global__ void k_knn_qhist(const double* __restrict__ q, int64_t N, int F, int* __restrict__ cnt,
                                   const int* __restrict__ sp, int kdn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) atomicAdd(&cnt[knn_qbucket(q + i * F, sp, kdn)], 1);
}
static __global__ __launch_bounds__(1024) void k_knn_qscan(int* __restrict__ cnt) {       // exclusive scan of 4096 counts, one block
    __shared__ int part[1024];
    const int t = threadIdx.x;
    int v[4], s = 0;
    for (int u = 0; u < 4; ++u) { v[u] = cnt[4 * t + u]; s += v[u]; }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const int add = t >= d ? part[t - d] : 0; __syncthreads(); part[t] += add; __syncthreads(); }
    int base = part[t] - s;
    for (int u = 0; u < 4; ++u) { cnt[4 * t + u] = base; base += v[u]; }
}
static __global__ void k_knn_qscatter(const double* __restrict__ q, int64_t N, int F, int* __restrict__ off,
                                      int* __restrict__ qperm, const int* __restrict__ sp, int kdn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) qperm[atomicAdd(&off[knn_qbucket(q + i * F, sp, kdn)], 1)] = (int)i;
}

__device__ __forceinline__ double readlane_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float f32_up(float x) { return __uint_as_float(__float_as_uint(x) + 1u); }     // x > 0 finite
// bar of a row: tau (fp64 k-th distance or the squared bound), uq2 = 2 u Q, e = u (19 Q^2 + ...) of the header
__device__ __forceinline__ float knn_bar_mfma(double tau, float uq2, float e) {
    if (!(tau < 1e37)) return INFINITY;
    float t = (float)tau;
    if ((double)t < tau) t = f32_up(t);
    const float slack = 1.01f * fmaf(uq2, 1.00001f * __builtin_amdgcn_sqrtf(t), e);      // raw v_sqrt_f32: 1 ulp, covered by the factor
    return f32_up(f32_up(fmaf(t, 1.000002f, slack)));                                     // the 17 u bar term of the header
}

// 4 waves x 16 queries per block, one feature set per blockIdx.y (+ tree0); the block streams the set's
// models through two LDS tiles (separate arrays: the LDS-DMA of the next tile does not stall the reads
// of the current one).
//  * The bar of a query row is folded INTO the product (slot F holds alpha - bar), so the screen of a
//    64-model step is the sign of 16 accumulators: 8 v_or3 and one compare next to the 8 MFMAs.
//  * The sorted top-k lists of the 16 queries of a wave live in LDS (dynamic: [wave][16 rows][kpad] fp64
//    distances, then the int32 indices), ordered by (distance, model index) -- the order an ascending
//    scan with first-come ties leaves, whatever the arrival order -- so the admission path is one rolled
//    loop over the pairs that passed the screen.
//  * seed (may be null): the neighbour table of feature set 0, already complete.  The K feature sets
//    are noise realisations of the SAME models, so the k neighbours found in set 0 are k distinct
//    models that are close in this set too: their exact distances start the list, and the bar starts
//    at their largest instead of +inf -- the k ln(M / k) admissions of a cold scan (half of them in
//    the first 1 % of the models) shrink to the few models that really sit inside that ball.
// FX: the feature count when it is a compile-time constant (5: the usual five bands), 0 = runtime F.
// NWB: waves per block (they share the LDS tiles and meet at one barrier per tile; the default launch is ONE wave per block).
// Scan state of one wave: the visiting order (outwards from the queries' own place, alternating sides), the batch of four
// candidate tiles whose bounding boxes are in registers, and the reachable tiles of the last tested batch that have not been
// taken yet.  (A plain struct with inlined members: lambdas capturing lambdas made the compiler keep this state in scratch.)
template <int FL>
struct KnnScan {
    int nl, nr, ntiles, right;
    unsigned long long m0, m1;                             // reachable groups of 2^gsl tiles (bit g; wave-uniform), see k_knn_maskboxes
    int gsl;
    int cand;                                             // this lane's candidate tile of the batch in registers: lane group c = lane >> 4 holds candidate c (-1: none)
    int pend;                                             // ... of the batch that was tested last
    fz_f4 bl0, bl1, bh0, bh1;                             // the candidate's box: lo[0..7], hi[0..7]
    unsigned pmask;                                       // reachable candidates of the tested batch that have not been taken yet
    // (no four-way selects over members here: the compiler turns them into indexed loads and moves the whole struct to scratch)
    __device__ __forceinline__ int next_tile() {
        while (true) {
            if (nr >= ntiles && nl < 0) return -1;
            const bool r = (right && nr < ntiles) || nl < 0;
            const int t = r ? nr : nl;
            const int g = t >> gsl;                         // < 128
            const unsigned long long mw = g < 64 ? m0 : m1;
            if ((mw >> (g & 63)) & 1ull) {
                if (r) ++nr; else --nl;
                right = 1 - right;
                return t;
            }
            if (r) nr = (g + 1) << gsl; else nl = (g << gsl) - 1;       // the whole group is out of every row's reach
        }
    }
    __device__ __forceinline__ void fetch(const float* bm, int sl) {
        int m = -1;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int t = next_tile(); m = (sl == c) ? t : m; }
        cand = m;
        const fz_f4* bx = reinterpret_cast<const fz_f4*>(bm + (size_t)(m >= 0 ? m : 0) * FZ_KM_TSTR + 512);
        bl0 = bx[0]; bl1 = bx[1]; bh0 = bx[2]; bh1 = bx[3];
    }
    // bit c: candidate c is within reach of some row (lane group c tests it: its 16 lanes are the 16 query rows)
    __device__ __forceinline__ unsigned test(const float (&qlo)[FL], const float (&qhi)[FL], float barrow) const {
        float lb = 0.f;
#pragma unroll
        for (int f = 0; f < FL; ++f) {
            const float lo = f < 4 ? bl0[f & 3] : bl1[f & 3], hi = f < 4 ? bh0[f & 3] : bh1[f & 3];
            const float m = fmaxf(fmaxf(lo - qhi[f], qlo[f] - hi), 0.f);        // the query's rounding already in qlo / qhi
            lb = fmaf(m, m, lb);
        }
        const unsigned long long rm = __ballot(cand >= 0 && lb * 0.999999f <= barrow);
        return ((rm & 0xffffull) ? 1u : 0u) | (((rm >> 16) & 0xffffull) ? 2u : 0u) | (((rm >> 32) & 0xffffull) ? 4u : 0u) | ((rm >> 48) ? 8u : 0u);
    }
    __device__ __forceinline__ int next_reachable(const float* bm, int sl, const float (&qlo)[FL], const float (&qhi)[FL], float barrow) {
        while (true) {
            if (pmask) {
                const int c = __builtin_ctz(pmask);
                pmask &= pmask - 1u;
                return __builtin_amdgcn_readlane(pend, 16 * c);
            }
            if (__builtin_amdgcn_readfirstlane(cand) < 0) return -1;      // no batch left (candidate 0 is the first to run out)
            pend = cand;
            pmask = test(qlo, qhi, barrow);
            fetch(bm, sl);                                 // the next batch's boxes travel while this batch's tiles are processed
        }
    }
};

// KPL > 0: ROW-PARALLEL ADMISSIONS.  The sorted list of query row R lives in the registers of the four lanes 4R .. 4R+3
// (KPL consecutive entries each, k <= 4 KPL), so the wave holds its 16 lists as 16 independent "teams".  Pairs that pass
// the exact fp64 re-check are not inserted one at a time by the whole wave (a serial LDS round trip + ~150 instructions
// each: 3/4 of the kernel's instructions at M = 1e5) but appended to their row's small LDS queue; when a queue holds
// FZ_KM_DT entries -- or the scan ends -- every team drains its own queue at the same time, one entry per round: compare
// against its KPL entries, find the place with two quad-DPP moves, shift.  Rows whose queue is empty sit the round out.
// Bars and k-th distances are refreshed once per drain, so between drains the screen is looser than it could be (a
// superset is admitted; an entry that no longer belongs is a no-op for its team).  The lists are ordered by
// (distance, original index) as before: the neighbour table does not depend on any of this.  KPL = 0: the lists stay in LDS
// and the wave inserts one candidate at a time (k > 32).
#ifndef FZ_KM_QC
#define FZ_KM_QC 24                     // queue entries per row: FZ_KM_DT - 1 may wait when a step begins and a step adds at most 16 to one row
#define FZ_KM_DT 8                      // drain when some row's queue holds this many
#endif
template <int TILE, int FX, int NWB, int KPL>
static __global__ __launch_bounds__(NWB * 64) void k_knn_mfma(const float* __restrict__ bmat, const float* __restrict__ cen,
                                                         const float* __restrict__ pmax, const float* __restrict__ feats, int FT,
                                                         int64_t Mp, int M, const double* __restrict__ q, int64_t N, int F, int k,
                                                         int kpad, double bound2, int64_t* idx, int K, int tree0, const int64_t* seed,
                                                         const int* __restrict__ qperm, const int* __restrict__ ktab,
                                                         const float* __restrict__ gbox, int gsl) {
    static_assert(TILE == 64 && NWB == 1, "one wave per block, one 64-model tile (+ its bounding box) per step");
    static_assert(KPL == 0 || KPL == 5 || KPL == 8, "register lists: k <= 20 or k <= 32");
    constexpr int TF = FZ_KM_TSTR;
    constexpr int FL = FX ? FX : 6;                    // feature loop bound of the exact re-check
    __shared__ __attribute__((aligned(16))) float tA[TF];
    __shared__ __attribute__((aligned(16))) float tB[TF];
    __shared__ double qs[NWB][16][8];                   // the queries in fp64 (exact re-check); slot 6 = tau, slot 7 = bar constants
    __shared__ int qn[16];                              // KPL: entries waiting in each row's queue
    extern __shared__ double s_lists[];
    // KPL: the per-row queues of admitted (distance, model) pairs reuse the lists' LDS -- the lists are only there while the
    // seeds are ranked and when the result is written; in between they live in registers
    double* qd = s_lists;                                // [16][FZ_KM_QC]
    int* qj = reinterpret_cast<int*>(s_lists + 16 * FZ_KM_QC);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tree = blockIdx.y + tree0;
    const int64_t i0 = ((int64_t)blockIdx.x * NWB + wave) * 16;
    const float* bm = bmat + (size_t)tree * (Mp >> 6) * FZ_KM_TSTR;
    const int row = lane & 15, sl = lane >> 4;
    double* Ld = s_lists + (size_t)wave * 16 * kpad;                                     // [16][kpad]
    int* Lj = reinterpret_cast<int*>(s_lists + (size_t)NWB * 16 * kpad) + (size_t)wave * 16 * kpad;
    for (int e = lane; e < 16 * kpad; e += 64) { Ld[e] = INFINITY; Lj[e] = M + e % kpad; }
    if (lane < 16) qn[lane] = 0;
    // ---- the wave's 16 queries: fp64 copies in LDS, A operands in registers ----
    // qperm (may be null): the queries grouped by their leaf in set 0's tree -- the 16 queries of a wave are neighbours in feature space
    const int64_t islot = i0 + row < N ? i0 + row : N - 1;
    const int64_t qi = qperm ? (int64_t)qperm[islot] : islot;       // this lane's row, as an index into q / idx
    for (int ff = sl; ff < 6; ff += 4) qs[wave][row][ff] = (ff < F) ? q[qi * F + ff] : 0.0;
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): own LDS writes done (the same wave reads them)
    float qc[6]; double qn2 = 0.0, qdc = 0.0, c2 = 0.0;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const double cf = (f < F) ? (double)cen[tree * 8 + f] : 0.0;
        qc[f] = (f < F) ? (float)(qs[wave][row][f] - cf) : 0.f;
        qn2 = fma((double)qc[f], (double)qc[f], qn2); qdc = fma((double)qc[f], cf, qdc); c2 = fma(cf, cf, c2);
    }
    const float alpha = (float)(qn2 + 2.0 * qdc);
    {
        const double u = 6.0e-8, Q = sqrt(qn2) * 1.000001, P = (double)pmax[tree], C = sqrt(c2) * 1.000001;
        const float uq2 = (float)(2.0 * u * Q * 1.000001);
        const float e = (float)(u * (19.0 * Q * Q + 18.0 * P * P + 70.0 * Q * C + 32.0 * Q * P) * 1.000001);
        if (sl == 0) { float2 pk; pk.x = uq2; pk.y = e; *reinterpret_cast<float2*>(&qs[wave][row][7]) = pk; qs[wave][row][6] = bound2; }
    }
    // ---- seeds: exact distances of set 0's neighbours in THIS set, ranked into the row's list ----
    if (seed) {
        const float* ft = feats + (size_t)tree * FT * Mp;
        for (int R = 0; R < 16; ++R) {
            const int64_t i = __shfl(qi, R, 64);
            const int64_t js64 = (lane < k) ? seed[(i * K) * k + lane] : (int64_t)M;
            const bool valid = js64 >= 0 && js64 < M;
            int js = valid ? (int)js64 : M + lane;                          // distinct keys for the empty entries
            double d2 = INFINITY;
            if (valid) {
                d2 = 0.0;
                for (int f = 0; f < F; ++f) { const double d = qs[wave][R][f] - (double)ft[(size_t)f * Mp + js]; d2 = fma(d, d, d2); }
                if (!(d2 < bound2)) { d2 = INFINITY; js = M + lane; }
            }
            double* ldr = Ld + R * kpad; int* ljr = Lj + R * kpad;
            if (lane < k) { ldr[lane] = d2; ljr[lane] = js; }
            int rank = 0;
            for (int m = 0; m < k; ++m) { const double dm = ldr[m]; const int jm = ljr[m]; rank += (dm < d2 || (dm == d2 && jm < js)) ? 1 : 0; }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane < k) { ldr[rank] = d2; ljr[rank] = js; }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane < 16) { const double kd = Ld[lane * kpad + k - 1]; qs[wave][lane][6] = kd < bound2 ? kd : bound2; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    // A operands: slots -2 q'_f | alpha - bar | 1 | 0; the bar slot F sits in a0 (F < 4) or a1 of the lanes with sl == F & 3
    auto aslot = [&](int s, float ab) -> float {
        float v = (s == F) ? ab : ((s == F + 1) ? 1.f : 0.f);
#pragma unroll
        for (int f = 0; f < 6; ++f) if (s == f && f < F) v = -2.f * qc[f];
        return v;
    };
    float a0, a1;
    {
        const float2 pk = *reinterpret_cast<const float2*>(&qs[wave][row][7]);
        const float ab = alpha - knn_bar_mfma(qs[wave][row][6], pk.x, pk.y);
        a0 = aslot(sl, ab); a1 = aslot(4 + sl, ab);
    }
    const bool bar_lane = sl == (F & 3);
    // for the tile test: this lane's row in fp32 (+ the rounding that cost), and the row's bar in distance units
    float qlo[FL], qhi[FL];
#pragma unroll
    for (int f = 0; f < FL; ++f) { const float qf = (float)qs[wave][row][f], eq = 2.0e-7f * fabsf(qf); qlo[f] = qf - eq; qhi[f] = qf + eq; }      // the conversion's rounding and that of qf -+ eq, with room
    float barrow;
    {
        const float2 pk = *reinterpret_cast<const float2*>(&qs[wave][row][7]);
        barrow = knn_bar_mfma(qs[wave][row][6], pk.x, pk.y);
    }

    // ---- row teams: the lists in registers, the drain ----
    constexpr int KP = KPL ? KPL : 1;
    double Lr[KP]; int Jr[KP];
    const int team = lane >> 2, tl = lane & 3;
    if constexpr (KPL > 0) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int e = 0; e < KPL; ++e) { Lr[e] = Ld[team * kpad + tl * KPL + e]; Jr[e] = Lj[team * kpad + tl * KPL + e]; }   // kpad >= 4 KPL
    }
    auto quad_prev_i = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_update_dpp(v, v, 0x90, 0xf, 0xf, false); };     // quad_perm [0,0,1,2]: lane t <- lane t - 1
    auto quad_or = [](int v) __attribute__((always_inline)) {
        v |= __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);       // [1,0,3,2]
        v |= __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);       // [2,3,0,1]
        return v;
    };
    auto refresh_bars = [&]() __attribute__((always_inline)) {                          // every lane: the bar of its row from the row's k-th distance
        const float2 pk = *reinterpret_cast<const float2*>(&qs[wave][row][7]);
        barrow = knn_bar_mfma(qs[wave][row][6], pk.x, pk.y);
        const float ab = alpha - barrow;
        if (bar_lane) { if (F < 4) a0 = ab; else a1 = ab; }
    };
    auto drain = [&]() __attribute__((always_inline)) {
        if constexpr (KPL > 0) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const int cnt = qn[team];
            for (int h = 0; __any(h < cnt); ++h) {
                const bool act = h < cnt;
                const double dn = qd[team * FZ_KM_QC + (act ? h : 0)];
                const int jn = qj[team * FZ_KM_QC + (act ? h : 0)];
                int cb = 0, dup = 0;
#pragma unroll
                for (int e = 0; e < KPL; ++e) {
                    cb += (Lr[e] < dn || (Lr[e] == dn && Jr[e] < jn)) ? 1 : 0;      // lexicographic (distance, index)
                    dup |= (Jr[e] == jn) ? 1 : 0;                                   // a model already listed (a seed) is not listed twice
                }
                dup = quad_or(dup);
                const int cbp = quad_prev_i(cb);
                const int lh = quad_prev_i(__double2hiint(Lr[KPL - 1])), ll = quad_prev_i(__double2loint(Lr[KPL - 1]));
                const int jl = quad_prev_i(Jr[KPL - 1]);
                const bool first = (tl == 0) || (cbp == KPL);                      // the entry lands in this lane (else: the lower lane's last one moves up)
                const double nd = first ? dn : __hiloint2double(lh, ll);
                const int nj = first ? jn : jl;
                if (act && !dup && cb < KPL) {
#pragma unroll
                    for (int e = KPL - 1; e >= 0; --e) {
                        const bool sh = e > cb, at = e == cb;
                        Lr[e] = sh ? Lr[e > 0 ? e - 1 : 0] : (at ? nd : Lr[e]);
                        Jr[e] = sh ? Jr[e > 0 ? e - 1 : 0] : (at ? nj : Jr[e]);
                    }
                }
            }
            // the k-th distance of every row, then the bars
            // (one predicated store per slot: a select chain over the slots becomes an indexed load and the list goes to scratch)
            const int ke = k - 1;
#pragma unroll
            for (int e = 0; e < KPL; ++e)
                if (tl * KPL + e == ke) qs[wave][team][6] = Lr[e] < bound2 ? Lr[e] : bound2;
            if (tl == 0) qn[team] = 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            refresh_bars();
        }
    };

    const int ntiles = (M + TILE - 1) / TILE;
    // Visiting order.  The set's models are stored in k-d order (upload: depth-first leaves) and the wave's queries are neighbours
    // (qperm), so the tiles around the queries' own place hold most of their neighbours: start there and work outwards,
    // alternating sides.  The bar then drops to nearly its final value within the first few per cent of the models and
    // the rest of the scan admits little (benchmark data: 55-65 admissions per query instead of 87).  The lists are
    // ordered by (distance, original index), so the result does not depend on the order.  One wave per block only:
    // waves sharing tiles would need a common start.
    // Tile skipping.  Every 64-model tile carries its bounding box (behind its operands in HBM).  The boxes of the next FOUR
    // tiles of the visiting order are tested at once -- lane group c = lane >> 4 takes candidate c, its 16 lanes the 16 query
    // rows: lower bound of the distance from the row's query to the box against the row's bar -- straight from global memory
    // (64 B per tile, L2-resident), and only tiles within reach of some row are staged into LDS and multiplied.  The boxes of
    // the following batch are requested before the current batch's tiles are processed.  (Before: every tile of a
    // reachable group of eight was staged, 2 KB, to read its box from LDS: 59 % of all tiles staged for 18 % multiplied, and
    // the per-tile test + staging + barrier was two thirds of the kernel's instructions.)
    KnnScan<FL> sc;
    sc.nl = -1; sc.nr = 0; sc.ntiles = ntiles; sc.right = 1; sc.pmask = 0u; sc.pend = -1;
    sc.m0 = ~0ull; sc.m1 = ~0ull; sc.gsl = gbox ? gsl : 30;
    // group mask: lane l tests groups l and l + 64 against all 16 rows (their queries from LDS, their bars from lanes 0..15)
    auto build_mask = [&]() __attribute__((always_inline)) {
        if (!gbox) return;
        unsigned long long mm[2];
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const fz_f4* bx = reinterpret_cast<const fz_f4*>(gbox + ((size_t)tree * 128 + lane + 64 * h) * 16);
            const fz_f4 l0 = bx[0], l1 = bx[1], h0 = bx[2], h1 = bx[3];
            bool reach = false;
#pragma unroll 1
            for (int R = 0; R < 16; ++R) {
                const float br = __shfl(barrow, R, 64);
                float lb = 0.f;
#pragma unroll
                for (int f = 0; f < FL; ++f) {
                    const float qf = (float)qs[wave][R][f], eq = 2.0e-7f * fabsf(qf);
                    const float lo = f < 4 ? l0[f & 3] : l1[f & 3], hi = f < 4 ? h0[f & 3] : h1[f & 3];
                    const float m = fmaxf(fmaxf(lo - (qf + eq), (qf - eq) - hi), 0.f);
                    lb = fmaf(m, m, lb);
                }
                reach = reach || (lb * 0.999999f <= br);
            }
            mm[h] = __ballot(reach);
        }
        sc.m0 = mm[0]; sc.m1 = mm[1];
    };
    bool need_mask = seed != nullptr;                    // seeded sets: the bars are near their final values from the start
    if (NWB == 1 && ktab) {
        // the leaf of the wave's middle query in THIS set's tree (11 dependent pairs of scalar loads at M = 1e5; the query straight from
        // LDS: a dynamic index into registers would go to scratch)
        const int hm = knn_kd_leaf(&qs[wave][8][0], ktab + (size_t)tree * ntiles * 2, ntiles);
        sc.nr = __builtin_amdgcn_readfirstlane(hm < ntiles ? hm : ntiles - 1); sc.nl = sc.nr - 1;
    }
    sc.fetch(bm, sl);
    auto stage = [&](int tile, float* dst) __attribute__((always_inline)) {             // 2176 contiguous bytes: 136 16-byte chunks over the wave's 64 lanes
        const char* src = reinterpret_cast<const char*>(bm + (size_t)tile * TF);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < 2 || tid < 8)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * 1024 + (uint32_t)tid * 16u),
                                                 (__attribute__((address_space(3))) void*)(dst + c * 256), 16, 0, 0);
    };
    // admission path.  pm: this lane's 16-bit mask of (g, r) products under the bar.
    auto slow = [&](unsigned pm, const float* blk, int jb) __attribute__((always_inline)) {
        while (__any(pm != 0u)) {
            // every lane with something left takes its lowest pair: query row 4 sl + r, model 16 g + col of the step
            const bool has = pm != 0u;
            const int e = has ? __builtin_ctz(pm) : 0;
            pm &= pm - 1u;
            const int g = e >> 2, r = e & 3, R = 4 * sl + r;
            const int jpos = jb + 16 * g + row;                      // position in the (k-d-ordered) set
            double qv[FL]; float pv[FL];                             // exact distance from the original query and features
#pragma unroll
            for (int f = 0; f < FL; ++f) { qv[f] = qs[wave][R][f]; pv[f] = blk[(f >> 2) * 256 + ((f & 3) * 16 + row) * 4 + g]; }
            const int j = (FX ? FX <= 5 : F <= 5) ? (int)blk[256 + (48 + row) * 4 + g] : jpos;       // the model's original index (slot 7)
            const double taur = qs[wave][R][6];
            double d2 = 0.0;
#pragma unroll
            for (int f = 0; f < FL; ++f) { const double d = (FX || f < F) ? qv[f] - (double)pv[f] : 0.0; d2 = fma(d, d, d2); }
            const bool adm = has && jpos < M && d2 <= taur && d2 < bound2;
            if constexpr (KPL > 0) {
                int slot = 0;
                if (adm) {
                    slot = atomicAdd(&qn[R], 1);                   // < FZ_KM_QC: queues are drained from FZ_KM_DT entries on, a step adds <= 16 per row
                    qd[R * FZ_KM_QC + slot] = d2; qj[R * FZ_KM_QC + slot] = j;
                }
                if (__any(adm && slot + 1 >= FZ_KM_DT)) drain();
                continue;
            }
            unsigned long long cm = __ballot(adm);
            while (cm) {                                             // one candidate at a time into its row's sorted list
                const int s1 = __builtin_ctzll(cm);
                cm &= cm - 1;
                const double dn = readlane_d(d2, s1);                // wave-uniform: scalar registers, no LDS round trip
                const int Rn = __builtin_amdgcn_readlane(R, s1), jn = __builtin_amdgcn_readlane(j, s1);
                double* ldr = Ld + Rn * kpad; int* ljr = Lj + Rn * kpad;
                // everything the insertion needs, in one LDS round trip
                double l = ldr[lane < k ? lane : 0];
                int lj = ljr[lane < k ? lane : 0];
                double pk2 = ldr[k > 1 ? k - 2 : 0];
                float2 pk = *reinterpret_cast<const float2*>(&qs[wave][Rn][7]);
                asm volatile("" : "+v"(l), "+v"(lj), "+v"(pk2), "+v"(pk.x), "+v"(pk.y));     // all loads in flight before anything is consumed
                // lexicographic (distance, index) order; a model already listed (a seed) is not listed twice
                const unsigned long long before = __ballot(lane < k && (l < dn || (l == dn && lj < jn)));
                const unsigned long long same = __ballot(lane < k && lj == jn);
                const int pos = __builtin_popcountll(before);
                const bool ok = pos < k && same == 0ull;              // (wave-uniform) pos == k: the row's bar moved while draining
                if (ok && lane >= pos && lane < k - 1) { ldr[lane + 1] = l; ljr[lane + 1] = lj; }
                if (ok && lane == pos) { ldr[lane] = dn; ljr[lane] = jn; }
                if (ok) {
                    const double nk = (pos >= k - 1) ? dn : pk2;     // the new k-th distance
                    const double tau = nk < bound2 ? nk : bound2;
                    const float nbar = knn_bar_mfma(tau, pk.x, pk.y);
                    const float ab = alpha - nbar;
                    if (row == Rn) barrow = nbar;
                    if (lane == 0) qs[wave][Rn][6] = tau;
                    if (bar_lane && row == Rn) { if (F < 4) a0 = ab; else a1 = ab; }
                }
            }
        }
    };
    int cur_t = sc.next_reachable(bm, sl, qlo, qhi, barrow);
    int ndone = 0;
    auto run_tile = [&](const float* cur, float* nxt) __attribute__((always_inline)) -> bool {
        const int nx = sc.next_reachable(bm, sl, qlo, qhi, barrow);     // (tested against the bars as they are now: a superset of what will still matter)
        if (nx >= 0) stage(nx, nxt);
        const int t = cur_t;
        fz_f4 nb0 = *reinterpret_cast<const fz_f4*>(cur + lane * 4);
        fz_f4 nb1 = *reinterpret_cast<const fz_f4*>(cur + 256 + lane * 4);
#pragma unroll 2
        for (int s = 0; s < TILE / 64; ++s) {
            const fz_f4 b0 = nb0, b1 = nb1;
            if (s + 1 < TILE / 64) {
                nb0 = *reinterpret_cast<const fz_f4*>(cur + (s + 1) * 512 + lane * 4);
                nb1 = *reinterpret_cast<const fz_f4*>(cur + (s + 1) * 512 + 256 + lane * 4);
            }
            fz_f4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[g], fz_f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[g], acc[g], 0, 0, 0);
            int sg = 0;                                              // OR of the 16 sign bits: any product under its row's bar
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) sg |= __float_as_int(acc[g][r]);
            if (__any(sg < 0)) {
                unsigned pm = 0u;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pm |= (__float_as_uint(acc[g][r]) >> 31) << (4 * g + r);
                slow(pm, cur + s * 512, t * TILE + s * 64);
            }
        }
        cur_t = nx;
        // the bars have dropped since the mask was built (a cold scan starts with none): again after 8 and after 40 tiles
        ++ndone;
        need_mask = need_mask || ndone == 8 || ndone == 40;
        __syncthreads();
        return nx >= 0;
    };
    if (cur_t >= 0) {
        stage(cur_t, tA);
        __syncthreads();
        while (true) {
            if (need_mask) { build_mask(); need_mask = false; }      // (the one call site: inlined three times it cost the kernel its occupancy)
            if (!run_tile(tA, tB)) break;
            if (!run_tile(tB, tA)) break;
        }
    }
    if constexpr (KPL > 0) {
        drain();
#pragma unroll
        for (int e = 0; e < KPL; ++e) { Ld[team * kpad + tl * KPL + e] = Lr[e]; Lj[team * kpad + tl * KPL + e] = Jr[e]; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    for (int R = 0; R < 16; ++R) {
        const int64_t i = __shfl(qi, R, 64);
        if (i0 + R < N && lane < k) idx[(i * K + tree) * k + lane] = (Ld[R * kpad + lane] < bound2) ? Lj[R * kpad + lane] : M;
    }
}

}  // namespace fz
