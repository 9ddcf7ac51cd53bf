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
// query's sorted list (registers of a 4-lane team, ordered by (distance, original index)) -- the neighbour table is the
// exact fp64 top-k, bit for bit, whatever the visiting order (tests/test_hip_knn.py
// test_matrix_pipe_search_is_the_exact_search, test_reachability_mask_at_every_group_size, tests/test_hip_fullsize.py at M = 1e5,
// K = 25, k = 20).
// Around that: models in k-d order (tiles = leaves) and queries grouped by leaf, an outward scan from the queries' own leaf, that
// leaf entered through its k nearest, and a bit mask of the tiles whose bounding box some query's bar still reaches (comments at
// k_knn_mfma).  DESIGN.md 3.5 has the measurements each of these was chosen by.
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
#ifdef FZ_KM_STATS                      // development counters (tools/devbuild.sh with FZ_DEV_MAINFLAGS=-DFZ_KM_STATS): never in a release build
__device__ unsigned long long fz_kmstats[128];
#define KMSTAT(i, v) do { if ((i) < 8) kmc[(i) & 7] += (unsigned)(v); } while (0)         // per-wave registers, flushed once at the end (global atomics inside the loops would be what is measured)
#define KMFLUSH(i, v) do { const unsigned long long kmv_ = (unsigned long long)(v); if (lane == 0) atomicAdd(&fz_kmstats[i], kmv_); } while (0)
// wall-clock cycles since the last stamp go to section i (8 + i in fz_kmstats)
#define KMT(i) do { const long long kmt_ = (long long)clock64(); kmtime[i] += kmt_ - kmlast; kmlast = kmt_; } while (0)
#else
#define KMSTAT(i, v) do { } while (0)
#define KMT(i) do { } while (0)
#endif
#define FZ_KM_TSTR 512                  // floats per 64-model tile in HBM and LDS: 8 slots x 64 models = two 1 KB LDS-DMA rounds of the wave

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


// bounding box of every 64-model tile, [set][tile][lo 8 | hi 8] (contiguous: the scan tests 64 tiles at a time, one per lane): a tile
// farther from each of the wave's 16 queries than that query's admission bar is never staged.  Stored slightly enlarged (one float
// rounding), so that the fp32 lower bound formed from it stays below the exact distance.
static __global__ __launch_bounds__(64) void k_knn_boxes(const float* __restrict__ in, int64_t M, int F, int64_t Mp,
                                                         const int* __restrict__ perm, float* __restrict__ tbox) {
    const int lane = threadIdx.x, t = blockIdx.y;
    const int64_t blk = blockIdx.x, j = blk * 64 + lane;
    float* o = tbox + ((size_t)t * (Mp >> 6) + blk) * 16;
    const int64_t oj = (j < M) ? (perm ? (int64_t)perm[(size_t)t * M + j] : j) : 0;
    for (int f = 0; f < 8; ++f) {
        float lo = INFINITY, hi = -INFINITY;
        if (f < F && j < M) { const float v = in[((size_t)t * M + oj) * F + f]; if (v == v) { lo = v; hi = v; } }
        for (int d = 32; d > 0; d >>= 1) { lo = fminf(lo, __shfl_xor(lo, d, 64)); hi = fmaxf(hi, __shfl_xor(hi, d, 64)); }
        if (lane == 0) {
            o[f] = f < F ? lo - 1.2e-7f * fabsf(lo) : -INFINITY;
            o[8 + f] = f < F ? hi + 1.2e-7f * fabsf(hi) : INFINITY;
        }
    }
}

// first level of the skipping: boxes of at most 128 groups of 2^gsl consecutive tiles (k-d order: consecutive leaves are subtrees or
// neighbouring subtrees), [set][128][lo 8 | hi 8].  A wave tests them all at once against its 16 queries' bars (k_knn_mfma: two
// lane-parallel rounds) and keeps the result as a 128-bit mask in scalar registers; the tiles of unreachable groups are not even tested.
// Groups past the last tile get an empty box (never reachable).
static __global__ __launch_bounds__(64) void k_knn_maskboxes(int64_t Mp, int M, int gsl, const float* __restrict__ tbox, float* __restrict__ gbox) {
    const int lane = threadIdx.x, t = blockIdx.y, g = blockIdx.x;
    const int64_t ntl = ((int64_t)M + 63) >> 6, t0 = (int64_t)g << gsl, t1 = t0 + ((int64_t)1 << gsl) < ntl ? t0 + ((int64_t)1 << gsl) : ntl;
    if (lane >= 16) return;
    const float* base = tbox + (size_t)t * (Mp >> 6) * 16;
    float v = lane < 8 ? INFINITY : -INFINITY;
    for (int64_t u = t0; u < t1; ++u) {
        const float b = base[u * 16 + lane];
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

// lane exchange inside a row of 16 lanes (DPP): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140 --
// after combining through the four in this order every lane of the row holds the reduction over its 16 lanes
template <int CTRL> __device__ __forceinline__ int row16_xchg(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float row16_xchg(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)); }

// One wave per block = 16 queries against one feature set (blockIdx.y); the wave streams the set's reachable 64-model tiles through
// two LDS buffers (separate arrays: the LDS-DMA of the next tile does not stall the reads of the current one).
//  * The bar of a query row is folded INTO the product (slot F holds alpha - bar), so the screen of a
//    64-model step is the sign of 16 accumulators: 8 v_or3 and one compare next to the 8 MFMAs.
//  * The sorted top-k lists of the 16 queries are ordered by (distance, model index) -- the order an ascending
//    scan with first-come ties leaves, whatever the arrival order -- so the table does not depend on the visiting order.
//  * Every feature set is searched cold.  (Rounds 2-4 started sets 1..K-1 from the exact distances of set 0's neighbours: the
//    Monte-Carlo noise of the benchmark displaces them so far -- their k-th distance is 13x the true one -- that ranking them cost
//    more than they saved once the first tile's admissions were spread over all row teams: 23.9 -> 22.6 ms per step without them.)
// FX: the feature count when it is a compile-time constant (5: the usual five bands), 0 = runtime F.
//
// Scan state of one wave: the visiting order (outwards from the queries' own leaf, alternating sides) over the tiles whose bit is set
// in the wave's reachability mask -- one bit per tile in LDS (tm), built by build_mask from the tiles' bounding boxes, under a
// 128-bit mask of reachable tile GROUPS in scalar registers.  (A plain struct with inlined members: lambdas capturing lambdas made
// the compiler keep this state in scratch.)
struct KnnScan {
    int nl, nr, ntiles, right;
    unsigned long long m0, m1;                             // reachable groups of 2^gsl tiles (bit g; wave-uniform), see k_knn_maskboxes
    int gsl;
    const unsigned long long* tm;                          // LDS: bit t & 63 of word t >> 6 = tile t may hold a model under some row's bar
    // (the word of a scan head is read from LDS at every step: keeping the two current words in scalar registers measured 15.85 against 15.45 ms)
    __device__ __forceinline__ int next_tile() {
        while (true) {
            if (nr >= ntiles && nl < 0) return -1;
            const bool r = (right && nr < ntiles) || nl < 0;
            const int t = r ? nr : nl;
            const int g = t >> gsl;                         // < 128
            const unsigned long long mw = g < 64 ? m0 : m1;
            if (!((mw >> (g & 63)) & 1ull)) {               // the whole group is out of every row's reach
                if (r) nr = (g + 1) << gsl; else nl = (g << gsl) - 1;
                continue;
            }
            const unsigned long long wv = tm[t >> 6];       // (every lane reads the same word)
            const unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wv >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)wv);
            if (r) {
                const unsigned long long rest = w >> (t & 63);
                if (rest) { const int tt = t + __builtin_ctzll(rest); nr = tt + 1; right = 0; return tt; }
                nr = ((t >> 6) + 1) << 6;
            } else {
                const unsigned long long rest = w << (63 - (t & 63));
                if (rest) { const int tt = t - __builtin_clzll(rest); nl = tt - 1; right = 1; return tt; }
                nl = ((t >> 6) << 6) - 1;
            }
        }
    }
};

// KPL > 0: ROW-PARALLEL ADMISSIONS.  The sorted list of query row R lives in the registers of the four lanes 4R .. 4R+3
// (KPL consecutive entries each, k <= 4 KPL), so the wave holds its 16 lists as 16 independent "teams".  Pairs that pass
// the exact fp64 re-check are not inserted one at a time by the whole wave (a serial LDS round trip + ~150 instructions
// each) but appended to their row's small LDS queue; when a queue holds FZ_KM_DT entries -- or the scan ends -- every team
// drains its own queue at the same time, one entry per round: compare against its KPL entries, find the place with two
// quad-DPP moves, shift.  Rows whose queue is empty sit the round out.
// Bars and k-th distances are refreshed once per drain, so between drains the screen is looser than it could be (a
// superset is admitted; an entry that no longer belongs is a no-op for its team).  The lists are ordered by
// (distance, original index) as before: the neighbour table does not depend on any of this.  KPL = 0: the lists stay in LDS
// and the wave inserts one candidate at a time (k > 32).
#ifndef FZ_KM_QC
#define FZ_KM_QC 12                     // queue entries per row (a pair that finds its row's queue full waits for the drain it triggers)
#define FZ_KM_DT 12                     // drain when some row's queue holds this many (QC / DT 24 / 8: 18.6 ms per step, 12 / 8: 16.6 -- LDS bounds the occupancy --, 12 / 10: 16.4, 12 / 12: 16.2, 16 / 12: 16.5, 8 / 6: 16.8)
#endif
#ifndef FZ_KM_B1
#define FZ_KM_B1 28                     // the reachability mask is built after this many tiles (the scan starts without one: the own leaf and its
#define FZ_KM_B2 1000000                // neighbours set the bars; building costs as much as ~15 tiles, so once, late: 6 + 40: 19.7 ms per step, 14: 18.4, 20: 18.1, 28: 17.9, 40: 17.9) ... and again after this many
#endif
// dynamic LDS: the lists / queues, then one bit per tile (fz_knn_host.inc sizes the launch with the same expression)
__host__ __device__ inline size_t knn_mfma_list_bytes(int kpad, int kpl) {
    // register lists (kpl > 0): only the rows' admission queues live in LDS; else the sorted lists themselves
    return ((kpl ? (size_t)16 * FZ_KM_QC * 12 : (size_t)16 * kpad * 12) + 15) & ~(size_t)15;
}
template <int TILE, int FX, int NWB, int KPL>
static __global__ __launch_bounds__(NWB * 64) void k_knn_mfma(const float* __restrict__ bmat, const float* __restrict__ cen,
                                                         const float* __restrict__ pmax, int64_t Mp, int M,
                                                         const double* __restrict__ q, int64_t N, int F, int k,
                                                         int kpad, double bound2, int64_t* idx, int K,
                                                         const int* __restrict__ qperm, const int* __restrict__ ktab,
                                                         const float* __restrict__ tbox, const float* __restrict__ gbox, int gsl) {
    static_assert(TILE == 64 && NWB == 1, "one wave per block, one 64-model tile per step");
    static_assert(KPL == 0 || KPL == 5 || KPL == 8, "register lists: k <= 20 or k <= 32");
    constexpr int TF = FZ_KM_TSTR;
    constexpr int FL = FX ? FX : 6;                    // feature loop bound of the exact re-check
    __shared__ __attribute__((aligned(16))) float tA[TF];       // the tile being multiplied and the one in flight (separate arrays and
    __shared__ __attribute__((aligned(16))) float tB[TF];       // two copies of the tile step: one copy with a toggled buffer index measured 18.6 against 17.8 ms)
    __shared__ double qs[16][8];                        // the queries in fp64 (exact re-check); slot 6 = tau, slot 7 = bar constants
    __shared__ __attribute__((aligned(16))) float qbx[16][12];   // the queries in fp32 with their rounding: lo[6] | hi[6] (tile tests)
    __shared__ int qn[16];                              // KPL: entries waiting in each row's queue
    extern __shared__ double s_lists[];
    // KPL: the per-row queues of admitted (distance, model) pairs take the place of the lists, which live in registers from the
    // first tile to the output
    double* qd = s_lists;                                // [16][FZ_KM_QC]
    int* qj = reinterpret_cast<int*>(s_lists + 16 * FZ_KM_QC);
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef FZ_KM_STATS
    long long kmtime[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kmlast = (long long)clock64();
    unsigned kmc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int tree = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * 16;
    const float* bm = bmat + (size_t)tree * (Mp >> 6) * FZ_KM_TSTR;
    const int row = lane & 15, sl = lane >> 4;
    const int ntiles = (M + TILE - 1) / TILE, nw = (ntiles + 63) >> 6;
    double* Ld = s_lists;                                                                // [16][kpad]
    int* Lj = reinterpret_cast<int*>(s_lists + (size_t)16 * kpad);
    unsigned long long* tm = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(s_lists) + knn_mfma_list_bytes(kpad, KPL));
    if constexpr (KPL == 0) for (int e = lane; e < 16 * kpad; e += 64) { Ld[e] = INFINITY; Lj[e] = M + e % kpad; }
    for (int w = lane; w < nw; w += 64) tm[w] = (w + 1 < nw || (ntiles & 63) == 0) ? ~0ull : ((1ull << (ntiles & 63)) - 1ull);     // every tile, until the first build
    if (lane < 16) qn[lane] = 0;
    // ---- the wave's 16 queries: fp64 copies in LDS, A operands in registers ----
    // qperm (may be null): the queries grouped by their leaf in set 0's tree -- the 16 queries of a wave are neighbours in feature space
    const int64_t islot = i0 + row < N ? i0 + row : N - 1;
    const int64_t qi = qperm ? (int64_t)qperm[islot] : islot;       // this lane's row, as an index into q / idx
    for (int ff = sl; ff < 6; ff += 4) {
        const double v = (ff < F) ? q[qi * F + ff] : 0.0;
        qs[row][ff] = v;
        const float qf = (float)v, eq = 2.0e-7f * fabsf(qf);         // the conversion's rounding and that of qf -+ eq, with room
        qbx[row][ff] = qf - eq; qbx[row][6 + ff] = qf + eq;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): own LDS writes done (the same wave reads them)
    float qc[6]; double qn2 = 0.0, qdc = 0.0, c2 = 0.0;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const double cf = (f < F) ? (double)cen[tree * 8 + f] : 0.0;
        qc[f] = (f < F) ? (float)(qs[row][f] - cf) : 0.f;
        qn2 = fma((double)qc[f], (double)qc[f], qn2); qdc = fma((double)qc[f], cf, qdc); c2 = fma(cf, cf, c2);
    }
    const float alpha = (float)(qn2 + 2.0 * qdc);
    {
        const double u = 6.0e-8, Q = sqrt(qn2) * 1.000001, P = (double)pmax[tree], C = sqrt(c2) * 1.000001;
        const float uq2 = (float)(2.0 * u * Q * 1.000001);
        const float e = (float)(u * (19.0 * Q * Q + 18.0 * P * P + 70.0 * Q * C + 32.0 * Q * P) * 1.000001);
        if (sl == 0) { float2 pk; pk.x = uq2; pk.y = e; *reinterpret_cast<float2*>(&qs[row][7]) = pk; qs[row][6] = bound2; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // A operands: slots -2 q'_f | alpha - bar | 1 | 0; the bar slot F sits in a0 (F < 4) or a1 of the lanes with sl == F & 3
    auto aslot = [&](int s, float ab) -> float {
        float v = (s == F) ? ab : ((s == F + 1) ? 1.f : 0.f);
#pragma unroll
        for (int f = 0; f < 6; ++f) if (s == f && f < F) v = -2.f * qc[f];
        return v;
    };
    // The first tile of an unbounded search (the queries' own leaf) is multiplied with bar 0 -- the products ARE the squared distances,
    // to fp32 rounding -- and only the ~k nearest of its 64 models per row go to the lists first (run_tile); the tile is then visited
    // again as any other, under the bars those leave.  (All 64 pass an infinite bar, in storage order: 48 insertions per row where
    // ~22 do, a quarter of all the insertions of a scan.)
    bool first = KPL > 0 && !(bound2 < 1e37), second = false;
    unsigned pm_first = 0u;
    float a0, a1, barrow;                               // barrow: the row's bar in distance units (lanes 0..15 hold rows 0..15)
    {
        const float2 pk = *reinterpret_cast<const float2*>(&qs[row][7]);
        barrow = knn_bar_mfma(qs[row][6], pk.x, pk.y);
        const float ab = first ? alpha : alpha - barrow;
        a0 = aslot(sl, ab); a1 = aslot(4 + sl, ab);
    }
    const bool bar_lane = sl == (F & 3);

    // ---- row teams: the lists in registers, the drain ----
    constexpr int KP = KPL ? KPL : 1;
    double Lr[KP]; int Jr[KP];
    const int team = lane >> 2, tl = lane & 3;
    if constexpr (KPL > 0) {
#pragma unroll
        for (int e = 0; e < KPL; ++e) { Lr[e] = INFINITY; Jr[e] = M + tl * KPL + e; }           // (distinct keys for the empty entries)
    }
    auto quad_prev_i = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_update_dpp(v, v, 0x90, 0xf, 0xf, false); };     // quad_perm [0,0,1,2]: lane t <- lane t - 1
    auto refresh_bars = [&]() __attribute__((always_inline)) {                          // every lane: the bar of its row from the row's k-th distance
        const float2 pk = *reinterpret_cast<const float2*>(&qs[row][7]);
        barrow = knn_bar_mfma(qs[row][6], pk.x, pk.y);
        const float ab = alpha - barrow;
        if (bar_lane) { if (F < 4) a0 = ab; else a1 = ab; }
    };
    auto drain = [&]() __attribute__((always_inline)) {
        if constexpr (KPL > 0) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const int cnt = qn[team] < FZ_KM_QC ? qn[team] : FZ_KM_QC;
            KMSTAT(5, 1);
            for (int h = 0; __any(h < cnt); ++h) {
                const bool act = h < cnt;
                KMSTAT(6, 1);
                const double dn = qd[team * FZ_KM_QC + (act ? h : 0)];
                const int jn = qj[team * FZ_KM_QC + (act ? h : 0)];
                // c[e]: entry e precedes the candidate in (distance, index) order (a model is met once per scan: no duplicates) -- or
                // the team sits the round out.  The list is sorted, so c[] is a prefix; the entries after it move up by one, the
                // candidate (or, in the lanes above the one it lands in, the lower lane's last entry) takes the first free place.
                // Selects on the comparison masks only: no position count, no branch (a branch around the shift costs a register copy
                // of the whole list per round).
                bool c[KPL];
#pragma unroll
                for (int e = 0; e < KPL; ++e) {
                    const bool lt = Lr[e] < dn, eq = Lr[e] == dn, jl = Jr[e] < jn;
                    c[e] = !act | lt | (eq & jl);
                }
                const bool first = (tl == 0) | (quad_prev_i(c[KPL - 1] ? 1 : 0) != 0);      // every entry of the lower lane precedes the candidate: it lands in this lane or a higher one
                const int lh = quad_prev_i(__double2hiint(Lr[KPL - 1])), ll = quad_prev_i(__double2loint(Lr[KPL - 1]));
                const int jl = quad_prev_i(Jr[KPL - 1]);
                const double nd = first ? dn : __hiloint2double(lh, ll);
                const int nj = first ? jn : jl;
                KMSTAT(7, __builtin_popcountll(__ballot(act && !c[KPL - 1] && first)));   // rows that take an entry this round
#pragma unroll
                for (int e = KPL - 1; e >= 1; --e) {
                    Lr[e] = c[e] ? Lr[e] : (c[e - 1] ? nd : Lr[e - 1]);
                    Jr[e] = c[e] ? Jr[e] : (c[e - 1] ? nj : Jr[e - 1]);
                }
                Lr[0] = c[0] ? Lr[0] : nd;
                Jr[0] = c[0] ? Jr[0] : nj;
            }
            // the k-th distance of every row, then the bars
            // (one predicated store per slot: a select chain over the slots becomes an indexed load and the list goes to scratch)
            const int ke = k - 1;
#pragma unroll
            for (int e = 0; e < KPL; ++e)
                if (tl * KPL + e == ke) qs[team][6] = Lr[e] < bound2 ? Lr[e] : bound2;
            if (tl == 0) qn[team] = 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            refresh_bars();
        }
    };

    // Visiting order.  The set's models are stored in k-d order (upload: depth-first leaves) and the wave's queries are neighbours
    // (qperm), so the tiles around the queries' own leaf hold most of their neighbours: start there and work outwards,
    // alternating sides.  The bars then drop to nearly their final values within the first few tiles and the rest of the scan
    // admits little.  The lists are ordered by (distance, original index), so the result does not depend on the order.
    // Tile skipping.  Every 64-model tile has a bounding box (tbox), every group of 2^gsl tiles another (gbox).  build_mask tests the
    // <= 128 group boxes against the 16 rows' bars (lane l: groups l and l + 64), then the tiles of the reachable groups, 64 at a
    // time (lane l: tile 64 w + l, its box one coalesced 64-byte read; the 16 rows in a loop) and leaves one bit per tile in LDS;
    // the scan then only visits set bits: no per-tile test, no dependent global read between two tiles.  (Round 3-4 tested the next
    // four tiles of the visiting order just before they were staged, against the bars of that moment: fewer tiles multiplied --
    // 110 per wave -- but a chain of dependent L2 round trips, 36-44 % of a wave's cycles.)  A cold scan starts without a mask,
    // builds it after FZ_KM_B1 tiles -- the own leaf and its neighbours, which set the bars -- and again after FZ_KM_B2.
    KnnScan sc;
    sc.nl = -1; sc.nr = 0; sc.ntiles = ntiles; sc.right = 1;
    sc.m0 = ~0ull; sc.m1 = ~0ull; sc.gsl = gbox ? gsl : 30; sc.tm = tm;
    auto build_mask = [&]() __attribute__((always_inline)) {
        if (!gbox) return;
        const fz_f4* qb4 = reinterpret_cast<const fz_f4*>(&qbx[0][0]);
        // reach of one box (this lane's) by any of the 16 rows
        auto reach16 = [&](const fz_f4 l0, const fz_f4 l1, const fz_f4 h0, const fz_f4 h1) __attribute__((always_inline)) -> bool {
            bool reach = false;
#pragma unroll 1
            for (int R = 0; R < 16; ++R) {
                const float br = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(barrow), R));
                const fz_f4 a = qb4[R * 3], b = qb4[R * 3 + 1], c = qb4[R * 3 + 2];      // lo0..3 | lo4 lo5 hi0 hi1 | hi2..5
                const float ql[6] = {a[0], a[1], a[2], a[3], b[0], b[1]}, qh[6] = {b[2], b[3], c[0], c[1], c[2], c[3]};
                float lb = 0.f;
#pragma unroll
                for (int f = 0; f < FL; ++f) {
                    const float lo = f < 4 ? l0[f & 3] : l1[f & 3], hi = f < 4 ? h0[f & 3] : h1[f & 3];
                    const float m = fmaxf(fmaxf(lo - qh[f], ql[f] - hi), 0.f);
                    lb = fmaf(m, m, lb);
                }
                reach = reach || (lb * 0.999999f <= br);
            }
            return reach;
        };
        unsigned long long mm[2];
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const fz_f4* bx = reinterpret_cast<const fz_f4*>(gbox + ((size_t)tree * 128 + lane + 64 * h) * 16);
            mm[h] = __ballot(reach16(bx[0], bx[1], bx[2], bx[3]));
        }
        sc.m0 = mm[0]; sc.m1 = mm[1];
        const float* tb = tbox + (size_t)tree * (Mp >> 6) * 16;
        const int gs = sc.gsl;
        if (gs >= 4 && gs <= 6) {
            // groups of 16 / 32 / 64 tiles (65 k < M <= 524 k models): the tiles of 4 / 2 / 1 REACHABLE groups per pass, whatever words
            // they sit in (word by word, a pass would test 64 tiles for every word that holds one reachable group of 16: twice the tests)
            for (int w = lane; w < nw; w += 64) tm[w] = 0ull;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const int lpg = 1 << gs, gpp = 64 >> gs, c = lane >> gs, tin = lane & (lpg - 1);
            unsigned long long ma = sc.m0, mb = sc.m1;
#pragma unroll 1
            while (ma | mb) {
                unsigned pk = 0u; int nc = 0;
#pragma unroll 1
                while (nc < gpp && (ma | mb)) {
                    int g;
                    if (ma) { g = __builtin_ctzll(ma); ma &= ma - 1ull; } else { g = 64 + __builtin_ctzll(mb); mb &= mb - 1ull; }
                    const int ta = g << gs;
                    if (ta >= ntiles || (ta > sc.nl && ta + lpg - 1 < sc.nr)) continue;      // no such tiles / every tile of the group has been visited
                    pk |= (unsigned)g << (8 * nc); ++nc;
                }
                if (!nc) break;
                const int g = (int)((pk >> (8 * c)) & 0xffu), tile = (g << gs) + tin;
                const bool v = c < nc && tile < ntiles;
                const fz_f4* bx = reinterpret_cast<const fz_f4*>(tb + (size_t)(v ? tile : 0) * 16);
                const bool rc = reach16(bx[0], bx[1], bx[2], bx[3]);
                const unsigned long long bal = __ballot(v && rc);
                if (tin == 0 && c < nc) {
                    const unsigned long long bits = (bal >> (c << gs)) & (lpg == 64 ? ~0ull : ((1ull << lpg) - 1ull));
                    atomicOr(&tm[tile >> 6], bits << (tile & 63));
                }
            }
        } else {
#pragma unroll 1
            for (int w = 0; w < nw; ++w) {
                const int t0 = w << 6;
                if (t0 > sc.nl && t0 + 63 < sc.nr) continue;                  // every tile of the word has been visited
                const int tile = t0 + lane, g = (tile < ntiles ? tile : ntiles - 1) >> gs;
                const unsigned long long mw = g < 64 ? sc.m0 : sc.m1;
                const bool gr = tile < ntiles && ((mw >> (g & 63)) & 1ull);
                unsigned long long word = 0ull;
                if (__any(gr)) {
                    const fz_f4* bx = reinterpret_cast<const fz_f4*>(tb + (size_t)(tile < ntiles ? tile : ntiles - 1) * 16);
                    const bool rc = reach16(bx[0], bx[1], bx[2], bx[3]);
                    word = __ballot(gr && rc);
                }
                if (lane == 0) tm[w] = word;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    };
    bool need_mask = false;
    if (ktab) {
        // the leaf of the wave's middle query in THIS set's tree (11 dependent pairs of scalar loads at M = 1e5; the query straight from
        // LDS: a dynamic index into registers would go to scratch)
        const int hm = knn_kd_leaf(&qs[8][0], ktab + (size_t)tree * ntiles * 2, ntiles);
        sc.nr = __builtin_amdgcn_readfirstlane(hm < ntiles ? hm : ntiles - 1); sc.nl = sc.nr - 1;
    }
    auto stage = [&](int tile, float* dst) __attribute__((always_inline)) {             // 2048 contiguous bytes: 128 16-byte chunks over the wave's 64 lanes
        const char* src = reinterpret_cast<const char*>(bm + (size_t)tile * TF);
#pragma unroll
        for (int c = 0; c < 2; ++c)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * 1024 + (uint32_t)tid * 16u),
                                             (__attribute__((address_space(3))) void*)(dst + c * 256), 16, 0, 0);
    };
    int ndone = 0;
    // admission path.  pm: this lane's 16-bit mask of (g, r) products under the bar.
    auto slow = [&](unsigned pm, const float* blk, int jb) __attribute__((always_inline)) {
        // The four row bits of every model group are rotated by the lane's column, so that ONE pass of the loop spreads its pairs
        // over all 16 query rows (4 each when every pair of the tile passes -- the first tile of a scan, where nearly half of all
        // admissions happen) instead of 16 pairs for each of 4 rows: the row teams' drain then runs with every team busy
        // (329 -> 230 drain rounds per wave and set on the benchmark; 27.2 -> 23.9 ms per step).
        const int rot = row & 3;
        pm = ((pm >> rot) & (0x1111u * (0xFu >> rot))) | ((pm << (4 - rot)) & (0x1111u * ((0xF0u >> rot) & 0xFu)));
        while (__any(pm != 0u)) {
            // every lane with something left takes its lowest pair: query row 4 sl + r, model 16 g + col of the step
            const bool has = pm != 0u;
            const int e = has ? __builtin_ctz(pm) : 0;
            pm &= pm - 1u;
            const int g = e >> 2, r = (e + rot) & 3, R = 4 * sl + r;
            const int jpos = jb + 16 * g + row;                      // position in the (k-d-ordered) set
            double qv[FL]; float pv[FL];                             // exact distance from the original query and features
#pragma unroll
            for (int f = 0; f < FL; ++f) { qv[f] = qs[R][f]; pv[f] = blk[(f >> 2) * 256 + ((f & 3) * 16 + row) * 4 + g]; }
            const int j = (FX ? FX <= 5 : F <= 5) ? (int)blk[256 + (48 + row) * 4 + g] : jpos;       // the model's original index (slot 7)
            const double taur = qs[R][6];
            double d2 = 0.0;
#pragma unroll
            for (int f = 0; f < FL; ++f) { const double d = (FX || f < F) ? qv[f] - (double)pv[f] : 0.0; d2 = fma(d, d, d2); }
            const bool adm = has && jpos < M && d2 <= taur && d2 < bound2;
            KMSTAT(2, 1); KMSTAT(3, __builtin_popcountll(__ballot(has))); KMSTAT(4, __builtin_popcountll(__ballot(adm)));
            if constexpr (KPL > 0) {
                int slot = 0;
                if (adm) {
                    slot = atomicAdd(&qn[R], 1);
                    if (slot < FZ_KM_QC) { qd[R * FZ_KM_QC + slot] = d2; qj[R * FZ_KM_QC + slot] = j; }
                    else pm |= 1u << e;                            // the row's queue is full: again after the drain (which this very slot number triggers)
                }
                if (__any(adm && slot + 1 >= FZ_KM_DT) || (first && !__any(pm != 0u))) { KMT(4); drain(); KMT(5); }     // (first tile: its bars before anything else)
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
                float2 pk = *reinterpret_cast<const float2*>(&qs[Rn][7]);
                asm volatile("" : "+v"(l), "+v"(lj), "+v"(pk2), "+v"(pk.x), "+v"(pk.y));     // all loads in flight before anything is consumed
                const unsigned long long before = __ballot(lane < k && (l < dn || (l == dn && lj < jn)));     // lexicographic (distance, index)
                int pos = __builtin_popcountll(before);
                if (k > 64) {
                    // lists longer than the wave (64 < k <= FZ_KNN_KMAX; knn.py:190-193 takes any k): the rank over the further 64-entry
                    // segments, then the shift from the top segment down -- a segment's entries are read by all of its lanes before any
                    // of them is written one place up, and the place its last entry moves to belongs to the segment above, already moved
                    for (int sg = 64; sg < k; sg += 64) {
                        const int e = sg + lane;
                        const double le = ldr[e < k ? e : 0];
                        const int je = ljr[e < k ? e : 0];
                        pos += __builtin_popcountll(__ballot(e < k && (le < dn || (le == dn && je < jn))));
                    }
                    if (pos < k) {
                        for (int sg = (k - 1) & ~63; sg >= 64; sg -= 64) {
                            const int e = sg + lane;
                            const double le = ldr[e < k ? e : 0];
                            const int je = ljr[e < k ? e : 0];
                            if (e >= pos && e < k - 1) { ldr[e + 1] = le; ljr[e + 1] = je; }
                            if (e == pos) { ldr[e] = dn; ljr[e] = jn; }
                        }
                    }
                }
                const bool ok = pos < k;                              // (wave-uniform) pos == k: the row's bar moved while draining
                if (ok && lane >= pos && lane < k - 1) { ldr[lane + 1] = l; ljr[lane + 1] = lj; }     // (k > 64: lane 63 moves into the segment above)
                if (ok && lane == pos) { ldr[lane] = dn; ljr[lane] = jn; }
                if (ok) {
                    const double nk = (pos >= k - 1) ? dn : pk2;     // the new k-th distance
                    const double tau = nk < bound2 ? nk : bound2;
                    const float nbar = knn_bar_mfma(tau, pk.x, pk.y);
                    const float ab = alpha - nbar;
                    if (row == Rn) barrow = nbar;
                    if (lane == 0) qs[Rn][6] = tau;
                    if (bar_lane && row == Rn) { if (F < 4) a0 = ab; else a1 = ab; }
                }
            }
        }
    };
    int cur_t = sc.next_tile();
    auto run_tile = [&](const float* cur, float* nxt) __attribute__((always_inline)) -> bool {
        KMT(3);
        const int nx = first ? cur_t : sc.next_tile();           // (the first tile twice, see above)
        if (nx >= 0) stage(nx, nxt);
        KMT(2);
        const int t = cur_t;
        KMSTAT(1, 1);
        const fz_f4 b0 = *reinterpret_cast<const fz_f4*>(cur + lane * 4);
        const fz_f4 b1 = *reinterpret_cast<const fz_f4*>(cur + 256 + lane * 4);
        fz_f4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[g], fz_f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[g], acc[g], 0, 0, 0);
        unsigned pm = 0u;
        bool anyp;
        if (first) {
            // per row (4 r of this lane's group x its 16 lanes = the row's 64 models) a value T with #(acc <= T) >= k, close to the
            // k-th smallest: six bisection steps between the row's extremes.  Any T gives the right table (the second visit screens
            // what is left with the rigorous bars); a good one makes the first k insertions the only ones.
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float lo = fminf(fminf(acc[0][r], acc[1][r]), fminf(acc[2][r], acc[3][r]));
                float hi = fmaxf(fmaxf(acc[0][r], acc[1][r]), fmaxf(acc[2][r], acc[3][r]));
                lo = fminf(lo, row16_xchg<0xB1>(lo)); lo = fminf(lo, row16_xchg<0x4E>(lo)); lo = fminf(lo, row16_xchg<0x141>(lo)); lo = fminf(lo, row16_xchg<0x140>(lo));
                hi = fmaxf(hi, row16_xchg<0xB1>(hi)); hi = fmaxf(hi, row16_xchg<0x4E>(hi)); hi = fmaxf(hi, row16_xchg<0x141>(hi)); hi = fmaxf(hi, row16_xchg<0x140>(hi));
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const float mid = 0.5f * (lo + hi);
                    int cnt = (acc[0][r] <= mid ? 1 : 0) + (acc[1][r] <= mid ? 1 : 0) + (acc[2][r] <= mid ? 1 : 0) + (acc[3][r] <= mid ? 1 : 0);
                    cnt += row16_xchg<0xB1>(cnt); cnt += row16_xchg<0x4E>(cnt); cnt += row16_xchg<0x141>(cnt); cnt += row16_xchg<0x140>(cnt);
                    const bool ge = cnt >= k;
                    hi = ge ? mid : hi; lo = ge ? lo : mid;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) pm |= (acc[g][r] <= hi ? 1u : 0u) << (4 * g + r);
            }
            pm_first = pm;
            anyp = true;
        } else {
            int sg = 0;                                          // OR of the 16 sign bits: any product under its row's bar
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) sg |= __float_as_int(acc[g][r]);
            anyp = __any(sg < 0);
            if (anyp) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pm |= (__float_as_uint(acc[g][r]) >> 31) << (4 * g + r);
                if (second) pm &= ~pm_first;                     // the pairs the first visit took
            }
        }
        if (anyp) {
            KMT(3);
            slow(pm, cur, t * TILE);
            KMT(4);
        }
        second = first; first = false;
        cur_t = nx;
        // the bars have dropped since the scan began (and since the first mask)
        ++ndone;
        need_mask = need_mask || ndone == FZ_KM_B1 || ndone == FZ_KM_B2;
        KMT(3);
        __syncthreads();
        KMT(7);
        return nx >= 0;
    };
    KMSTAT(0, 1);
    KMT(0);
    if (cur_t >= 0) {
        stage(cur_t, tA);
        __syncthreads();
        while (true) {
            if (need_mask) {                                         // (the one call site: inlined three times it cost the kernel its occupancy)
                KMT(3);
                build_mask(); need_mask = false;                      // (from the bars of the last drain: at most FZ_KM_DT - 1 entries per row are waiting)
                // the tile in flight was chosen before the mask: it stays (a superset is always valid); the next ones follow the mask
                KMT(1);
            }
            if (!run_tile(tA, tB)) break;
            if (!run_tile(tB, tA)) break;
        }
    }
    KMT(3);
    if constexpr (KPL > 0) {
        drain();
        KMT(5);
        const int64_t i = __shfl(qi, team, 64);                     // the team's row (lanes 0..15 hold rows 0..15)
        if (i0 + team < N) {
#pragma unroll
            for (int e = 0; e < KPL; ++e)
                if (tl * KPL + e < k) idx[(i * K + tree) * k + tl * KPL + e] = (Lr[e] < bound2) ? Jr[e] : M;
        }
    } else {
        KMT(5);
        for (int R = 0; R < 16; ++R) {
            const int64_t i = __shfl(qi, R, 64);
            if (i0 + R < N)
                for (int e = lane; e < k; e += 64) idx[(i * K + tree) * k + e] = (Ld[R * kpad + e] < bound2) ? Lj[R * kpad + e] : M;
        }
    }
    KMT(6);
#ifdef FZ_KM_STATS
    for (int u = 0; u < 8; ++u) { KMFLUSH(u, kmc[u]); KMFLUSH(8 + u, kmtime[u]); }
#endif
}

}  // namespace fz
