// k_plane_rows: predict() from a stored (N, M) ln-weight plane when a row fits the REGISTERS of one block
// (BruteForce._predict, bruteforce.py:303-372 -> pdf.py:585-622 with a single dictionary kernel).
//
// k_plane_fused streams a row once but cannot know the row's maximum while it does, so every entry within the weight
// threshold of the RUNNING best -- one in ten on the benchmark plane -- is written to a per-wave list in HBM and walked
// twice afterwards (PMC, profiles/r3_v2_pmc_predict_before.txt: one 16-B store per 64 entries read, 2.7 GB of list
// traffic beside the 8 GB plane, 44 vector instructions per entry).  Rows up to NW * 64 * 2 * E2 entries long (10 240 at
// the (8, 10) shape: the benchmark's 1e4 models) need none of that: the block loads the whole row into registers (E2
// 16-byte non-temporal loads per lane, all in flight together), takes the exact maximum (one LDS exchange), and then
// weighs every entry in fp64 against it: w = exp(l - max), the evidence sum, and -- the reference's rule
// wt > wt_thresh * max(wt) (pdf.py:591) is w > wt_thresh in these units -- one ds_add_f64 into the block's histogram at
// the entry's label index for the stacked ones.  A lane always holds the same model columns, so their label indices sit
// in registers for the life of the block: no gathers.  Entries within 1e-9 of the threshold are decided by the
// reference's own expression once the evidence is known (parked in LDS until then; a row with more of them than the block
// parks is read once more).  The next row's loads are
// issued before the convolution of the current one, which touches LDS and registers only (kernel taps in two registers
// per wave, kernel masses staged in LDS), so they fly while the PDF is formed.  Everything is fp64: there is no fp32
// remainder in the evidence and no list in HBM; traffic is the plane once and the PDFs once.
#pragma once
#include "fz_hist.h"

namespace fz {

// wave-wide sum / maximum of a double through the DPP network (no LDS pipe: the shuffles of wave_sum / wave_max take six
// dependent ds_bpermute round trips); the result is wave-uniform, in scalar registers
__device__ __forceinline__ double dpp_shuffle_d(double v, int ctrl_quad1, int ctrl_quad2, int which) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (which == 0) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); }
    else if (which == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); }
    else if (which == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); }
    else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); }
    (void)ctrl_quad1; (void)ctrl_quad2;
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double plane_readlane_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v += dpp_shuffle_d(v, 0, 0, k);          // every lane: the sum of its row of 16
    return (plane_readlane_d(v, 0) + plane_readlane_d(v, 16)) + (plane_readlane_d(v, 32) + plane_readlane_d(v, 48));
}
__device__ __forceinline__ double wave_max_dpp(double v) {             // nan operands are skipped (v_max_f64)
#pragma unroll
    for (int k = 0; k < 4; ++k) v = vmax_raw(v, dpp_shuffle_d(v, 0, 0, k));
    return vmax_raw(vmax_raw(plane_readlane_d(v, 0), plane_readlane_d(v, 16)), vmax_raw(plane_readlane_d(v, 32), plane_readlane_d(v, 48)));
}

// ln(x) for x in [1, 1e300] without tables in memory and without a constant pool (libm's log keeps polynomial constants in
// registers across the object loop: they spill, and the reload waits for the row in flight): v_log_f32 gives y to 1e-7, then
// ln x = y + ln(x e^-y) = y + d - d^2 / 2 with d = x e^-y - 1 (|d| < 2e-7: the cubic term is 3e-21); e^-y from the LDS table
__device__ __forceinline__ double log_by_exp(double x, const double* __restrict__ s_exp) {
    const double y = (double)(__builtin_amdgcn_logf((float)x) * 0.69314718f);
    const double d = fma(x, exp_small_tab(-y, s_exp), -1.0);
    return y + fma(-0.5 * d, d, d);
}

// exp(x) for x <= 0 (or nan -> e^-700): exp_small_tab without the upper clamp and with the degree-4 polynomial
// (|r| <= ln2 / 512: the dropped term r^5 / 120 is below 4e-17)
__device__ __forceinline__ double exp_nonpos_tab(double x, const double* __restrict__ tab) {
    const double MAGIC = 6755399441055744.0;                     // 1.5 * 2^52
    x = vmax_raw(x, -700.0);
    const double d = fma(x, 369.3299304675746, MAGIC);           // 256 / ln 2
    const double r = fma(d - MAGIC, -0.0027076061740622863, x);  // ln 2 / 256
    const int n = __double2loint(d);
    const double t = tab[n & (FZ_HEXP_K - 1)];
    double p = fma(r, 1.0 / 24.0, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double v = t * p;                                      // in [1,2)
#ifdef FZ_PLANE_EXP_BITS
    return __hiloint2double(__double2hiint(v) + ((n >> 8) << 20), __double2loint(v));
#else
    return __builtin_amdgcn_ldexp(v, n >> 8);                    // (n >> 8 >= -1011: a normal number, the same bits as the exponent-field add)
#endif
}

typedef double plane_d4 __attribute__((ext_vector_type(4)));

// development build (-DFZ_PLANE_STATS, tools/mainbuild.sh): wall-clock cycles of every wave of k_plane_rows by section, summed
// over the launch and printed by the launcher (per wave and row) -- the attribution DESIGN 3.3 asks for
#ifdef FZ_PLANE_STATS
__device__ unsigned long long fz_plstats[16];
#define PLT(i) do { const long long plt_ = (long long)clock64(); pltime[i] += plt_ - pllast; pllast = plt_; } while (0)
#else
#define PLT(i) do { } while (0)
#endif

// K steps (of four) of the convolution as a matrix product, see the epilogue of k_plane_rows
__host__ __device__ inline int plane_conv_ksteps(int w2) { return (16 + w2 + 3) >> 2; }

template <int NW, int E2>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4))) void k_plane_rows(const double* __restrict__ plane, int64_t ld, const KdeView* __restrict__ kvp,
                                                         int acc_stride, int64_t N, int M, double wt_thresh, int normalize,
                                                         double* __restrict__ lmap, double* __restrict__ levid, double* __restrict__ pdfs) {
    constexpr int NT = NW * 64, CAPA = NT;
    extern __shared__ double smem[];
    double* s_exp = smem;                               // [FZ_HEXP_K] 2^(k/256)
    double* hist = s_exp + FZ_HEXP_K;                   // [acc_stride] x 3: the histograms of the row being weighed, of the previous one (its epilogue) and the one being cleared
    double* s_inv = hist + 3 * acc_stride;              // [acc_stride] 1 / kernel mass per padded index
    double* s_max = s_inv + acc_stride;                 // [2][NW] per-wave maxima (by the parity of the row)
    double* s_sum = s_max + 2 * NW;                     // [2][2 NW] per-wave sums and stacked sums
    double* s_late = s_sum + 4 * NW;                    // [2] the late (ambiguous) stacked sum
    int* s_flag = reinterpret_cast<int*>(s_late + 2);   // [2][NW] bit 0: a nan in the wave's part, bit 1: the row's first entry is nan
    int* s_amb = s_flag + 2 * NW;                       // [2] entries within rounding of the threshold (count)
    int* s_tag = s_amb + 2;                             // [NT E2] padded label indices of models 2 k, 2 k + 1 in the halves of word k
    double* s_ambl = reinterpret_cast<double*>(s_tag + NT * E2);    // [CAPA] ln-weights within rounding of the threshold (decided once the evidence is known)
    int* s_ambp = reinterpret_cast<int*>(s_ambl + CAPA);            // [CAPA] their histogram indices
    double* s_T = reinterpret_cast<double*>(s_ambp + CAPA);        // [KS 64] the kernel taps as the B operands of the convolution's matrix product
    // (each exchange array is written in one barrier interval and read in the next one only -- two copies by the parity of the
    //  row -- so a wave that runs ahead into the next row can never overwrite what a slower one still reads)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const KdeView kv = *kvp;
    const int G = (int)kv.G, w0 = kv.w0, w2 = 2 * w0, GP = G + w2;
    const FastTabs tb = global_tabs();
    for (int k = tid; k < FZ_HEXP_K; k += NT) s_exp[k] = FZ_EXP_TAB[k * (FZ_EXP_K / FZ_HEXP_K)];
    for (int k = tid; k < acc_stride; k += NT) {
        hist[k] = 0.0; hist[acc_stride + k] = 0.0; hist[2 * acc_stride + k] = 0.0;
        s_inv[k] = (k < GP) ? 1.0 / kv.normtab[k] : 1.0;
    }
    if (tid < 2) { s_late[tid] = 0.0; s_amb[tid] = 0; }
    // a lane holds the same model columns for every object -- entry e: models 2 (e NT + tid) and +1 -- so their label indices
    // are staged once (LDS rather than registers: the row itself takes 4 E2 of them)
    for (int k = tid; k < NT * E2; k += NT) {
        const int j = 2 * k;
        const int p0 = (j < M) ? kv.pos[j] + w0 : 0, p1 = (j + 1 < M) ? kv.pos[j + 1] + w0 : 0;
        s_tag[k] = p0 | (p1 << 16);
    }
    // The convolution out[t] = sum_h row[t + h] kr[w2 - h] as a matrix product on the fp64 matrix pipe: with t = 16 a + b,
    // out[16 a + b] = sum_j R[a][j] T[j][b],  R[a][j] = row[16 a + j] (overlapping windows of the histogram),
    // T[j][b] = kr[w2 - (j - b)] for 0 <= j - b <= w2, else 0 -- a constant banded Toeplitz matrix, (16 + w2) x 16.
    // One v_mfma_f64_16x16x4 takes 16 windows (256 outputs) through four values of j; a wave owns 256 outputs.  T is staged
    // once per block in the operand layout of the instruction (lane: k = lane >> 4, column b = lane & 15).
    const double* kr = kv.kern + kv.koff0;
    const int KS = plane_conv_ksteps(w2);
    for (int k = tid; k < KS * 64; k += NT) {
        const int h = 4 * (k >> 6) + ((k & 63) >> 4) - (k & 15);
        s_T[k] = (h >= 0 && h <= w2) ? kr[w2 - h] : 0.0;
    }
    const double thr_hi = uniform_d(wt_thresh * (1.0 + 1e-9)), thr_lo = uniform_d(wt_thresh * (1.0 - 1e-9));     // wave-uniform values live in scalar registers
    __syncthreads();

    fz_d2 l[E2];
    // Row loads by hand: scalar base (the row) + ONE lane offset per entry formed on the spot, so the E2 requests need no
    // address registers beside the E2 x 4 they fill (the compiler's form: a 64-bit address pair and a branch per entry, and
    // parts of the row spilled to scratch).  Lanes past the end of a short row re-read its last entry and are set to -inf when
    // the data are used.  The compiler does not count these loads: row_wait() is the s_waitcnt, tied to every register of the row
    // (tests/test_abi.py checks the compiled code: nothing touches a row register between its request and the wait).
    const int last2 = M / 2 - 1;
    // FZ_PLANE_COUNTED_LOADS (the build's fallback when the listing check fails, __graft_entry__.check_hand_scheduled): ordinary
    // non-temporal loads the compiler counts and waits for itself -- slower (address registers, parts of the row may spill) but safe
    // with any register allocation.
    auto load_row = [&](int64_t i) {
        const char* rb = reinterpret_cast<const char*>(plane + i * ld);
#if defined(FZ_PLANE_COUNTED_LOADS)
#pragma unroll
        for (int e = 0; e < E2; ++e) {
            const unsigned off = (unsigned)min(e * NT + tid, last2) * 16u;
            l[e] = __builtin_nontemporal_load(reinterpret_cast<const fz_d2*>(rb + off));
        }
        return;
#endif
        int t = tid;
        asm volatile("" : "+v"(t));                                          // offsets formed here, per row (hoisted, the E2 of them spill)
        const unsigned off0 = (unsigned)t * 16u;
#pragma unroll
        for (int e = 0; e < E2; ++e) {
            if (2 * (e + 1) * NT <= M) {                                     // wave-uniform: the entry lies inside the row for every lane -- the lane's own
                // offset and the entry's start in the SCALAR base (two scalar adds instead of three vector instructions per request)
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(l[e]) : "v"(off0), "s"(rb + (size_t)e * NT * 16) : "memory");
            } else {
                const unsigned off = (unsigned)min(e * NT + t, last2) * 16u;
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(l[e]) : "v"(off), "s"(rb) : "memory");
            }
        }
    };
    auto row_wait = [&]() {
        static_assert(E2 == 5 || E2 == 10, "operand list below");
#if defined(FZ_PLANE_COUNTED_LOADS)
        if constexpr (false)
#else
        if constexpr (E2 == 10)
#endif
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(l[4]), "+v"(l[5]), "+v"(l[6]), "+v"(l[7]), "+v"(l[8]), "+v"(l[9]));
#if !defined(FZ_PLANE_COUNTED_LOADS)
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(l[4]));
#endif
#pragma unroll
        for (int e = 0; e < E2; ++e)
            if (2 * (e + 1) * NT > M) {                                      // wave-uniform: only the entries the row's end falls into (or past)
                if (2 * (e * NT + tid) >= M) l[e] = fz_d2{-INFINITY, -INFINITY};
            }
    };

    // ONE block barrier per row.  Iteration i: wait for row i, post its per-wave maxima, BARRIER -- behind it the maxima of row i
    // are complete, and so are the sums and the histogram of row i - 1 (posted / added before the barrier) -- then: clear the
    // histogram row i + 1 will use, settle the parked ties of row i - 1 (rare), weigh row i into its histogram, request row i + 1,
    // post the sums of row i, and only then the EPILOGUE OF ROW i - 1 (ln-evidence by wave 0, convolution by the waves that own
    // outputs, store) while the other waves already wait for row i + 1: the epilogue runs under the memory latency of the next row
    // instead of in front of it.  Three histograms rotate (weighed / convolved / cleared); every exchange word has a copy per row parity.
#ifdef FZ_PLANE_STATS
    long long pltime[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pllast = (long long)clock64(), plrows = 0;
#endif
    int64_t i = blockIdx.x;
    if (i < N) load_row(i);
    bool havep = false;                                                      // a previous row waits for its epilogue (block-uniform)
    int64_t ip = 0;
    double mxp = 0.0;
    int flp = 0;
    int par = 0, hb = 0;
    while (true) {
        const bool have = i < N;                                             // block-uniform
        if (!have && !havep) break;
        PLT(6);
        if (have) {
            row_wait();
            PLT(0);
            // ---- per-wave maximum (nan never becomes the best: v_max_f64 returns the other operand) ----
            double m = -INFINITY;
            bool an = false;
#pragma unroll
            for (int e = 0; e < E2; ++e) {
                m = vmax_raw(m, l[e].x); m = vmax_raw(m, l[e].y);
                an |= __builtin_isunordered(l[e].x, l[e].y);                  // one compare: either of the pair is a nan
            }
            const bool fnl = (tid == 0) && (l[0].x != l[0].x);
            m = wave_max_dpp(m);
            const int wf = (__any(an) ? 1 : 0) | (__any(fnl) ? 2 : 0);
            if (lane == 0) { s_max[par * NW + wave] = m; s_flag[par * NW + wave] = wf; }
        }
        PLT(1);
        __syncthreads();                                                     // the row's one barrier
        PLT(2);
        double* row = hist + hb * acc_stride;                                // row i (zero: cleared one row ago)
        double* rowp = hist + (hb == 0 ? 2 : hb - 1) * acc_stride;           // row i - 1: complete, convolved below
        {
            double* rowz = hist + (hb == 2 ? 0 : hb + 1) * acc_stride;       // row i - 2: its epilogue ended before the barrier; cleared for row i + 1
            int tz = tid;
            asm volatile("" : "+v"(tz));
            for (int k = tz; k < GP; k += NT) rowz[k] = 0.0;
        }
        // ---- row i - 1: its sums; its parked ties (block-uniform, rare): the reference's own expression decides ----
        double S = 0.0, T = 0.0;
        bool okp = false;
        if (havep) {
            const int namb = __builtin_amdgcn_readfirstlane(s_amb[par ^ 1]);
            if (namb || wave * 256 < G) {                                    // the sums: only the waves that write outputs (wave 0 among them) -- and everyone for the ties
                const double* ss = s_sum + (par ^ 1) * 2 * NW;
#pragma unroll
                for (int w = 0; w < NW; ++w) { S += ss[w]; T += ss[NW + w]; }
                S = uniform_d(S); T = uniform_d(T);
            }
            okp = !(flp & 1) && (mxp - mxp == 0.0);                          // a finite evidence: S >= 1 (the best entry itself)
            if (namb) {
                if (okp) {
                    const double le = mxp + log_by_exp(S, s_exp);
                    const double thr = wt_thresh * exp_neg(mxp - le, tb);   // wt_thresh * max(wt)
                    if (namb <= CAPA) {
                        for (int k = tid; k < namb; k += NT) {
                            const double lv = s_ambl[k];
                            const int p = s_ambp[k];
                            if (exp_neg(lv - le, tb) > thr) {               // strict
                                const double w = exp_nonpos_tab(lv - mxp, s_exp);
                                unsafeAtomicAdd(&rowp[p], w * s_inv[p]);
                                unsafeAtomicAdd(&s_late[par ^ 1], w);
                            }
                        }
                    } else {
                        // more ties than the block parks (a degenerate row): its entries are read once more and the band decided in place
                        const double* r = plane + ip * ld;
                        for (int k = tid; k < M / 2; k += NT) {
                            const int pt = s_tag[k];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const double lv = r[2 * k + q];
                                const double w = exp_nonpos_tab(lv - mxp, s_exp);
                                if (w >= thr_lo && !(w > thr_hi) && exp_neg(lv - le, tb) > thr) {
                                    const int p = q ? (pt >> 16) : (pt & 0xffff);
                                    unsafeAtomicAdd(&rowp[p], w * s_inv[p]);
                                    unsafeAtomicAdd(&s_late[par ^ 1], w);
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                T += uniform_d(s_late[par ^ 1]);
                __syncthreads();                                             // (the parked entries are free for row i from here on)
                if (tid == 0) { s_late[par ^ 1] = 0.0; s_amb[par ^ 1] = 0; }   // this parity's next use lies behind the next barrier
            }
        }
        PLT(3);
        // ---- row i: exact maximum, weights against it, evidence sum, histogram ----
        double mx = 0.0;
        int fl = 0;
        if (have) {
            mx = s_max[par * NW];
            fl = s_flag[par * NW];
#pragma unroll
            for (int w = 1; w < NW; ++w) { mx = vmax_raw(mx, s_max[par * NW + w]); fl |= s_flag[par * NW + w]; }
            mx = uniform_d(mx);
            fl = __builtin_amdgcn_readfirstlane(fl);
            const bool tame = (mx - mx == 0.0);                              // a finite best
            double s = 0.0, ssel = 0.0;
            if (tame) {
                int tt = tid;
                asm volatile("" : "+v"(tt));
                const int* tg = s_tag + tt;
#pragma unroll
                for (int e = 0; e < E2; ++e) {
                    const int pt = tg[e * NT];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const double lv = q ? l[e].y : l[e].x;
                        const double w = exp_nonpos_tab(lv - mx, s_exp);    // -inf, nan and pad columns: 1e-304
                        s += w;
                        if (w >= thr_lo) {                                   // stacked, or within rounding of the threshold
                            const int p = q ? (pt >> 16) : (pt & 0xffff);
                            if (w > thr_hi) {
                                ssel += w;
                                unsafeAtomicAdd(&row[p], w * s_inv[p]);     // weight / kernel mass of the index (pdf.py:613-617)
                            } else {                                         // (rare) parked until the evidence is known
                                const int k = atomicAdd(&s_amb[par], 1);
                                if (k < CAPA) { s_ambl[k] = lv; s_ambp[k] = p; }   // (more than CAPA of them: the row is read again)
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);                       // two entries at a time (all E2 at once spill the row)
                }
            }
            PLT(4);
            // ---- the row is used up: the next one is requested now, and everything below touches LDS and registers only ----
            const int64_t inext = i + gridDim.x;
            if (inext < N) load_row(inext);
            s = wave_sum_dpp(s);
            ssel = wave_sum_dpp(ssel);
            if (lane == 0) { s_sum[par * 2 * NW + wave] = s; s_sum[par * 2 * NW + NW + wave] = ssel; }
        }
        PLT(5);
        // ---- epilogue of row i - 1, under the latency of row i + 1 ----
#ifdef FZ_PLANE_STATS
        plrows += have ? 1 : 0;
#endif
        if (havep) {
            const bool anynan = flp & 1, firstnan = flp & 2;
            if (wave == 0) {                                                 // ln-evidence: one wave, no tables (a table load would wait for the row in flight)
                double le;
                if (anynan) le = (double)NAN;
                else if (!(mxp - mxp == 0.0)) le = mxp;                      // +inf, or -inf for a row of -inf
                else le = mxp + log_by_exp(S, s_exp);
                if (lane == 0) {
                    if (lmap) lmap[ip] = firstnan ? (double)NAN : mxp;       // builtin max: NaN only if first
                    if (levid) levid[ip] = le;
                }
            }
            // ---- PDF: convolve, normalise, write ----
            double* out = pdfs + ip * G;
            if (!okp) {
                int t0 = tid;
                asm volatile("" : "+v"(t0));
                for (int t = t0; t < G; t += NT) out[t] = NAN;
            } else if (wave * 256 < G) {                                     // wave-uniform: this wave's 256 outputs on the matrix pipe
                int ln = lane;
                asm volatile("" : "+v"(ln));                                 // (addresses formed per object: hoisted, they spill -- and a scratch reload here would wait for the row in flight)
                const int base = wave * 256 + 16 * (ln & 15) + (ln >> 4);    // A operand: window a = lane & 15 of this wave, k = lane >> 4
                const double* tp = s_T + ln;
                plane_d4 acc = {0.0, 0.0, 0.0, 0.0};
                for (int ks = 0; ks < KS; ++ks) {
                    // (indices past the padded row belong to taps that are zero or to outputs beyond the grid: any finite entry serves)
                    const double a = rowp[min(base + 4 * ks, GP - 1)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tp[ks * 64], acc, 0, 0, 0);
                }
                // pdf /= pdf.sum(): the sum over the grid of the convolved histogram is the sum over the indices of (weight / mass) x
                // (the taps that land on the grid) = the sum of the stacked weights -- known since the sums were exchanged, no second reduction
                const double scale = normalize ? 1.0 / T : 1.0 / S;
                // D: column b = lane & 15, window a = (lane >> 4) + 4 r
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = wave * 256 + 16 * ((ln >> 4) + 4 * r) + (ln & 15);
                    if (t < G) out[t] = acc[r] * scale;
                }
            }
        }
        havep = have; ip = i; mxp = mx; flp = fl;
        if (have) i += gridDim.x;
        par ^= 1;
        hb = (hb == 2) ? 0 : hb + 1;
    }
#ifdef FZ_PLANE_STATS
    PLT(6);
    if (lane == 0) {
        atomicAdd(&fz_plstats[0], 1ull); atomicAdd(&fz_plstats[1], (unsigned long long)plrows);
        for (int u = 0; u < 7; ++u) atomicAdd(&fz_plstats[2 + u], (unsigned long long)pltime[u]);
        if (wave * 256 < G) { atomicAdd(&fz_plstats[9], 1ull); atomicAdd(&fz_plstats[10], (unsigned long long)pltime[6]); }
    }
#endif
}

}  // namespace fz
