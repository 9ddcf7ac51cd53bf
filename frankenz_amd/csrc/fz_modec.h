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
//
// FAST solve (mask-free tame data): the eleven IEEE divisions of a solve (~12 instructions each, 40 % of it) become one
// Newton-refined reciprocal per band (and one for the shape) that the three quotients of the band share.  The iterates then
// differ from the IEEE ones in the last ulps -- eleven orders below ltol -- which can only change an object's iteration count if
// its max |dlnl| lands within rounding of ltol.  That case is DETECTED, not assumed away: every model's error e_j carries a
// bound d_j = 256 ulp (|lnl_new| + |lnl_old|) on what the two arithmetics can differ by, the step kernel reduces BOTH
// max_j (e_j - d_j) and max_j (e_j + d_j) per object, and an object for which ltol falls between the two at any iteration is
// flagged and re-run from the start with the IEEE solve.  Every other object provably takes the reference's iteration count.
#pragma once
#include "fz_device.h"

namespace fz {

struct ModeCState {
    double* s;    // scale           (Nc,M)
    double* l;    // Gaussian lnl    (Nc,M)
    double* c;    // chi2            (Nc,M)
    double* sh;   // shape           (Nc,M)
    unsigned long long* err;   // (Nc) max |dlnl| as ordered bits (FAST solve: max of the lower bounds |dlnl| - d)
    unsigned long long* errhi; // FAST solve: (Nc) max of the upper bounds |dlnl| + d
    int* firstnan;             // (Nc) |dlnl[0]| is NaN
    const int* list;           // objects this launch advances (nullptr: all Nc), written by the last check
    int* list_next;            // objects that go on (filled by k_modec_check)
    int* nactive;              // (1) length of list_next
    const int* ncur;           // (1) length of `list` on the device (nullptr: Nc): the host launches for an upper bound
    int* last_iter;            // (1) highest iteration after which some object still went on
    int* amb;                  // FAST solve: objects whose error came within rounding of ltol (list), may be nullptr
    int* namb;                 // (1) their number
    int* ambflag;              // (Nc) already listed
    // k_modec_persist only: the caller wants nothing but the final ln-like plane (fit_predict without stored fits) -- the converged
    // object's rows of l are written in their final form (lnl_only = 1: as they are; 2: with the dimensionality prior, pdf.py:226-229)
    // and s / c / sh are not written at all (three 8 B-per-pair planes and the k_modec_final pass less)
    int lnl_only;
    const double* lgtab;
    int* qhead;                // (1) k_modec_rounds: the next object of the launch (zeroed by the host)
    int rfixed;                // k_modec_rounds, tests (FZ_MODEC_RFIXED): every round this long instead of predicted -- rounds that run past the stop
    int* niter;                // (Nc) iterations each object took (the count of pdf.py:199's loop passes); may be nullptr
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
    static constexpr int NB = BT;
    static constexpr int REC_W = 2 * BT + ((6 - (2 * BT) % 4) % 4);       // doubles per row of the array-of-records copy (fluxes | squared errors | pad)
    __device__ __forceinline__ void load_rec(int64_t j, double (&rec)[REC_W]) const {
        const double2* rp = reinterpret_cast<const double2*>(mv.rec0) + (uint32_t)j * (uint32_t)(REC_W / 2);
#pragma unroll
        for (int q = 0; q < REC_W / 2; ++q) { const double2 w = rp[q]; rec[2 * q] = w.x; rec[2 * q + 1] = w.y; }
    }
    // the FAST solve on a record already in registers (fluxes | squared errors), see solve<true>
    __device__ __forceinline__ void solve_rec(int64_t i, const double (&rec)[REC_W], double sprev, double& s, double& lnl,
                                              double& chi2, double& shape, const double2* lt) const {
        double x[BT], v[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) { x[b] = ov.x[i * BT + b]; v[b] = ov.v[i * BT + b]; }
        solve_xv(x, v, rec, sprev, s, lnl, chi2, shape, lt);
    }
    // ... and with the object's row (fluxes x, variances v = xe^2) in registers too: k_modec_rounds runs several solves on one record
    __device__ __forceinline__ void solve_xv(const double (&x)[BT], const double (&v)[BT], const double (&rec)[REC_W], double sprev,
                                             double& s, double& lnl, double& chi2, double& shape, const double2* lt) const {
        double rv[BT], y[BT];
        double inter = 0.0; shape = 0.0;
        double vprod = 1.0; int vexp = 0;
        const double s2 = sprev * sprev;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            y[b] = rec[b];
            const double var = fma(s2, rec[BT + b], v[b]);              // xe^2 + s^2 ye^2
            rv[b] = rcp_nr<2>(var);
            const double yr = y[b] * rv[b];
            inter = fma(yr, x[b], inter);
            shape = fma(yr, y[b], shape);
            int e; vprod *= frexp(var, &e); vexp += e;
        }
        s = inter * rcp_nr<2>(shape);
        chi2 = 0.0;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            const double d = fma(-s, y[b], x[b]);
            chi2 = fma(d * d, rv[b], chi2);
        }
        FastTabs tl = global_tabs();
        if (lt) tl.logt = lt;
        const double slog = log_pos(vprod, tl) + (double)vexp * FZ_LN2;
        lnl = -0.5 * chi2 - 0.5 * ((double)nband * FZ_LN2PI + slog);
    }
    // lt: the 2 KB log table in LDS (k_modec_persist), or nullptr: the global copy
    template <bool FAST = false>
    __device__ __forceinline__ void solve(int64_t i, int64_t j, double sprev, double& s, double& lnl,
                                          double& chi2, double& shape, int& ndim, const double2* lt = nullptr) const {
        if constexpr (FAST && !MASKED) {
            if (sub.nbr) j = sub.nbr[i * sub.W + j];
            ndim = nband;
            // the model's record (fluxes, squared errors: 2 BT doubles in one 16-byte-aligned row of the array-of-records copy) in
            // BT 16-byte loads instead of 2 BT 8-byte ones from the band-major arrays: the record is re-read from L2 on every
            // iteration (it does not fit registers beside the other models of the thread), half the load instructions
            double rec[REC_W];
            load_rec(j, rec);
            solve_rec(i, rec, sprev, s, lnl, chi2, shape, lt);
            return;
        }
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

template <class MC, bool FAST = false>
__global__ __launch_bounds__(256) void k_modec_step(MC mc, ModeCState st, int64_t Nc, int64_t M, int init) {
    const int64_t tiles = (M + 255) / 256;
    const int64_t slot = blockIdx.x / tiles;             // Nc = number of objects of this launch (an upper bound when st.ncur is set)
    if (st.ncur && slot >= *st.ncur) return;             // block-uniform: this slot's object stopped since the host last looked
    const int64_t i = st.list ? (int64_t)st.list[slot] : slot;
    const int64_t j = (blockIdx.x % tiles) * 256 + threadIdx.x;
    const bool valid = j < M && (!mc.sub.nnb || j < mc.sub.nnb[i]);
    double e = 0.0, eh = 0.0;
    if (valid) {
        const int64_t k = i * M + j;
        double s, l, c, sh; int nd;
        if (init) {
            mc.template solve<FAST>(i, j, 1.0, s, l, c, sh, nd);
        } else {
            const double lold = st.l[k];
            mc.template solve<FAST>(i, j, st.s[k], s, l, c, sh, nd);
            e = fabs(l - lold);
            if (j == 0 && e != e) st.firstnan[i] = 1;
            if (FAST) {                                     // what the IEEE iterates could differ by: both ends of the interval
                const double dl = 2.9e-14 * (fabs(l) + fabs(lold));
                eh = e + dl; e = fmax(e - dl, 0.0);
            }
        }
        st.s[k] = s; st.l[k] = l; st.c[k] = c; st.sh[k] = sh;
    }
    if (init) return;
    // block max, NaNs dropped (a > b is false for NaN)
    if (!(e == e)) e = 0.0;
    if (!(eh == eh)) eh = 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { e = fmax(e, __shfl_xor(e, o, 64)); if (FAST) eh = fmax(eh, __shfl_xor(eh, o, 64)); }
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = e; red[4 + (threadIdx.x >> 6)] = eh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        e = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        atomicMax(&st.err[i], (unsigned long long)__double_as_longlong(e));   // e >= 0: bit order == value order
        if (FAST) {
            eh = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
            atomicMax(&st.errhi[i], (unsigned long long)__double_as_longlong(eh));
        }
    }
}

// ---- the whole fixed point of an object inside ONE block (M <= FZ_MCP_MAXM models) ----
// The plane kernels above move 48 B of state per (object, model, iteration) through HBM -- 290 GB on the 2e4 x 1e4 benchmark:
// and one launch pair per iteration.  Here a block of T threads owns an object for all its iterations: thread t
// keeps the scale and ln-like of models t, t + T, ... in registers, the previous scale of every model sits in LDS (the final
// chi2 / shape are functions of it: they are recomputed once after the stop instead of being carried), the stop rule is a
// block reduction, and nothing but the model records (L2-resident) is read per iteration.  Same arithmetic, same stop rule
// (builtin-max NaN semantics, FAST solve with interval detection).
// Blocks take objects from a grid-stride loop, so one object that needs hundreds of iterations delays nobody.
#define FZ_MCP_MAXM 16384                                // (LDS: 8 B per model)
// FAST: the reciprocal-based solve; an object found ambiguous (ltol inside the interval of its error) is listed in st.amb and
// left unfinished: the host runs the IEEE instantiation over that list.  MPT: models per thread (register arrays: 2 MPT doubles).
// T x MPT: threads per block x models per thread.  The solve itself holds ~90 VGPRs, the state 4 per model: (1024, <= 4) at
// 128 VGPRs, (768, 14) at 168 -- the 1e4-model benchmark -- and (512, 32) at 256 are the spill-free shapes.
template <class MC, bool FAST, int FZ_MCP_T, int MPT>
__global__ __launch_bounds__(FZ_MCP_T) void k_modec_persist(MC mc, ModeCState st, int64_t Nc, int M, double ltol, int max_iter, int* status) {
    extern __shared__ double s_old[];                     // [M] the scale every model's last solve started from
    __shared__ double red[2][FZ_MCP_T / 64];
    __shared__ int s_flag[2];
    // the log table (2 KB) in LDS: one table read per solve that would otherwise go to L2 beside the model record
    __shared__ __attribute__((aligned(16))) double s_logt[FAST ? 256 : 2];
    const int tid = threadIdx.x;
    const double2* lt = nullptr;
    if constexpr (FAST) {
        for (int k = tid; k < 256; k += FZ_MCP_T) s_logt[k] = FZ_LOG_TAB[k];
        lt = reinterpret_cast<const double2*>(s_logt);
        __syncthreads();
    }
    for (int64_t slot = blockIdx.x; slot < Nc; slot += gridDim.x) {
        const int64_t i = st.list ? (int64_t)st.list[slot] : slot;
        const int Mi = mc.sub.nnb ? (int)(mc.sub.nnb[i] < M ? mc.sub.nnb[i] : M) : M;
        double sc[MPT], ll[MPT];
#pragma unroll
        for (int m = 0; m < MPT; ++m) {
            int j = tid + m * FZ_MCP_T;
            asm volatile("" : "+v"(j));                  // re-formed per use: every (model, array) address kept live across the iteration loop spills the state
            sc[m] = 1.0; ll[m] = 0.0;
            if (j < Mi) {
                double c, sh; int nd;
                mc.template solve<FAST>(i, j, 1.0, sc[m], ll[m], c, sh, nd, lt);
                s_old[j] = 1.0;
            }
        }
        int iters = 0, f = 0;
        while (true) {
            double e = 0.0, eh = 0.0; int fnan = 0;
#pragma unroll
            for (int m = 0; m < MPT; ++m) {
                int j = tid + m * FZ_MCP_T;
            asm volatile("" : "+v"(j));                  // re-formed per use: every (model, array) address kept live across the iteration loop spills the state
                if (j < Mi) {
                    double sn, ln, c, sh; int nd;
                    mc.template solve<FAST>(i, j, sc[m], sn, ln, c, sh, nd, lt);
                    double ej = fabs(ln - ll[m]);
                    if (j == 0 && ej != ej) fnan = 1;
                    double ehj = ej;
                    if (FAST) { const double dl = 2.9e-14 * (fabs(ln) + fabs(ll[m])); ehj = ej + dl; ej = fmax(ej - dl, 0.0); }
                    if (ej == ej) e = fmax(e, ej);         // NaNs dropped (builtin max)
                    if (ehj == ehj) eh = fmax(eh, ehj);
                    s_old[j] = sc[m]; sc[m] = sn; ll[m] = ln;
                }
                __builtin_amdgcn_sched_barrier(0);       // one solve at a time: interleaving MPT of them spills the state arrays
            }
            ++iters;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { e = fmax(e, __shfl_xor(e, o, 64)); eh = fmax(eh, __shfl_xor(eh, o, 64)); }
            if ((tid & 63) == 0) { red[0][tid >> 6] = e; red[1][tid >> 6] = eh; }
            if (tid == 0) s_flag[1] = fnan;
            __syncthreads();
            if (tid == 0) {
                double E = 0.0, EH = 0.0;
                for (int w = 0; w < FZ_MCP_T / 64; ++w) { E = fmax(E, red[0][w]); EH = fmax(EH, red[1][w]); }
                const bool fn = s_flag[1] != 0;
                int g = (!fn && E > ltol) ? 1 : 0;                               // `while lerr > ltol`
                if (FAST && !fn && E <= ltol && EH > ltol) g = 2;                // ltol inside the interval: the IEEE iterates decide
                if (g == 1 && iters >= max_iter) g = 3;
                s_flag[0] = g;
            }
            __syncthreads();
            f = s_flag[0];
            __syncthreads();
            if (f != 1) break;
        }
        if (f == 3 && tid == 0) atomicMax(status, 1);
        if (f == 2) {                                     // left to the IEEE instantiation
            if (tid == 0) st.amb[atomicAdd(st.namb, 1)] = (int)i;
            continue;
        }
        // the last solve once more, from the scale it started from: its chi2 and shape (scale and ln-like are the registers')
#pragma unroll
        for (int m = 0; m < MPT; ++m) {
            int j = tid + m * FZ_MCP_T;
            asm volatile("" : "+v"(j));                  // re-formed per use: every (model, array) address kept live across the iteration loop spills the state
            if (j < Mi) {
                const int64_t k = i * M + j;
                if (st.lnl_only == 1) { st.l[k] = ll[m]; continue; }          // (the registers' ln-like IS that of the last solve)
                double sn, ln, c, sh; int nd;
                mc.template solve<FAST>(i, j, s_old[j], sn, ln, c, sh, nd);
                if (st.lnl_only) st.l[k] = chi2_logpdf<false>(0.5 * ((double)nd - 1.0) - 1.0, c, st.lgtab[nd], global_tabs());
                else { st.s[k] = sn; st.l[k] = ln; st.c[k] = c; st.sh[k] = sh; }
            }
        }
        if (tid == 0) { atomicMax(st.last_iter, iters); if (st.niter) st.niter[i] = iters; }
        __syncthreads();
    }
}

// ---- rounds: several iterations per record read (round 5; mask-free tame data, the FAST solve) ----
// k_modec_persist re-reads every model record from L2 on every iteration: 580 GB of L2 -> CU traffic on the 2e4 x 1e4 benchmark,
// waves waiting 46 % of their cycles (profiles/r5_v1_pmc_modeC.txt).  But a model's trajectory s_0 -> s_1 -> ... depends on
// nothing but its own record; only the STOP rule couples the models.  So a thread takes one of its models, reads the record ONCE
// and runs R iterations on it, noting per iteration whether its error was above ltol (a bit per iteration); the block then ANDs
// the bits: the reference stops at the first iteration t* at which no model was above ltol.  R is chosen so that t*, if it comes
// in this round at all, is the round's LAST iteration: the errors decay geometrically, the round's last errors give the rate, and
// the round after covers the predicted remainder (the slower of the last two rates, 0.8 of it in the first passes; never too
// long on 880 traced objects of the benchmark and of the tests, docs/modec.md).  A round that turns out too long -- t* before
// its end: the models have moved past the state the reference returns -- hands the object to the IEEE instantiation of
// k_modec_persist, as the interval guard does for an error within rounding of ltol (st.amb): exact, rare, ~2 x the cost.
// Between rounds the scale of every model sits in LDS and its ln-like in the output plane (L2); chi2 and shape of a round's last
// solve are written every round (8 / 16 B per model: which round is the last is known only afterwards).
// Per iteration and pair that is 1 / R record reads (R ~ 4.6 on the benchmark) and no block reduction; no per-thread state arrays:
// <= 128 VGPRs at any model count.
#define FZ_MCR_RMAX 16
// SLDS: the scales between rounds in LDS (8 B per model: M <= FZ_MCP_MAXM, one block per CU beyond ~9 000 models) or in the
// scale plane (any M, several small blocks per CU).  Objects are taken from a queue (st.qhead): iteration counts differ by
// two orders of magnitude from object to object, a fixed share per block would leave the chip waiting for the unluckiest block.
template <class MC, int TPB, bool SLDS>
__global__ __launch_bounds__(TPB) void k_modec_rounds(MC mc, ModeCState st, int64_t Nc, int M, double ltol, int max_iter, int* status) {
    extern __shared__ double s_cur[];                     // SLDS: [M] the scale of every model after the last round
    __shared__ double red[4][TPB / 64];
    __shared__ unsigned redm[2][TPB / 64];
    __shared__ int s_ctl[3];                              // what the block does next (0 stop, 1 go on, 2 hand over, 3 not converged), length of the next round, next object
    __shared__ __attribute__((aligned(16))) double s_logt[256];
    constexpr int BT = MC::NB;
    const int tid = threadIdx.x;
    for (int k = tid; k < 256; k += TPB) s_logt[k] = FZ_LOG_TAB[k];
    const double2* lt = reinterpret_cast<const double2*>(s_logt);
    __syncthreads();
    while (true) {
        if (tid == 0) s_ctl[2] = atomicAdd(st.qhead, 1);
        __syncthreads();
        const int64_t slot = __builtin_amdgcn_readfirstlane(s_ctl[2]);          // (block-uniform, and the compiler is told so)
        if (slot >= Nc) break;
        const int64_t i = st.list ? (int64_t)st.list[slot] : slot;
        const int Mi = mc.sub.nnb ? (int)(mc.sub.nnb[i] < M ? mc.sub.nnb[i] : M) : M;
        double x[BT], v[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) { x[b] = mc.ov.x[i * BT + b]; v[b] = mc.ov.v[i * BT + b]; }
        int iters = 0, R = max_iter < 2 ? 1 : 2, f = 0;       // (iters: thread 0 counts, only it uses the count)
        bool first = true;
#ifdef FZ_MCR_DEBUG
        int dbg_rounds = 0;
#endif
        double h0 = 0.0, h1 = 0.0, h2 = 0.0;              // thread 0: the last three errors E(t) = max_j (lower bound of |dlnl_j|)
        while (true) {
            double elast = 0.0, eprev = 0.0, eprev2 = 0.0, ehlast = 0.0;
            unsigned above = 0, fnan = 0;                   // bit r: this thread had a model above ltol / model 0's error was NaN at the round's iteration r
            for (int j = tid; j < Mi; j += TPB) {
                const int64_t k = i * M + j;
                double rec[MC::REC_W];
                mc.load_rec(mc.sub.nbr ? mc.sub.nbr[i * mc.sub.W + j] : (int64_t)j, rec);
                double s, l, c = 0.0, sh = 0.0;
                if (first) mc.solve_xv(x, v, rec, 1.0, s, l, c, sh, lt);      // the initial pass of pdf.py:171-194
                else { s = SLDS ? s_cur[j] : st.s[k]; l = st.l[k]; }
                for (int r = 0; r < R; ++r) {
                    double sn, ln;
                    mc.solve_xv(x, v, rec, s, sn, ln, c, sh, lt);
                    double ej = fabs(ln - l);
                    if (j == 0 && ej != ej) fnan |= 1u << r;
                    const double dl = 2.9e-14 * (fabs(ln) + fabs(l));         // what the IEEE iterates could differ by (k_modec_persist)
                    const double ehj = ej + dl;
                    ej = fmax(ej - dl, 0.0);                                  // (a NaN becomes 0: dropped, as by the builtin max)
                    if (ej > ltol) above |= 1u << r;
                    if (r == R - 1) { elast = fmax(elast, ej); if (ehj == ehj) ehlast = fmax(ehlast, ehj); }
                    else if (r == R - 2) eprev = fmax(eprev, ej);
                    else if (r == R - 3) eprev2 = fmax(eprev2, ej);
                    s = sn; l = ln;
                }
                if (SLDS) s_cur[j] = s; else st.s[k] = s;
                st.l[k] = l;
                if (st.lnl_only != 1) st.c[k] = c;
                if (!st.lnl_only) st.sh[k] = sh;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                elast = fmax(elast, __shfl_xor(elast, o, 64)); eprev = fmax(eprev, __shfl_xor(eprev, o, 64));
                eprev2 = fmax(eprev2, __shfl_xor(eprev2, o, 64)); ehlast = fmax(ehlast, __shfl_xor(ehlast, o, 64));
                above |= (unsigned)__shfl_xor((int)above, o, 64); fnan |= (unsigned)__shfl_xor((int)fnan, o, 64);
            }
            if ((tid & 63) == 0) {
                red[0][tid >> 6] = elast; red[1][tid >> 6] = eprev; red[2][tid >> 6] = eprev2; red[3][tid >> 6] = ehlast;
                redm[0][tid >> 6] = above; redm[1][tid >> 6] = fnan;
            }
            __syncthreads();
            if (tid == 0) {
                double E = 0.0, E1 = 0.0, E2 = 0.0, EH = 0.0; unsigned ab = 0, fn = 0;
                for (int w = 0; w < TPB / 64; ++w) {
                    E = fmax(E, red[0][w]); E1 = fmax(E1, red[1][w]); E2 = fmax(E2, red[2][w]); EH = fmax(EH, red[3][w]);
                    ab |= redm[0][w]; fn |= redm[1][w];
                }
                const unsigned all = (R >= 32) ? 0xffffffffu : ((1u << R) - 1u);
                const unsigned stop = (~ab | fn) & all;                       // `while lerr > ltol`: the iterations after which the reference stops
                if (R >= 3) { h0 = E2; h1 = E1; h2 = E; } else if (R == 2) { h0 = h2; h1 = E1; h2 = E; } else { h0 = h1; h1 = h2; h2 = E; }
                iters += R;
                int g, Rn = 1;
                if (stop) {
                    const int ts = __builtin_ctz(stop);
                    if (ts != R - 1) g = 2;                                   // the round ran past the stop: the IEEE kernel redoes the object
                    else g = (!((fn >> ts) & 1u) && EH > ltol) ? 2 : 0;       // ltol inside the interval of the error: the IEEE iterates decide
                } else if (iters >= max_iter) g = 3;
                else {
                    g = 1;
                    // remaining iterations if the errors keep their rate: the slower of the last two rates; 0.8 of it in the first passes
                    double n = 1.0;
                    if (h2 < h1 && h2 > 0.0) {
                        double rho = h2 / h1;
                        if (iters >= 3 && h1 < h0) rho = fmax(rho, h1 / h0);
                        n = log(ltol / h2) / log(rho);
                        if (iters < 6) n *= 0.8;
                    }
                    Rn = (n >= (double)FZ_MCR_RMAX) ? FZ_MCR_RMAX : ((n >= 1.0) ? (int)n : 1);      // (NaN: 1)
                    if (st.rfixed > 0) Rn = st.rfixed < FZ_MCR_RMAX ? st.rfixed : FZ_MCR_RMAX;
                    if (Rn > max_iter - iters) Rn = max_iter - iters;
                }
#ifdef FZ_MCR_DEBUG
                if (++dbg_rounds <= 40 || dbg_rounds % 100 == 0 || g != 1)
                    printf("obj %d round %d: R %d iters %d E %.3e E1 %.3e EH %.3e ab %x fn %x stop %x -> g %d Rn %d\n", (int)i, dbg_rounds, R, iters, E, E1, EH, ab, fn, stop, g, Rn);
                if (dbg_rounds > 20000) g = 3;
#endif
                s_ctl[0] = g; s_ctl[1] = Rn;
            }
            __syncthreads();
            f = __builtin_amdgcn_readfirstlane(s_ctl[0]); R = __builtin_amdgcn_readfirstlane(s_ctl[1]);
            first = false;
            __syncthreads();
            if (f != 1) break;
        }
        if (f == 3 && tid == 0) atomicMax(status, 1);
        if (f == 2) {                                       // left to the IEEE instantiation of k_modec_persist
            if (tid == 0) st.amb[atomicAdd(st.namb, 1)] = (int)i;
        } else if (st.lnl_only == 2) {                      // the planes in their final form: every thread finishes the models it wrote itself
            for (int j = tid; j < Mi; j += TPB) {
                const int64_t k = i * M + j;
                st.l[k] = chi2_logpdf<false>(0.5 * ((double)mc.nband - 1.0) - 1.0, st.c[k], st.lgtab[mc.nband], global_tabs());
            }
        } else if (SLDS && !st.lnl_only) {
            for (int j = tid; j < Mi; j += TPB) st.s[i * M + j] = s_cur[j];
        }
        if (tid == 0 && f != 2) { atomicMax(st.last_iter, iters); if (st.niter) st.niter[i] = iters; }
        __syncthreads();
    }
}

// (Round 4 tried the records STREAMED through two LDS buffers by LDS-DMA, tile m + 1 landing while tile m is solved, the previous
//  scales moved from LDS to the state plane: 53.4 ms against 39.5 ms for this kernel on 2e4 x 1e4.  Twelve waves that meet at a
//  barrier per tile hide less latency than twelve independent waves, and the bytes are the same: 0.96 MB of records per object and
//  pass, 19 TB/s of L2 -> CU traffic over the launch, about half of what the eight L2s deliver -- the kernel sits between the
//  L2 and the vector ALU, and only fewer bytes per pair-iteration (records that stay on chip) would move it.)
static __global__ void k_modec_check(ModeCState st, int64_t Nc, double ltol, int iter) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= Nc || (st.ncur && slot >= *st.ncur)) return;
    const int i = st.list ? st.list[slot] : (int)slot;
    const double e = __longlong_as_double((long long)st.err[i]);
    const bool go = !st.firstnan[i] && (e > ltol);       // `while lerr > ltol`
    st.err[i] = 0ull;
    if (st.niter) st.niter[i] = iter;                    // the last check an object sees is the one that stops it
    // FAST solve: ltol between the two bounds of the object's error -- the IEEE iterates could decide the other way
    if (st.amb) {
        const double ehi = __longlong_as_double((long long)st.errhi[i]);
        st.errhi[i] = 0ull;
        if (!st.firstnan[i] && e <= ltol && ehi > ltol && !st.ambflag[i]) { st.ambflag[i] = 1; st.amb[atomicAdd(st.namb, 1)] = i; }
    }
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
