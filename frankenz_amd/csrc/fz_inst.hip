// Per-band-count instantiations of the photometric kernels.  Compiled once per
// FZ_BT in {4, 5, 6, 7, 8, 12, 16, 24, 32} (separate translation units so they build in parallel).
#ifndef FZ_BT
#error "compile with -DFZ_BT=4|5|6|7|8|12|16|24|32"
#endif
#include "fz_ctx.h"
#include "fz_kernels.h"
#include "fz_launch.h"
#include "fz_modec.h"

using namespace fz;

#define FZ_CAT_(a, b) a##b
#define FZ_CAT(a, b) FZ_CAT_(a, b)
#define FZ_NAME(base) FZ_CAT(base, FZ_BT)

// VAR_FAST exists only when BT can equal the real band count (B in 4..8): padded band
// counts always carry mask bits.
#define FZ_EXACT_BT (FZ_BT >= 4 && FZ_BT <= 8)
#if defined(FZ_DEV_FAST)
// development builds (tools/devbuild.sh): the mask-free variant only, no ln-prior instantiations -- a quarter of the compile time
#define FZ_SWITCH_VAR(MODE_, CALL)                                         \
    switch (var) {                                                         \
        case 0: { CALL(FZ_BT, MODE_, 0); } break;                          \
        default: return fail(-1, "FZ_DEV_FAST build: variant %d not compiled", var); \
    }
#elif FZ_EXACT_BT
#define FZ_SWITCH_VAR(MODE_, CALL)                                         \
    switch (var) {                                                         \
        case 0: { CALL(FZ_BT, MODE_, 0); } break;                          \
        case 1: { CALL(FZ_BT, MODE_, 1); } break;                          \
        default: { CALL(FZ_BT, MODE_, 2); } break;                         \
    }
#else
#define FZ_SWITCH_VAR(MODE_, CALL)                                         \
    switch (var) {                                                         \
        case 2: { CALL(FZ_BT, MODE_, 2); } break;                          \
        default: { CALL(FZ_BT, MODE_, 1); } break;                         \
    }
#endif
#define FZ_SWITCH(CALL)                                                    \
    switch (mode) {                                                        \
        case 0: FZ_SWITCH_VAR(0, CALL) break;                              \
        case 1: FZ_SWITCH_VAR(1, CALL) break;                              \
        case 2: FZ_SWITCH_VAR(2, CALL) break;                              \
        default: return fail(-1, "internal: bad likelihood mode %d", mode); \
    }

int FZ_NAME(fz_planes_bt)(fz_ctx* c, int mode, int var, int dim_prior, int64_t n, double* lnl, double* chi2,
                          int64_t* ndim, double* scale, double* serr) {
    const int64_t M = c->M;
    // two adjacent models per thread (16-B stores) when every plane row starts 16-B aligned
    const uintptr_t al = (uintptr_t)lnl | (uintptr_t)chi2 | (uintptr_t)ndim | (uintptr_t)scale | (uintptr_t)serr;
    const char* e_mpt = fz_dbg("FZ_PLANES_MPT");
    const int MPT = (M % 2 == 0 && (al & 15) == 0 && !(e_mpt && atoi(e_mpt) == 1)) ? 2 : 1;
    const int64_t mblocks = (M + 256 * MPT - 1) / (256 * MPT);
    // objects per block: 256 when the grid still holds >= 4 blocks per CU (+11 % at 1e5 x 1e4), else 16
    const bool big = ((n + 255) / 256) * mblocks >= 4 * (int64_t)c->cu_count;
    const int TO = big ? 256 : 16;
    dim3 grid((unsigned)((n + TO - 1) / TO), (unsigned)mblocks);
    Timer t(c, &c->tm.ms_planes, &c->tm.n_planes);
#define FZ_CALL_PLANES(BT_, MODE_, VAR_)                                                                  \
    PhotSrc<BT_, MODE_, VAR_> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, MODE_, dim_prior); \
    using PH_ = PhotSrc<BT_, MODE_, VAR_>;                                                                \
    auto kern = MPT == 2 ? (dim_prior ? (big ? k_planes<PH_, 256, 1, 2> : k_planes<PH_, 16, 1, 2>)         \
                                      : (big ? k_planes<PH_, 256, 0, 2> : k_planes<PH_, 16, 0, 2>))        \
                         : (dim_prior ? (big ? k_planes<PH_, 256, 1, 1> : k_planes<PH_, 16, 1, 1>)         \
                                      : (big ? k_planes<PH_, 256, 0, 1> : k_planes<PH_, 16, 0, 1>));       \
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, c->stream, ph, n, M, lnl, chi2, ndim, scale, serr);
    FZ_SWITCH(FZ_CALL_PLANES)
    HIPCHK(hipGetLastError());
    return 0;
}

int FZ_NAME(fz_fitpredict_bt)(fz_ctx* c, int mode, int var, int dim_prior, int64_t n, const fz_kde_opts* ko, double* lmap,
                              double* levid, double* pdfs) {
    const int64_t M = c->M;
#if defined(FZ_DEV_FAST)
#define FZ_CALL_FUSED(BT_, MODE_, VAR_)                                                                   \
    PhotSrc<BT_, MODE_, VAR_> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, MODE_, dim_prior); \
    return fz_launch_fitpredict(c, ph, n, M, ko, lmap, levid, pdfs);
#else
#define FZ_CALL_FUSED(BT_, MODE_, VAR_)                                                                   \
    if (c->prior.tab) {                                                                                    \
        PhotSrc<BT_, MODE_, VAR_, true> ph; ph.mv = model_view(c); ph.ov = obj_view(c);                    \
        ph.lp = like_params(c, MODE_, dim_prior); ph.pv = c->prior;                                        \
        return fz_launch_fitpredict(c, ph, n, M, ko, lmap, levid, pdfs);                                   \
    }                                                                                                      \
    PhotSrc<BT_, MODE_, VAR_> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, MODE_, dim_prior); \
    return fz_launch_fitpredict(c, ph, n, M, ko, lmap, levid, pdfs);
#endif
#if FZ_EXACT_BT
    if (var == VAR_SEG) {
        // masked models / unobserved object bands against per-model errors: k_hist on the segmented model layout (mask-free arithmetic,
        // the handed-back objects are swept by the masked variant); +1: not applicable, the caller takes the masked route
        int r = 1;
        switch (mode) {
#define FZ_CALL_SEG(MODE_)                                                                                                       \
            case MODE_: { PhotSrc<FZ_BT, MODE_, VAR_FAST> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, MODE_, dim_prior); \
                          PhotSrc<FZ_BT, MODE_, VAR_MASKED> pm; pm.mv = ph.mv; pm.ov = ph.ov; pm.lp = ph.lp;                       \
                          r = fz_launch_hist_seg(c, ph, pm, n, M, ko, lmap, levid, pdfs); } break;
            FZ_CALL_SEG(0) FZ_CALL_SEG(1) FZ_CALL_SEG(2)
#undef FZ_CALL_SEG
            default: break;
        }
        return r < 0 ? r : (r == 0 ? 0 : 1);
    }
#else
    if (var == VAR_SEG) return 1;
#endif
#if !defined(FZ_DEV_FAST)
    // (also: no dimensionality prior on mask-free data in modes Ai / B -- the power-0 form of the same kernel, pdf.py:94-98)
    const bool nodp_fast = !dim_prior && (var == VAR_FAST || var == VAR_PAD) && (mode == 1 || mode == 2) && !c->prior.tab;
    if (var == VAR_OBJMASK || nodp_fast) {
        // objects with unobserved bands against unmasked models, modes Ai / B: k_hist with per-object band counts on the mask-free
        // arithmetic (the handed-back objects are swept by the masked variant); +1: not applicable, the caller takes the masked route
        int r = 1;
        switch (mode) {
            case 1: { PhotSrc<FZ_BT, 1, VAR_FAST> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, 1, dim_prior);
                      PhotSrc<FZ_BT, 1, VAR_MASKED> pm; pm.mv = ph.mv; pm.ov = ph.ov; pm.lp = ph.lp;
                      r = fz_launch_hist_objmask(c, ph, pm, n, M, ko, lmap, levid, pdfs); } break;
            case 2: { PhotSrc<FZ_BT, 2, VAR_FAST> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, 2, dim_prior);
                      PhotSrc<FZ_BT, 2, VAR_MASKED> pm; pm.mv = ph.mv; pm.ov = ph.ov; pm.lp = ph.lp;
                      r = fz_launch_hist_objmask(c, ph, pm, n, M, ko, lmap, levid, pdfs); } break;
            default: break;
        }
        if (var == VAR_OBJMASK) return r < 0 ? r : (r == 0 ? 0 : 1);
        if (r <= 0) return r;
    }
#else
    if (var == VAR_OBJMASK) return 1;
#endif
#if !FZ_EXACT_BT
    // 9-32 bands without a masked REAL band on tame data: the one-pass histogram kernel in its mask-free form (fz_hist.h: the pad
    // bands up to 16 / 32 are zeros, the power of chi2 follows the real band count); every other case of these band counts --
    // masks, a prior, the KDE forms k_hist does not take -- runs the masked variants
    if ((var == VAR_FAST || var == VAR_PAD) && !c->prior.tab) {
        int r = 1;
        switch (mode) {
            case 0: { PhotSrc<FZ_BT, 0, VAR_FAST> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, 0, dim_prior);
                      r = fz_launch_hist_only(c, ph, n, M, ko, lmap, levid, pdfs); } break;
            case 1: { PhotSrc<FZ_BT, 1, VAR_FAST> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, 1, dim_prior);
                      r = fz_launch_hist_only(c, ph, n, M, ko, lmap, levid, pdfs); } break;
            case 2: { PhotSrc<FZ_BT, 2, VAR_FAST> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, 2, dim_prior);
                      r = fz_launch_hist_only(c, ph, n, M, ko, lmap, levid, pdfs); } break;
            default: break;
        }
        if (r <= 0) return r;
    }
#endif
    FZ_SWITCH(FZ_CALL_FUSED)
    return 0;
}

// ---------------------------------------------------------------------------
// mode C driver on a prepared chunk: leaves converged state in c->d_mc[*]
// ---------------------------------------------------------------------------
template <int BT, bool MASKED>
static int run_modec(fz_ctx* c, int64_t n, const fz_like_opts* o, const SubsetView& sub, bool tame) {
    const int64_t M = sub.nbr ? sub.W : c->M;
    const size_t pl = (size_t)n * M * 8;
    for (int k = 0; k < 4; ++k) FZCHK(c->d_mc[k].ensure(pl));
    FZCHK(c->d_mcerr.ensure(2 * n * 8)); FZCHK(c->d_mcfn.ensure(n * 4)); FZCHK(c->d_mcact.ensure(4 * n * 4)); FZCHK(c->d_mccnt.ensure(64));
    if (n > 0x7fffffffLL) return fail(-1, "mode C chunk too large");
    ModeCState st; st.s = c->d_mc[0].as<double>(); st.l = c->d_mc[1].as<double>(); st.c = c->d_mc[2].as<double>();
    st.sh = c->d_mc[3].as<double>(); st.err = c->d_mcerr.as<unsigned long long>(); st.errhi = st.err + n; st.firstnan = c->d_mcfn.as<int>();
    st.lnl_only = 0; st.lgtab = c->d_lgB.as<double>();
    st.qhead = nullptr; st.rfixed = fz_dbg("FZ_MODEC_RFIXED") ? atoi(fz_dbg("FZ_MODEC_RFIXED")) : 0;
    FZCHK(c->d_mcniter.ensure(n * 4)); st.niter = c->d_mcniter.as<int>(); c->mc_niter_n = n;
    HIPCHK(hipMemsetAsync(st.niter, 0, n * 4, c->stream));
    // Active-object lists and their lengths live on the device and alternate between two slots; the host
    // queues FZ_MODEC_BURST iterations (step + stop rule, launched for the object count it last saw: blocks
    // of objects that stopped since exit at once) before it looks at the count again, so the loop is not
    // paced by one host round trip per iteration (a quarter of the time before).
    int* lists[2] = {c->d_mcact.as<int>(), c->d_mcact.as<int>() + n};
    int* counts = c->d_mccnt.as<int>();                  // [0], [1]: list lengths; [2]: last iteration that left objects active; [3]: ambiguous objects
    HIPCHK(hipMemsetAsync(counts, 0, 16, c->stream));
    HIPCHK(hipMemsetAsync(st.err, 0, 2 * n * 8, c->stream));
    HIPCHK(hipMemsetAsync(st.firstnan, 0, n * 4, c->stream));
    // the reciprocal-based solve for mask-free tame data (fz_modec.h); FZ_MODEC_IEEE=1 keeps the IEEE divisions throughout
    const bool fast = !MASKED && tame && !fz_dbg("FZ_MODEC_IEEE");
    st.amb = fast ? c->d_mcact.as<int>() + 2 * n : nullptr; st.namb = counts + 3; st.ambflag = fast ? c->d_mcact.as<int>() + 3 * n : nullptr;
    if (fast) HIPCHK(hipMemsetAsync(st.ambflag, 0, n * 4, c->stream));
    ModeC<BT, MASKED> mc; mc.mv = model_view(c); mc.ov = obj_view(c); mc.nband = c->B; mc.sub = sub;
    const int64_t tiles = (M + 255) / 256;
    if (n * tiles > 0x7fffffffLL) return fail(-1, "mode C chunk too large");
    const int max_iter = o->max_iter > 0 ? o->max_iter : 10000;
    const int burst = fz_dbg("FZ_MODEC_BURST") ? std::max(1, atoi(fz_dbg("FZ_MODEC_BURST"))) : 8;
    Timer t(c, &c->tm.ms_modec, &c->tm.n_modec);
    int it_max = 0;
    const int want_lnl_only = c->mc_lnl_only;
    c->mc_lnl_only = 0;                                  // honoured below by the one-block-per-object path only (and not for neighbour subsets)
    // one run of the fixed point over `n0` objects (all of the chunk, or the listed ones), to convergence
    auto iterate = [&](const int* list0, int n0, bool fst) -> int {
        auto step = [&](const ModeCState& s2, int nobj, int init) {
            if (fst) hipLaunchKernelGGL((k_modec_step<ModeC<BT, MASKED>, true>), dim3((unsigned)(nobj * tiles)), dim3(256), 0, c->stream, mc, s2, nobj, M, init);
            else hipLaunchKernelGGL((k_modec_step<ModeC<BT, MASKED>, false>), dim3((unsigned)(nobj * tiles)), dim3(256), 0, c->stream, mc, s2, nobj, M, init);
        };
        ModeCState s2 = st;
        if (!fst) s2.amb = nullptr;
        s2.list = list0; s2.ncur = nullptr; s2.list_next = lists[0]; s2.nactive = counts; s2.last_iter = counts + 2;
        HIPCHK(hipMemsetAsync(counts, 0, 8, c->stream));
        step(s2, n0, 1);
        // iteration 0 runs on the initial set; iteration t >= 1 on list t & 1 ... written by the check of t - 1
        int it = 0, nact = n0;        // nact: an upper bound of the active count
        bool first = true;
        while (nact > 0) {
            if (it >= max_iter)
                return fail(-7, "mode C (free_scale with model errors): %d objects not converged after %d iterations "
                                "(the reference loop at pdf.py:199 would not terminate)", nact, max_iter);
            for (int b = 0; b < burst && it < max_iter; ++b, ++it) {
                const int cur = it & 1, nxt = cur ^ 1;
                s2.list = first ? list0 : lists[cur]; s2.ncur = first ? nullptr : counts + cur;
                s2.list_next = lists[nxt]; s2.nactive = counts + nxt;
                HIPCHK(hipMemsetAsync(counts + nxt, 0, 4, c->stream));
                step(s2, nact, 0);
                hipLaunchKernelGGL(k_modec_check, dim3((unsigned)((nact + 255) / 256)), dim3(256), 0, c->stream, s2, nact, o->ltol, it + 1);
                first = false;
            }
            HIPCHK(hipMemcpyAsync(&nact, counts + (it & 1), 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        return 0;
    };
    // beyond FZ_MCP_MAXM models (mask-free tame data): k_modec_rounds with the scales in the scale plane instead of LDS -- no limit on M;
    // objects it hands back are redone by the state-plane kernels with IEEE divisions
    {
        const char* ev = fz_dbg("FZ_MODEC_ROUNDS");
        if (fast && M > FZ_MCP_MAXM && !fz_dbg("FZ_MODEC_PLANES") && !(ev && ev[0] == '0')) {
            using MCT = ModeC<BT, MASKED>;
            if constexpr (!MASKED) {
                HIPCHK(hipMemsetAsync(counts + 4, 0, 4, c->stream));
                int bpc = 1;
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)k_modec_rounds<MCT, 512, false>, 512, 0));
                const int64_t blocks = std::min<int64_t>(n, (int64_t)std::max(1, bpc) * c->cu_count);
                ModeCState s2 = st; s2.last_iter = counts + 2; s2.list = nullptr; s2.namb = counts + 3; s2.qhead = counts + 4;
                hipLaunchKernelGGL((k_modec_rounds<MCT, 512, false>), dim3((unsigned)blocks), dim3(512), 0, c->stream, mc, s2, n, (int)M, o->ltol,
                                   max_iter, counts + 1);
                int res[3] = {0, 0, 0};
                HIPCHK(hipMemcpyAsync(res, counts + 1, 12, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if (res[0]) return fail(-7, "mode C (free_scale with model errors): objects not converged after %d iterations "
                                            "(the reference loop at pdf.py:199 would not terminate)", max_iter);
                int slow = res[1];
                if (res[2] > 0) {
                    HIPCHK(hipMemsetAsync(st.err, 0, n * 8, c->stream));
                    HIPCHK(hipMemsetAsync(counts + 2, 0, 4, c->stream));
                    FZCHK(iterate(st.amb, res[2], false));
                    int it2 = 0;
                    HIPCHK(hipMemcpyAsync(&it2, counts + 2, 4, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(hipStreamSynchronize(c->stream));
                    slow = std::max(slow, it2 + 1);
                }
                HIPCHK(hipGetLastError());
                c->mc_info[0] += res[2]; c->mc_info[1] = std::max<int64_t>(c->mc_info[1], slow); c->mc_info[2] = 3; c->mc_info[3] = 512;
                c->tm.n_modec += slow;
                return 0;
            }
        }
    }
    if (M <= FZ_MCP_MAXM && !fz_dbg("FZ_MODEC_PLANES")) {
        if (want_lnl_only && !sub.nbr && !fz_dbg("FZ_MODEC_FINAL")) { st.lnl_only = o->dim_prior ? 2 : 1; c->mc_lnl_only = 1; }
        // the whole fixed point of an object inside one block (fz_modec.h, k_modec_persist): no state planes through HBM
        auto launch = [&](auto kern, int T, size_t lds, const int* list, int64_t nobj) -> int {
            HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int bpc = 1;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, T, lds));
            const int64_t blocks = std::min<int64_t>(nobj, (int64_t)std::max(1, bpc) * c->cu_count);
            ModeCState s2 = st; s2.last_iter = counts + 2; s2.list = list; s2.namb = counts + 3; s2.qhead = counts + 4;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(T), lds, c->stream, mc, s2, nobj, (int)M, o->ltol, max_iter, counts + 1);
            return 0;
        };
        auto run = [&](auto fastc, const int* list, int64_t nobj) -> int {
            constexpr bool F = decltype(fastc)::value;
            using MCT = ModeC<BT, MASKED>;
            const size_t lds = (size_t)M * 8;                    // the previous scale of every model
            if constexpr (F && !MASKED) {
                // several iterations per record read (k_modec_rounds); FZ_MODEC_ROUNDS=0: one iteration per read (k_modec_persist)
                const char* ev = fz_dbg("FZ_MODEC_ROUNDS");
                if (!(ev && ev[0] == '0')) {
                    c->mc_info[2] = 3;
                    HIPCHK(hipMemsetAsync(counts + 4, 0, 4, c->stream));
                    if (M <= 1024) { c->mc_info[3] = 256; return launch(k_modec_rounds<MCT, 256, true>, 256, lds, list, nobj); }
                    if (M <= 4096) { c->mc_info[3] = 512; return launch(k_modec_rounds<MCT, 512, true>, 512, lds, list, nobj); }
                    c->mc_info[3] = 1024;
                    return launch(k_modec_rounds<MCT, 1024, true>, 1024, lds, list, nobj);
                }
            }
            if (M <= 1024) { c->mc_info[3] = 1024; return launch(k_modec_persist<MCT, F, 1024, 1>, 1024, lds, list, nobj); }
            if (M <= 4096) { c->mc_info[3] = 1024; return launch(k_modec_persist<MCT, F, 1024, 4>, 1024, lds, list, nobj); }
            if (M <= 768 * 14) { c->mc_info[3] = 768; return launch(k_modec_persist<MCT, F, 768, 14>, 768, lds, list, nobj); }
            c->mc_info[3] = 512;
            return launch(k_modec_persist<MCT, F, 512, 32>, 512, lds, list, nobj);
        };
        c->mc_info[2] = 1;
        if (fast) FZCHK(run(std::true_type{}, nullptr, n)); else FZCHK(run(std::false_type{}, nullptr, n));
        const int64_t kind = c->mc_info[2], tpb = c->mc_info[3];            // (the re-run below is not what the call is reported as)
        int res[3] = {0, 0, 0};                              // status, slowest object's iterations, ambiguous objects
        HIPCHK(hipMemcpyAsync(res, counts + 1, 12, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const int namb = res[2];
        if (fast && namb > 0 && !res[0]) {
            FZCHK(run(std::false_type{}, st.amb, namb));
            HIPCHK(hipMemcpyAsync(res, counts + 1, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            res[2] = namb;
        }
        HIPCHK(hipGetLastError());
        if (res[0]) return fail(-7, "mode C (free_scale with model errors): objects not converged after %d iterations "
                                    "(the reference loop at pdf.py:199 would not terminate)", max_iter);
        c->mc_info[0] += fast ? res[2] : 0; c->mc_info[1] = std::max<int64_t>(c->mc_info[1], res[1]); c->mc_info[2] = kind; c->mc_info[3] = tpb;
        c->tm.n_modec += res[1];         // iterations of the slowest object (the two timed scopes add the other two counts the bench subtracts)
        return 0;
    }
    FZCHK(iterate(nullptr, (int)n, fast));
    int namb = 0;
    if (fast) {
        HIPCHK(hipMemcpyAsync(&namb, counts + 3, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (namb > 0) {
            // objects whose error came within rounding of ltol: once more from the start, IEEE divisions (their state planes are
            // simply overwritten; the list lives behind the two active lists)
            HIPCHK(hipMemsetAsync(st.err, 0, n * 8, c->stream));
            FZCHK(iterate(st.amb, namb, false));
        }
    }
    HIPCHK(hipMemcpyAsync(&it_max, counts + 2, 4, hipMemcpyDeviceToHost, c->stream));      // iterations the slowest object took, minus one
    HIPCHK(hipStreamSynchronize(c->stream));
    ++it_max;
    c->mc_info[0] += namb; c->mc_info[1] = std::max<int64_t>(c->mc_info[1], it_max); c->mc_info[2] = 2; c->mc_info[3] = 0;
    c->tm.n_modec += it_max;         // iterations of the slowest object of the chunk (+1 per timed scope: the initial pass)
    HIPCHK(hipGetLastError());
    return 0;
}

int FZ_NAME(fz_modec_bt)(fz_ctx* c, int var, int64_t n, const fz_like_opts* o, const int64_t* nbr, const int64_t* nnb, int W) {
    SubsetView sub; sub.nbr = nbr; sub.nnb = nnb; sub.W = W;
#if FZ_EXACT_BT
    return var == VAR_FAST ? run_modec<FZ_BT, false>(c, n, o, sub, true) : run_modec<FZ_BT, true>(c, n, o, sub, false);
#else
    (void)var;
    return run_modec<FZ_BT, true>(c, n, o, sub, false);
#endif
}

// ---------------------------------------------------------------------------
// k-NN: brute-force search over the K feature sets and the subset likelihood/PDF
// ---------------------------------------------------------------------------
int FZ_NAME(fz_knnquery_bt)(fz_ctx* c, const double* q, int64_t n, int k, double bound2, int64_t* idx, int pnorm) {
    constexpr int TQ = (FZ_BT <= 5) ? 4 : (FZ_BT <= 8 ? 2 : 1);     // queries per wave (register budget)
    const bool screen = pnorm == 2 && !fz_dbg("FZ_KNN_FP64");
    const int64_t per = (int64_t)(screen ? (TQ >= 2 ? TQ : 2) : TQ) * 4;
    dim3 grid((unsigned)((n + per - 1) / per), (unsigned)c->knn_K);
    Timer t(c, &c->tm.ms_knn, &c->tm.n_knn);
    if (screen) {
        // Euclidean norm: packed-fp32 screen + exact fp64 re-check (same result, ~2x fewer fp64 instructions)
        constexpr int TQ2 = TQ >= 2 ? TQ : 2;                          // screened in pairs (8 per wave measured slower: registers)
        hipLaunchKernelGGL((k_knn_query32<FZ_BT, TQ2>), grid, dim3(256), 0, c->stream, c->d_trees.as<float>(), c->Mp, (int)c->knn_M, q, n,
                           c->knn_F, k, bound2, idx, c->knn_K);
    } else {
        hipLaunchKernelGGL((k_knn_query<FZ_BT, TQ>), grid, dim3(256), 0, c->stream, c->d_trees.as<float>(), c->Mp, (int)c->knn_M, q, n,
                           c->knn_F, k, bound2, idx, c->knn_K, pnorm);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int FZ_NAME(fz_knnsubset_bt)(fz_ctx* c, int mode, int var, int dim_prior, int64_t n, const int64_t* idx, int W,
                             const fz_kde_opts* ko, const KnnOut* out, int* errflag) {
    KdeView kv;
    memset(&kv, 0, sizeof kv);
    if (out->pdfs) FZCHK(fz_kde_view(c, kv));
    else kv.acc_stride = 8;
    FZCHK(c->d_kv.ensure(sizeof(KdeView)));
    HIPCHK(hipMemcpyAsync(c->d_kv.p, &kv, sizeof(KdeView), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t per_wave = fz_knn_subset_lds_doubles(kv.acc_stride, W);      // list | hash table, then ln-likelihoods + accumulation row (fz_knn.h)
    int wpb = 4;
    while (wpb > 1 && per_wave * 8 * wpb > 53 * 1024) wpb >>= 1;             // (three blocks per CU)
    const size_t lds = per_wave * 8 * wpb;
    if (lds > 160 * 1024) return fail(-5, "k-NN PDF grid too large for LDS");
    Timer t(c, &c->tm.ms_knn, &c->tm.n_knn);
#define FZ_CALL_SUBSET(BT_, MODE_, VAR_)                                                                  \
    PhotSrc<BT_, MODE_, VAR_> ph; ph.mv = model_view(c); ph.ov = obj_view(c); ph.lp = like_params(c, MODE_, dim_prior); \
    auto kern = k_knn_subset<PhotSrc<BT_, MODE_, VAR_>>;                                                  \
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + wpb - 1) / wpb)), dim3(wpb * 64), lds, c->stream, ph,    \
                       c->d_kv.as<KdeView>(), kv.acc_stride, n, (int)c->M, idx, W, mode == 2 ? 1 : 0, ko->wt_thresh, \
                       ko->normalize, *out, errflag);
    FZ_SWITCH(FZ_CALL_SUBSET)
    HIPCHK(hipGetLastError());
    return 0;
}
