// Launch geometry of the softmax / KDE kernels (templates over the ln-weight source).
#pragma once
#include "fz_ctx.h"
#include "fz_kernels.h"
#include "fz_nolist.h"
#include "fz_hist.h"
#include "fz_plane.h"

inline int fz_kde_view(fz_ctx* c, fz::KdeView& kv) {
    using namespace fz;
    if (c->label_mode == 0) return fail(-1, "labels have not been uploaded");
    if (c->label_M != c->M)
        return fail(-1, "labels (%lld) do not match the model count (%lld)", (long long)c->label_M, (long long)c->M);
    memset(&kv, 0, sizeof kv);
    kv.G = c->G;
    kv.norm = c->d_norm.as<double>();
    if (c->label_mode == 1) {
        kv.pos = c->d_pos.as<int32_t>(); kv.cls = c->d_cls.as<int32_t>();
        kv.widths = c->d_widths.as<int64_t>(); kv.offsets = c->d_offsets.as<int64_t>(); kv.kern = c->d_kern.as<double>();
        kv.w0 = c->w0; kv.koff0 = c->h_offsets[c->cls0];
        kv.kmode = c->single_cls ? KDE_HIST : KDE_DICT;
        kv.normtab = c->single_cls ? c->d_normtab.as<double>() : nullptr;
        kv.acc_stride = (int)(kv.kmode == KDE_HIST ? c->G + 2 * c->w0 : c->G);
    } else {
        kv.ly = c->d_ly.as<double>(); kv.lstd = c->d_lstd.as<double>(); kv.lo = c->d_lo.as<int32_t>(); kv.hi = c->d_hi.as<int32_t>();
        kv.grid = c->d_grid.as<double>();
        kv.gstep = fz_dbg("FZ_GRID_RECUR") && atoi(fz_dbg("FZ_GRID_RECUR")) == 0 ? 0.0 : c->grid_step;
        kv.lrec = c->d_lrec.as<double>();
        kv.kmode = KDE_GRID; kv.acc_stride = (int)c->G;
    }
    kv.lane_window = fz_dbg("FZ_LANE_WINDOW") ? atoi(fz_dbg("FZ_LANE_WINDOW")) : FZ_LANE_WINDOW;
    if ((size_t)kv.acc_stride * 8 > 160 * 1024)
        return fail(-5, "PDF grid of %lld points needs %zu B of LDS per object (> 160 KiB)", (long long)kv.G,
                    (size_t)kv.acc_stride * 8);
    return 0;
}

template <class SRC>
int fz_launch_stats(fz_ctx* c, const SRC& src, int64_t n, int64_t M, int linear, double* lmap, double* levid) {
    constexpr int TW = 4, WPB = 4;
    const int64_t per = TW * WPB;
    Timer t(c, &c->tm.ms_stats, &c->tm.n_stats);
    hipLaunchKernelGGL((fz::k_stats<SRC, TW>), dim3((unsigned)((n + per - 1) / per)), dim3(WPB * 64), 0, c->stream, src, n, M,
                       linear, lmap, levid);
    HIPCHK(hipGetLastError());
    return 0;
}

template <class SRC, int TW>
int fz_launch_kde_tw(fz_ctx* c, const SRC& src, const fz::KdeView& kv, int wpb, int64_t n, int64_t M, int linear,
                     const double* lmap, const double* levid, const fz_kde_opts* ko, double* pdfs) {
    const int64_t per = (int64_t)TW * wpb;
    const size_t lds = (size_t)wpb * TW * kv.acc_stride * 8;
    auto kern = fz::k_kde<SRC, TW>;
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Timer t(c, &c->tm.ms_kde, &c->tm.n_kde);
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + per - 1) / per)), dim3(wpb * 64), lds, c->stream, src, kv, n, M, linear,
                       lmap, levid, ko->wt_thresh, ko->normalize, pdfs);
    HIPCHK(hipGetLastError());
    return 0;
}

template <class SRC>
int fz_launch_kde(fz_ctx* c, const SRC& src, int64_t n, int64_t M, int linear, const double* lmap, const double* levid,
                  const fz_kde_opts* ko, double* pdfs) {
    fz::KdeView kv;
    FZCHK(fz_kde_view(c, kv));
    // objects per wave / waves per block from the LDS each object's accumulator needs
    const size_t per_obj = (size_t)kv.acc_stride * 8, budget = 160 * 1024;
    if (per_obj * 8 <= 53 * 1024) return fz_launch_kde_tw<SRC, 2>(c, src, kv, 4, n, M, linear, lmap, levid, ko, pdfs);
    if (per_obj * 4 <= budget) return fz_launch_kde_tw<SRC, 1>(c, src, kv, 4, n, M, linear, lmap, levid, ko, pdfs);
    if (per_obj * 2 <= budget) return fz_launch_kde_tw<SRC, 1>(c, src, kv, 2, n, M, linear, lmap, levid, ko, pdfs);
    return fz_launch_kde_tw<SRC, 1>(c, src, kv, 1, n, M, linear, lmap, levid, ko, pdfs);
}

// register-resident rows (fz_plane.h); +1 = not applicable
template <int NW, int E2>
int fz_launch_plane_rows_g(fz_ctx* c, const double* plane, const fz::KdeView& kv, int64_t n, int64_t M, const fz_kde_opts* ko,
                           double* lmap, double* levid, double* pdfs) {
    auto kern = fz::k_plane_rows<NW, E2>;
    constexpr size_t NT = (size_t)NW * 64;
    // exp table | three histogram rows + 1 / mass | exchange words (two parities) | flags, tie counts | label indices of the lanes'
    // columns | parked ties | the kernel taps as matrix operands
    const size_t lds = ((size_t)FZ_HEXP_K + 4 * (size_t)kv.acc_stride + 6 * NW + 2) * 8 + (2 * NW + 2) * 4 + NT * E2 * 4 + NT * 12 +
                       (size_t)fz::plane_conv_ksteps(2 * kv.w0) * 64 * 8;
    if (lds > 160 * 1024) return 1;
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 1;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, NW * 64, lds));
    const int64_t blocks = std::min<int64_t>(n, (int64_t)std::max(1, bpc) * c->cu_count);
    FZCHK(c->d_kv.ensure(sizeof(fz::KdeView)));
    HIPCHK(hipMemcpyAsync(c->d_kv.p, &kv, sizeof(fz::KdeView), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));          // kv is a stack object
    Timer t(c, &c->tm.ms_fused, &c->tm.n_fused);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), lds, c->stream, plane, M, c->d_kv.as<fz::KdeView>(), kv.acc_stride, n,
                       (int)M, ko->wt_thresh, ko->normalize, lmap, levid, pdfs);
    HIPCHK(hipGetLastError());
#ifdef FZ_PLANE_STATS
    {
        unsigned long long h[16], z[16] = {0};
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(fz::fz_plstats), sizeof h);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fz::fz_plstats), z, sizeof z);
        const double wr = h[1] ? (double)h[1] : 1.0;                 // wave-rows
        fprintf(stderr, "PLSTATS k_plane_rows<%d,%d>: %llu waves, %.1f rows each | cycles per wave and row: wait for the row %.0f | maximum + post %.0f | "
                        "barrier %.0f | clear + previous row's sums / ties %.0f | weigh %.0f | request + sums + post %.0f | epilogue (all waves) %.0f "
                        "(waves with outputs: %.0f)\n", NW, E2, h[0], wr / (double)(h[0] ? h[0] : 1), h[2] / wr, h[3] / wr, h[4] / wr, h[5] / wr, h[6] / wr,
                h[7] / wr, h[8] / wr, h[9] ? h[10] / (wr * (double)h[9] / (double)h[0]) : 0.0);
    }
#endif
    c->last_form = "k_plane_rows";
    return 0;
}
inline int fz_launch_plane_rows(fz_ctx* c, const double* plane, const fz::KdeView& kv, int64_t n, int64_t M, const fz_kde_opts* ko,
                                double* lmap, double* levid, double* pdfs) {
    const int64_t G = kv.G, w2 = 2 * (int64_t)kv.w0;
    if (w2 >= 128 || G + w2 > 65535 || G + w2 > kv.acc_stride || G > 8 * 128) return 1;
    // shapes: 5 120 entries (8 waves x 5 register pairs), 10 240 (8 x 10), 20 480 (16 x 10); a row must fill 80 % of its shape
    // (a 15 000-entry row on the 20 480 shape is slower than k_plane_fused: 2.56 vs 2.40 ms per 66 000 rows)
    int nw = 0, e2 = 0;
    if (M <= 5120) { nw = 8; e2 = 5; }
    else if (M <= 10240) { nw = 8; e2 = 10; }
    else { nw = 16; e2 = 10; }
    const int64_t capn = (int64_t)nw * 64 * 2 * e2;
    if (M > capn || M * 5 < capn * 4) return 1;
    if (nw == 8 && e2 == 5) return fz_launch_plane_rows_g<8, 5>(c, plane, kv, n, M, ko, lmap, levid, pdfs);
    if (nw == 8 && e2 == 10) return fz_launch_plane_rows_g<8, 10>(c, plane, kv, n, M, ko, lmap, levid, pdfs);
    if (nw == 16 && e2 == 10) return fz_launch_plane_rows_g<16, 10>(c, plane, kv, n, M, ko, lmap, levid, pdfs);
    return 1;
}

// predict from a stored ln-weight plane: one pass (k_plane_fused) when its candidate lists fit the
// workspace budget, else / for linear weights / FZ_PLANE_TWOPASS=1 the two-pass kernels
inline int fz_launch_plane_predict(fz_ctx* c, const double* plane, int64_t n, int64_t M, int linear,
                                   const fz_kde_opts* ko, double* lmap, double* levid, double* pdfs) {
    using namespace fz;
    PlaneSrc ps; ps.p = plane; ps.ld = M;
    KdeView kv;
    FZCHK(fz_kde_view(c, kv));
    constexpr int NW = 8;          // 8 waves share one LDS copy of the log / exp tables: two blocks per CU at G = 701
    const size_t lds = ((size_t)FZ_TABS_DOUBLES + (size_t)NW * kv.acc_stride) * 8;
    const bool vec2 = (M % 2 == 0) && (((uintptr_t)plane & 15) == 0);
    const bool ho = kv.kmode == KDE_HIST;              // single-kernel label sets: the instantiation without the window code
    // the weights below wt_thresh of the best in fp32 unless the caller asked for the all-fp64 logsumexp (or thresholds nothing)
    const bool x32 = !ko->exact_evidence && !c->exact_evidence && !fz_dbg("FZ_EXACT_EVIDENCE") && ko->wt_thresh > 0.0;
    auto kern = x32 ? (vec2 ? (ho ? k_plane_fused<NW, 2, true, true> : k_plane_fused<NW, 2, false, true>)
                            : (ho ? k_plane_fused<NW, 1, true, true> : k_plane_fused<NW, 1, false, true>))
                    : (vec2 ? (ho ? k_plane_fused<NW, 2, true, false> : k_plane_fused<NW, 2, false, false>)
                            : (ho ? k_plane_fused<NW, 1, true, false> : k_plane_fused<NW, 1, false, false>));
    // rows that fit one block's registers: exact maximum first, then fp64 weights straight into the LDS histogram (fz_plane.h)
    if (!linear && !c->force_twopass && !fz_dbg("FZ_PLANE_TWOPASS") && vec2 && ho && kv.normtab && ko->wt_thresh >= 0.0 &&
        (!fz_dbg("FZ_PLANE_ROWS") || atoi(fz_dbg("FZ_PLANE_ROWS")) != 0)) {
        const int r = fz_launch_plane_rows(c, plane, kv, n, M, ko, lmap, levid, pdfs);
        if (r <= 0) return r;
    }
    int64_t blocks = 0;
    if (!linear && !c->force_twopass && !fz_dbg("FZ_PLANE_TWOPASS") && lds <= 160 * 1024 && M < ((int64_t)1 << 31)) {
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int bpc = 1;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, NW * 64, lds));
        const int64_t need = (n + NW - 1) / NW;
        const int64_t fit = (int64_t)(c->ws_limit / ((size_t)M * sizeof(Cand) * NW));
        blocks = std::min<int64_t>(need, (int64_t)std::max(1, bpc) * c->cu_count);
        if (fit < blocks) blocks = (fit >= c->cu_count) ? (fit / c->cu_count) * c->cu_count : 0;   // whole CUs or not at all
        if (blocks > 0 && c->d_cand.ensure((size_t)blocks * NW * M * sizeof(Cand)) != 0) blocks = 0;
    }
    if (blocks <= 0) {
        c->last_form = "k_stats + k_kde";
        FZCHK(fz_launch_stats(c, ps, n, M, linear, lmap, levid));
        return fz_launch_kde(c, ps, n, M, linear, lmap, levid, ko, pdfs);
    }
    FZCHK(c->d_kv.ensure(sizeof(KdeView)));
    HIPCHK(hipMemcpyAsync(c->d_kv.p, &kv, sizeof(KdeView), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    Timer t(c, &c->tm.ms_fused, &c->tm.n_fused);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), lds, c->stream, plane, M, c->d_kv.as<KdeView>(),
                       kv.acc_stride, n, (int)M, ko->wt_thresh, ko->normalize, c->d_cand.as<Cand>(), M, lmap, levid, pdfs);
    HIPCHK(hipGetLastError());
    c->last_form = "k_plane_fused";
    return 0;
}

// class-sorted copy of the model records (many dictionary widths, k_fused MC): row j' <- row perm[j'], pads in place
static __global__ void k_permute_records(const double* __restrict__ in, const int* __restrict__ perm, int64_t M, int64_t Mp, int rw,
                                         double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Mp * rw) return;
    const int64_t j = e / rw; const int r = (int)(e - j * rw);
    out[e] = in[(j < M ? (int64_t)perm[j] : j) * rw + r];
}
inline int fz_mc_records(fz_ctx* c, bool rec0) {
    bool& valid = rec0 ? c->mc_rec0_valid : c->mc_rec1_valid;
    if (valid) return 0;
    const int rw = rec0 ? fz_rec_width(2 * c->BT) : fz_rec_width(c->BT);
    DevBuf& dst = rec0 ? c->d_rec0p : c->d_rec1p;
    FZCHK(dst.ensure((size_t)c->Mp * rw * 8));
    const int64_t tot = c->Mp * rw;
    hipLaunchKernelGGL(k_permute_records, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                       (rec0 ? c->d_rec0 : c->d_rec1).as<double>(), c->d_mc_perm.as<int>(), c->M, c->Mp, rw, dst.as<double>());
    HIPCHK(hipGetLastError());
    valid = true;
    return 0;
}

// single-pass kernel; returns +1 (not an error) when its candidate workspace does
// not fit the budget and the caller should take the two-pass route
template <class SRC, int TW, int NW, bool WM>
int fz_launch_fused_wm(fz_ctx* c, const SRC& src_in, const fz::KdeView& kv_in, int64_t n, int64_t M, const fz_kde_opts* ko,
                       double* lmap, double* levid, double* pdfs) {
    SRC src = src_in;
    fz::KdeView kv = kv_in;
    // many dictionary widths on the weight-space body: class-sorted records, one histogram, one convolution per class
    // present (fz_kernels.h, pdf_stage_mc); FZ_NO_MC=1 keeps the per-model window adds
    bool mc = false;
    if constexpr (WM && (NW == 12 || NW == 4)) {      // (2,16) spills inside the model loop with this PDF stage: the dispatcher sends it to (2,12)
        if (kv.kmode == fz::KDE_DICT && c->mc_ok && !fz_dbg("FZ_NO_MC")) {
            FZCHK(fz_mc_records(c, SRC::LMODE == 0));
            src.mv.rec0 = c->d_rec0p.as<double>(); src.mv.rec1 = c->d_rec1p.as<double>();
            kv.mc_tag = c->d_mc_tag.as<int32_t>(); kv.mc_width = c->d_mc_width.as<int32_t>(); kv.mc_off = c->d_mc_off.as<int64_t>();
            kv.mc_norm = c->d_mc_norm.as<double>(); kv.mc_gp = c->mc_gp; kv.mc_w0 = c->mc_w0;
            kv.acc_stride = c->mc_gp + 16;            // + the tail pad the sliding window of the convolution reads into
            mc = true;
        }
    }
    // the PDF rows live inside the (static) tile buffers when half of the waves' rows fit in one, else in dynamic LDS
    const size_t TDB = (size_t)SRC::template tile_doubles<SRC::template tile_len<NW>()>() +
                       ((WM && (kv.kmode == fz::KDE_HIST || mc)) ? SRC::template tile_len<NW>() / 2 : 0);       // + the index words (k_fused, POSW)
    const size_t lds = ((size_t)((NW + 1) / 2) * kv.acc_stride <= TDB) ? 0 : (size_t)NW * kv.acc_stride * 8;
    auto kern = (kv.kmode == fz::KDE_HIST) ? fz::k_fused<SRC, TW, NW, WM, true> : fz::k_fused<SRC, TW, NW, WM, false>;
    if constexpr (WM && (NW == 12 || NW == 4)) { if (mc) kern = fz::k_fused<SRC, TW, NW, WM, true, true>; }
    {
        hipFuncAttributes fa;
        HIPCHK(hipFuncGetAttributes(&fa, (const void*)kern));
        if (fa.sharedSizeBytes + lds > 160 * 1024) return 1;
    }
    const int64_t groups = (n + TW - 1) / TW;
    const size_t per_wave = (size_t)TW * M * sizeof(fz::Cand);
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks_per_cu = 1;           // resident blocks per CU for this kernel's registers and LDS
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, (const void*)kern, NW * 64, lds));
    blocks_per_cu = std::max(1, blocks_per_cu);
    // whole multiples of the CU count (an uneven tail of blocks would idle most CUs)
    const int64_t need = (groups + NW - 1) / NW;
    const int64_t fit = (int64_t)(c->ws_limit / (per_wave * NW));
    int64_t blocks = need;
    if (need > c->cu_count) {
        const int64_t k = std::min<int64_t>(blocks_per_cu, fit / c->cu_count);
        if (k >= 1) blocks = std::min<int64_t>(need, k * c->cu_count);
        else if (fit >= c->cu_count / 2) blocks = fit;             // very large model sets: part of the chip still beats two passes
        else return 1;                                             // cannot fill even half the chip
    } else if (fit < need) return 1;
    if (c->d_cand.ensure((size_t)blocks * NW * per_wave) != 0) return 1;      // no room for the lists: two-pass route
    // slot 0: the view of the main launch; slot 1: the caller's view (the sweep below always runs the general kernels)
    FZCHK(c->d_kv.ensure(2 * sizeof(fz::KdeView)));
    HIPCHK(hipMemcpyAsync(c->d_kv.p, &kv, sizeof(fz::KdeView), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_kv.as<fz::KdeView>() + 1, &kv_in, sizeof(fz::KdeView), hipMemcpyHostToDevice, c->stream));
    // objects the weight-space body hands back (no candidate / no fp32 weight at all): counter + list
    if (WM) {
        FZCHK(c->d_redo.ensure(((size_t)n + 1) * sizeof(int)));
        HIPCHK(hipMemsetAsync(c->d_redo.p, 0, sizeof(int), c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));          // kv is a stack object
    {
        Timer t(c, &c->tm.ms_fused, &c->tm.n_fused);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), lds, c->stream, src, c->d_kv.as<fz::KdeView>(),
                           kv.acc_stride, n, (int)M, ko->wt_thresh, ko->normalize, c->d_cand.as<fz::Cand>(), M, lmap, levid, pdfs,
                           c->omap, c->d_redo.as<int>(), (const int*)nullptr);
        if constexpr (WM) {
            // sweep: the fp64 ln-space body over exactly the handed-back objects (their chunk-level
            // indices are the object map; the count stays on the device, so nothing waits for it)
            constexpr int SW = 4;
            auto sweep = (kv_in.kmode == fz::KDE_HIST) ? fz::k_fused<SRC, 1, SW, false, true> : fz::k_fused<SRC, 1, SW, false, false>;
            constexpr size_t TDB2 = (size_t)SRC::template tile_doubles<SRC::template tile_len<SW>()>();
            size_t lds2 = ((size_t)((SW + 1) / 2) * kv_in.acc_stride <= TDB2) ? 0 : (size_t)SW * kv_in.acc_stride * 8;
            {
                hipFuncAttributes fa;
                HIPCHK(hipFuncGetAttributes(&fa, (const void*)sweep));
                if (fa.sharedSizeBytes + lds2 > 160 * 1024) lds2 = (size_t)1 << 30;      // cannot run: the objects keep their (flagged) rows
            }
            const int64_t sblocks = std::min<int64_t>(std::min<int64_t>(c->cu_count, (n + SW - 1) / SW), (int64_t)((size_t)blocks * NW * per_wave / ((size_t)SW * M * sizeof(fz::Cand))));
            if (lds2 <= 160 * 1024 && sblocks >= 1) {
                HIPCHK(hipFuncSetAttribute((const void*)sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                hipLaunchKernelGGL(sweep, dim3((unsigned)sblocks), dim3(SW * 64), lds2, c->stream, src_in, c->d_kv.as<fz::KdeView>() + 1,
                                   kv_in.acc_stride, n, (int)M, ko->wt_thresh, ko->normalize, c->d_cand.as<fz::Cand>(), M, lmap, levid, pdfs,
                                   c->d_redo.as<int>() + 1, (int*)nullptr, c->d_redo.as<int>());
            }
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// the weight-space kernel body is used for the chi2^(k/2) likelihoods of the exact band counts
// (4-8 bands unmasked, dimensionality prior on) in modes A / Ai; with the free scale (mode B) it
// is used at 4 and 5 bands (5 bands: 100.0 vs 110.7 ms per 262 144 x 1e5 launch once the kernel
// no longer carried the window-scatter code; at 6 bands the ln-space body is faster, 108.9 vs
// 120.8 ms); from 7 bands up only mode Ai keeps it (register budget)
// (FZ_NO_WSPACE=1 forces the ln-space body: A/B aid)
template <class SRC>
constexpr bool fz_has_wspace() {
    return SRC::WPOW >= 1 && SRC::WPOW <= 6 && (SRC::LMODE != 2 || SRC::NB <= 5) && (SRC::NB < 7 || SRC::LMODE == 1);
}
// exact_evidence of the call being launched (fz_launch_fitpredict sets it): the fp32-remainder bodies are not used
inline bool& fz_exact_now() { static thread_local bool v = false; return v; }
template <class SRC>
bool fz_use_wspace(const SRC& src) {
    if constexpr (fz_has_wspace<SRC>()) return src.lp.dim_prior && !fz_dbg("FZ_NO_WSPACE") && !fz_exact_now();
    return false;
}
template <class SRC, int TW, int NW>
int fz_launch_fused_tw(fz_ctx* c, const SRC& src, const fz::KdeView& kv, int64_t n, int64_t M, const fz_kde_opts* ko,
                       double* lmap, double* levid, double* pdfs) {
    if constexpr (fz_has_wspace<SRC>()) {
        if (fz_use_wspace(src)) return fz_launch_fused_wm<SRC, TW, NW, true>(c, src, kv, n, M, ko, lmap, levid, pdfs);
    }
    return fz_launch_fused_wm<SRC, TW, NW, false>(c, src, kv, n, M, ko, lmap, levid, pdfs);
}

// share of the launch's (object, model) pairs within the weight threshold, from a sample (fz_nolist.h); < 0: not applicable.
// The sample of THIS launch is queued and read back by the NEXT one (pinned word + event), so only the first launch after a
// model upload waits for the device; consecutive chunks / steps of one data set have the same statistics.
template <class SRC>
double fz_nolist_probe(fz_ctx* c, const SRC& src, const fz::KdeView& kv, int64_t n, int64_t M, const fz_kde_opts* ko) {
    if constexpr (!fz_has_wspace<SRC>()) return -1.0;
    else {
        if (!src.lp.dim_prior || fz_dbg("FZ_NO_WSPACE") || kv.kmode != fz::KDE_HIST || !kv.normtab || !(ko->wt_thresh > 0.0)) return -1.0;
        const int S = 256;
        const double denom = (double)S * (double)((M + 255) / 256) * 64.0;
        if (!c->h_probe) {
            if (hipHostMalloc((void**)&c->h_probe, 8) != hipSuccess) { (void)hipGetLastError(); c->h_probe = nullptr; return -1.0; }
            if (hipEventCreateWithFlags(&c->ev_probe, hipEventDisableTiming) != hipSuccess) return -1.0;
        }
        if (c->probe_pending) {                                   // the previous launch's sample has long landed
            if (hipEventSynchronize(c->ev_probe) != hipSuccess) return -1.0;
            c->probe_share = (double)*c->h_probe / denom;
            c->probe_pending = false;
        }
        if (c->d_flags.ensure(64) != 0) return -1.0;
        if (hipMemsetAsync(c->d_flags.p, 0, 8, c->stream) != hipSuccess) return -1.0;
        hipLaunchKernelGGL((fz::k_nl_probe<SRC>), dim3(S / 4), dim3(256), 0, c->stream, src, n, (int)M, S, ko->wt_thresh, c->omap,
                           c->d_flags.as<unsigned long long>());
        if (hipMemcpyAsync(c->h_probe, c->d_flags.p, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1.0;
        if (hipEventRecord(c->ev_probe, c->stream) != hipSuccess) return -1.0;
        c->probe_pending = true;
        if (c->probe_share < 0.0) {                               // nothing known yet: this once, wait
            if (hipEventSynchronize(c->ev_probe) != hipSuccess) return -1.0;
            c->probe_share = (double)*c->h_probe / denom;
            c->probe_pending = false;
        }
        return c->probe_share;
    }
}

// single pass with in-kernel LDS histograms and no candidate lists (fz_hist.h); +1 = not applicable / does not fit.
// exact: every weight in fp64 (the all-fp64 evidence, and the form for broad likelihoods)
// OBJK / SWS: per-object band counts (fz_hist.h); the sweep over handed-back objects then runs on `sws`, the MASKED variant of
// the same likelihood when objects may have unobserved bands (the mask-free arithmetic of `src` does not know N_dim per object)
// SEG (segmented model layout): `src` / `kv` / `M` are the segment-ordered records, their view and padded length; the sweep runs on
// the caller's own records: `swsp`, `kv_sweep`, `M_sweep`
// HOS: the sweep's KDE stage knows one dictionary kernel only (false: the general window stack -- many widths)
template <class SRC, int TW, int NW, bool EXACT, bool OBJK = (SRC::NB > 8), class SWS = SRC, bool SEG = false, bool HOS = true>
int fz_launch_hist_g(fz_ctx* c, const SRC& src, const fz::KdeView& kv, int64_t n, int64_t M, const fz_kde_opts* ko,
                     double* lmap, double* levid, double* pdfs, const SWS* swsp = nullptr, const fz::KdeView* kv_sweep = nullptr,
                     int64_t M_sweep = 0) {
    constexpr int SW = 4;
    auto kern = fz::k_hist<SRC, TW, NW, EXACT, OBJK, SEG>;
    const int64_t Msw = kv_sweep ? M_sweep : M;
    const fz::KdeView& kvs = kv_sweep ? *kv_sweep : kv;
    const size_t lds = (size_t)NW * TW * kv.acc_stride * 8;
    {
        hipFuncAttributes fa;
        HIPCHK(hipFuncGetAttributes(&fa, (const void*)kern));
        if (fa.sharedSizeBytes + lds > 160 * 1024) return 1;
    }
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 1;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, NW * 64, lds));
    bpc = std::max(1, bpc);
    const int64_t groups = (n + TW - 1) / TW;
    const int64_t need = (groups + NW - 1) / NW;
    // the ambiguous lists: a thin band of the candidates when the best model fits well (~1 % of M on the benchmark).  Sized at
    // M up to 131 072 models (can never overflow), M / 8 beyond (an object that overflows is re-run by the exact sweep): a
    // 1e6-model set then keeps the whole chip busy inside the workspace budget
    int64_t acap = (M <= 131072) ? M : std::max<int64_t>(131072, M / 8);
    if (const char* e = fz_dbg("FZ_HIST_AMBCAP")) acap = std::max<int64_t>(1, atoll(e));      // (tests: forces the overflow hand-back)
    const size_t per_wave = (size_t)TW * acap * sizeof(fz::Cand);
    const int64_t fit = (int64_t)(c->ws_limit / (per_wave * NW));
    int64_t blocks = need;
    if (need > c->cu_count) {
        const int64_t k = std::min<int64_t>(bpc, fit / c->cu_count);
        if (k >= 1) blocks = std::min<int64_t>(need, k * c->cu_count);
        else if (fit >= c->cu_count / 2) blocks = fit;
        else return 1;
    } else if (fit < need) return 1;
    // the sweep over handed-back objects (exact ln-space body, candidate lists of M entries per wave): as many blocks as stay
    // resident and fit the workspace (with per-object band counts a few per cent of a chunk can land there), at least one per CU
    auto sweep = fz::k_fused<SWS, 1, SW, false, HOS>;
    constexpr size_t TDB2 = (size_t)SWS::template tile_doubles<SWS::template tile_len<SW>()>();
    const size_t lds2 = ((size_t)((SW + 1) / 2) * kvs.acc_stride <= TDB2) ? 0 : (size_t)SW * kvs.acc_stride * 8;
    bool sweep_ok = false;
    int bps = 1;
    {
        hipFuncAttributes fa;
        HIPCHK(hipFuncGetAttributes(&fa, (const void*)sweep));
        if (fa.sharedSizeBytes + lds2 <= 160 * 1024) {
            HIPCHK(hipFuncSetAttribute((const void*)sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bps, (const void*)sweep, SW * 64, lds2));
            sweep_ok = true;
        }
    }
    const size_t sweep_blk = (size_t)SW * Msw * sizeof(fz::Cand);
    bps = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(bps, 4), (int64_t)(c->ws_limit / (sweep_blk * c->cu_count))));
    const size_t sweep_ws = (size_t)bps * c->cu_count * sweep_blk;
    if (c->d_cand.ensure(std::max((size_t)blocks * NW * per_wave, sweep_ws)) != 0) return 1;
    FZCHK(c->d_kv.ensure(2 * sizeof(fz::KdeView)));
    HIPCHK(hipMemcpyAsync(c->d_kv.p, &kv, sizeof(fz::KdeView), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_kv.as<fz::KdeView>() + 1, &kvs, sizeof(fz::KdeView), hipMemcpyHostToDevice, c->stream));
    FZCHK(c->d_redo.ensure(((size_t)n + 1) * sizeof(int)));
    HIPCHK(hipMemsetAsync(c->d_redo.p, 0, sizeof(int), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));          // kv is a stack object
    Timer t(c, &c->tm.ms_fused, &c->tm.n_fused);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), lds, c->stream, src, c->d_kv.as<fz::KdeView>(), kv.acc_stride, n,
                       (int)M, ko->wt_thresh, ko->normalize, c->d_cand.as<fz::Cand>(), acap, lmap, levid, pdfs, c->omap, c->d_redo.as<int>());
    if (sweep_ok) {
        SWS sws;
        if constexpr (std::is_same<SWS, SRC>::value) sws = src; else sws = *swsp;
        hipLaunchKernelGGL(sweep, dim3((unsigned)std::min<int64_t>((int64_t)bps * c->cu_count, (n + SW - 1) / SW)), dim3(SW * 64), lds2, c->stream, sws,
                           c->d_kv.as<fz::KdeView>() + 1, kvs.acc_stride, n, (int)Msw, ko->wt_thresh, ko->normalize, c->d_cand.as<fz::Cand>(), Msw,
                           lmap, levid, pdfs, c->d_redo.as<int>() + 1, (int*)nullptr, c->d_redo.as<int>());
    }
    HIPCHK(hipGetLastError());
    return 0;
}
template <class SRC>
int fz_launch_hist(fz_ctx* c, const SRC& src, const fz::KdeView& kv, int64_t n, int64_t M, const fz_kde_opts* ko,
                   double* lmap, double* levid, double* pdfs, bool exact) {
    // exact band counts without masks: 4-8 bands, and the wide sets (16, 32 bands) except 32 bands with per-model errors, whose
    // object and record (192 doubles) leave the register file no room (k_fused keeps that case)
    if constexpr (!(SRC::WPOW >= 1 && SRC::WPOW <= 30) || (SRC::NB > 16 && SRC::LMODE == 0)) return 1;
    else {
        if (!src.lp.dim_prior || kv.kmode != fz::KDE_HIST || !kv.normtab || !(ko->wt_thresh > 0.0) || M >= ((int64_t)1 << 31)) return 1;
        // the free scale (mode B) weighs every pair in fp64 straight away: its chi2 takes two passes over the bands, and screening with
        // the closed form first was slower (fz_hist.h)
        const bool ex = exact || SRC::LMODE == 2;
        if constexpr (SRC::NB > 8) {
            // wide records: one object per wave, eight waves per block (up to 256 registers per lane)
            if (fz_dbg("FZ_HIST_WIDE") && atoi(fz_dbg("FZ_HIST_WIDE")) == 0) return 1;      // (tests: the masked kernels of round 2)
            if (ex) return fz_launch_hist_g<SRC, 1, 8, true>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            return fz_launch_hist_g<SRC, 1, 8, false>(c, src, kv, n, M, ko, lmap, levid, pdfs);
        } else {
            if (ex) return fz_launch_hist_g<SRC, 1, 16, true>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            if constexpr (SRC::LMODE != 2) return fz_launch_hist_g<SRC, 1, 16, false>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            return 1;
        }
    }
}

// k_hist with per-object band counts or nothing (+1): a chunk whose objects have unobserved bands, unmasked models, modes Ai / B
// (4-8 bands; the wide sets take this form anyway).  `src`: the mask-free variant, `sws`: the masked one (sweep).
template <class SRC, class SWS>
int fz_launch_hist_objmask(fz_ctx* c, const SRC& src, const SWS& sws, int64_t n, int64_t M, const fz_kde_opts* ko, double* lmap, double* levid,
                           double* pdfs) {
    if constexpr (!(SRC::WPOW >= 1 && SRC::WPOW <= 30) || SRC::LMODE == 0) return 1;
    else {
        fz::KdeView kv;
        FZCHK(fz_kde_view(c, kv));
        if (c->force_twopass || (fz_dbg("FZ_HIST") && atoi(fz_dbg("FZ_HIST")) == 0) || (fz_dbg("FZ_HIST_OBJMASK") && atoi(fz_dbg("FZ_HIST_OBJMASK")) == 0)) return 1;
        // (every form of k_hist forms and sums its weights in fp64: a request for the exact evidence needs no other kernel)
        if (!(ko->wt_thresh > 0.0) || M >= ((int64_t)1 << 31) || kv.kmode != fz::KDE_HIST || !kv.normtab) return 1;
        if (!src.lp.dim_prior && fz_dbg("FZ_HIST_NODIMPRIOR") && atoi(fz_dbg("FZ_HIST_NODIMPRIOR")) == 0) return 1;
        fz_exact_now() = false;
        constexpr int NWH = SRC::NB > 8 ? 8 : 16;
        int r;
        if constexpr (SRC::LMODE == 2) r = fz_launch_hist_g<SRC, 1, NWH, true, true, SWS>(c, src, kv, n, M, ko, lmap, levid, pdfs, &sws);
        else r = fz_launch_hist_g<SRC, 1, NWH, false, true, SWS>(c, src, kv, n, M, ko, lmap, levid, pdfs, &sws);
        if (r <= 0) c->last_form = SRC::LMODE == 2 ? "k_hist<exact> (per-object band counts)" : "k_hist<screen> (per-object band counts)";
        return r;
    }
}

// k_hist on the segmented model layout or nothing (+1): masked models (any mode), objects with unobserved bands against per-model
// errors.  `src`: the mask-free variant (its records are replaced by the segment-ordered copy), `sws`: the masked one (sweep over
// the handed-back objects, on the caller's own records).  FZ_HIST_SEG=0 (tests) keeps the masked kernels of k_fused.
template <class SRC, class SWS>
int fz_launch_hist_seg(fz_ctx* c, const SRC& src, const SWS& sws, int64_t n, int64_t M, const fz_kde_opts* ko, double* lmap, double* levid,
                       double* pdfs) {
    if constexpr (!(SRC::WPOW >= 1 && SRC::WPOW <= 6) || SRC::NB > 8) return 1;
    else {
        fz::KdeView kv0;
        FZCHK(fz_kde_view(c, kv0));
        if (c->force_twopass || (fz_dbg("FZ_HIST") && atoi(fz_dbg("FZ_HIST")) == 0) || (fz_dbg("FZ_HIST_SEG") && atoi(fz_dbg("FZ_HIST_SEG")) == 0)) return 1;
        // one dictionary kernel (histogram + one convolution), or many through the class-ordered segments (one convolution per class)
        const bool mcw = kv0.kmode == fz::KDE_DICT && c->mc_ok && !(fz_dbg("FZ_HIST_SEG_MC") && atoi(fz_dbg("FZ_HIST_SEG_MC")) == 0);
        if (!(ko->wt_thresh > 0.0) || M >= ((int64_t)1 << 31)) return 1;
        if (!mcw && (kv0.kmode != fz::KDE_HIST || !kv0.normtab)) return 1;
        // without the dimensionality prior the ln-like of mode A carries sum_b ln(xe^2 + ye^2) of the PAIR (pdf.py:96-98): not a power-0 form
        if (!src.lp.dim_prior && SRC::LMODE == 0) return 1;
        if (SRC::LMODE == 0 && c->models_big) return 1;           // (fluxes beyond 1e9: the bound on an unobserved object band's term, fz_hist.h)
        const int rs = fz_segments(c, SRC::LMODE == 0);
        if (rs != 0) return rs;
        fz::KdeView kv = kv0;
        kv.mc_tag = c->d_seg_tag.as<int32_t>(); kv.seg_mask = c->d_seg_mask.as<uint32_t>(); kv.seg_rank = c->d_seg_rank.as<int32_t>();
        kv.seg_start = c->d_seg_start.as<int32_t>(); kv.seg_n = c->seg_n; kv.seg_nrank = c->seg_nrank;
        if (mcw != (c->seg_nrank > 1)) return 1;                  // (cannot happen: both say "more than one class present")
        if (mcw) {
            kv.mc_width = c->d_mc_width.as<int32_t>(); kv.mc_off = c->d_mc_off.as<int64_t>(); kv.mc_norm = c->d_mc_rnorm.as<double>();      // (reciprocals)
            kv.mc_gp = c->mc_gp; kv.mc_w0 = c->mc_w0;
            kv.acc_stride = c->mc_gp + 16;                       // + the tail pad the sliding window of the convolution reads into
        }
        SRC s2 = src;
        s2.mv.rec0 = c->d_seg_rec0.as<double>(); s2.mv.rec1 = c->d_seg_rec1.as<double>();
        fz_exact_now() = false;
        // direct form: the free scale always (fz_hist.h); broad likelihoods by the sampled share of pairs within the threshold (the
        // sample's mask-free arithmetic is an estimate here, which is all the choice needs); FZ_NOLIST=1 / 0 forces / forbids
        bool ex = SRC::LMODE == 2 || (fz_dbg("FZ_EXACT_EVIDENCE") && atoi(fz_dbg("FZ_EXACT_EVIDENCE")) != 0);
        if (!ex) {
            const char* e = fz_dbg("FZ_NOLIST");
            const int want = e ? atoi(e) : -1;
            if (want < 0 && n >= 16384) ex = fz_nolist_probe<SRC>(c, src, kv0, n, M, ko) > 0.12;
            else ex = want == 1;
        }
        int r;
        constexpr int NWS = (SRC::LMODE == 0) ? FZ_HIST_SEG_NW0 : 16;      // waves per block (per-model errors: see FZ_HIST_SEG_NW0)
        // many widths: 12 waves per block (a histogram row of G + 2 W0 + 16 entries: 16 of them do not fit beside the tiles)
        if (mcw) {
            if (ex) r = fz_launch_hist_g<SRC, 1, 12, true, true, SWS, true, false>(c, s2, kv, n, c->seg_Ms, ko, lmap, levid, pdfs, &sws, &kv0, M);
            else if constexpr (SRC::LMODE != 2) r = fz_launch_hist_g<SRC, 1, 12, false, true, SWS, true, false>(c, s2, kv, n, c->seg_Ms, ko, lmap, levid, pdfs, &sws, &kv0, M);
            else r = 1;
        }
        else if (ex) r = fz_launch_hist_g<SRC, 1, NWS, true, true, SWS, true>(c, s2, kv, n, c->seg_Ms, ko, lmap, levid, pdfs, &sws, &kv0, M);
        else if constexpr (SRC::LMODE != 2) r = fz_launch_hist_g<SRC, 1, NWS, false, true, SWS, true>(c, s2, kv, n, c->seg_Ms, ko, lmap, levid, pdfs, &sws, &kv0, M);
        else r = 1;
        if (r <= 0) c->last_form = mcw ? (ex ? "k_hist<exact> (segmented models, many widths)" : "k_hist<screen> (segmented models, many widths)")
                                       : (ex ? "k_hist<exact> (segmented models)" : "k_hist<screen> (segmented models)");
        return r;
    }
}

// k_hist or nothing (+1): the wide band sets (16 / 32 real bands, no masks), whose other kernels exist in the masked variants only
template <class SRC>
int fz_launch_hist_only(fz_ctx* c, const SRC& src, int64_t n, int64_t M, const fz_kde_opts* ko, double* lmap, double* levid, double* pdfs) {
    fz::KdeView kv;
    FZCHK(fz_kde_view(c, kv));
    if (c->force_twopass || (fz_dbg("FZ_HIST") && atoi(fz_dbg("FZ_HIST")) == 0)) return 1;
    // FZ_EXACT_EVIDENCE=1 (tests): the form that weighs every pair without classifying it first
    const bool exact = fz_dbg("FZ_EXACT_EVIDENCE") && atoi(fz_dbg("FZ_EXACT_EVIDENCE")) != 0;
    fz_exact_now() = exact || c->exact_evidence;
    const int r = fz_launch_hist<SRC>(c, src, kv, n, M, ko, lmap, levid, pdfs, exact);
    if (r <= 0) c->last_form = (exact || SRC::LMODE == 2) ? "k_hist<exact>" : "k_hist<screen>";
    return r;
}

// fit_predict on a prepared chunk: single pass when possible, else two passes
template <class SRC>
int fz_launch_fitpredict(fz_ctx* c, const SRC& src, int64_t n, int64_t M, const fz_kde_opts* ko, double* lmap,
                         double* levid, double* pdfs) {
    fz::KdeView kv;
    FZCHK(fz_kde_view(c, kv));
    if (!c->force_twopass) {
        // (objects per wave, waves per block).  Few objects: one per wave so that the
        // chunk spreads over the chip.  FZ_FUSED_CFG=tw,nw overrides (tuning aid).
        int r = 1;
        // the default where it applies: one pass, LDS histograms, no candidate lists, every weight and sum in fp64 (fz_hist.h); FZ_HIST=0
        // keeps k_fused, FZ_EXACT_EVIDENCE=1 (tests) the form that weighs every pair without classifying it first.  like_opts.exact_evidence
        // matters to k_fused's weight-space body only (fp32 remainder of the evidence there)
        const bool exact = fz_dbg("FZ_EXACT_EVIDENCE") && atoi(fz_dbg("FZ_EXACT_EVIDENCE")) != 0;
        fz_exact_now() = exact || c->exact_evidence;
        double share = -2.0;                                     // not sampled yet
        if (!fz_dbg("FZ_HIST") || atoi(fz_dbg("FZ_HIST")) != 0) {
            // The classifier pays when it drops most pairs: with 7 % of the pairs within wt_thresh of the best (41 % above the drop
            // bar) it runs level with the form that weighs every pair directly (53.4 vs 54.9 ms per 2.6e10 pairs), with 3 % ahead of
            // it (46.0 vs 55.2); for broader likelihoods (faint data: the reference's own mock sits at 41 %; bench.py --noise-scale
            // 3 / 10: 53 % / 95 %) the direct form is the faster one.  FZ_NOLIST=1 / 0 forces / forbids the switch.
            bool broad = false;
            if (!exact) {
                const char* e = fz_dbg("FZ_NOLIST");
                const int want = e ? atoi(e) : -1;
                if (want < 0 && n >= 16384) { share = fz_nolist_probe<SRC>(c, src, kv, n, M, ko); broad = share > 0.12; }
                else broad = want == 1;
            }
            r = fz_launch_hist<SRC>(c, src, kv, n, M, ko, lmap, levid, pdfs, exact || broad);
            if (r <= 0) { c->last_form = (exact || broad || SRC::LMODE == 2) ? ((exact || SRC::LMODE == 2) ? "k_hist<exact>" : "k_hist<exact> (broad likelihoods)") : "k_hist<screen>"; return r; }
        }
        if constexpr (SRC::NB > 16) {
            // wide records (17-32 bands): one object per wave keeps the kernel inside the register file
            r = fz_launch_fused_tw<SRC, 1, 4>(c, src, kv, n, M, ko, lmap, levid, pdfs);
        } else if constexpr (SRC::HAS_PRIOR) {
            // ln-prior rows are streamed from HBM beside the LDS model tiles: two geometries per body
            if (n < (int64_t)c->cu_count * 64) r = fz_launch_fused_tw<SRC, 1, 4>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            else if constexpr (SRC::PREF_2x16) r = fz_launch_fused_tw<SRC, 2, 16>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            else r = fz_launch_fused_tw<SRC, 4, 8>(c, src, kv, n, M, ko, lmap, levid, pdfs);
        } else {
            int tw = (n >= (int64_t)c->cu_count * 64) ? 4 : 1, nw = (tw == 1) ? 4 : 8;
            // measured best geometry per kernel body (profiles/README.md): up to 6 bands 16 waves x
            // 2 objects (128 VGPRs) for the weight-space body and every ln-space body except
            // unmasked mode A and masked mode B, which want the 256-VGPR budget of 8 waves x 4;
            // wider records spill at 128 VGPRs (PREF_2x16 / PREF_2x8 in PhotSrc)
            if (tw == 4 && (fz_use_wspace(src) || SRC::PREF_2x16)) { tw = 2; nw = 16; }
            // weight-space body with wide records (general mode A: y and ye^2; 7 and 8 bands): 16 waves leave
            // 128 VGPRs, and the body then spills inside the model loop (3-4x slower); 12 waves (168 VGPRs) do not
            if (tw == 2 && nw == 16 && fz_use_wspace(src) && (SRC::LMODE == 0 || SRC::NB >= 7)) nw = 12;
            // ... and so does the class-sorted stack of many dictionary widths (its PDF stage keeps a 12-register result row)
            if (tw == 2 && nw == 16 && fz_use_wspace(src) && kv.kmode == fz::KDE_DICT && c->mc_ok && !fz_dbg("FZ_NO_MC")) nw = 12;
            else if (tw == 4 && SRC::PREF_2x8) { tw = 2; nw = 8; }
            if (const char* e = fz_dbg("FZ_FUSED_CFG")) sscanf(e, "%d,%d", &tw, &nw);
            if (tw == 4 && nw == 8) r = fz_launch_fused_tw<SRC, 4, 8>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            else if (tw == 2 && nw == 8) r = fz_launch_fused_tw<SRC, 2, 8>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            else if (tw == 2 && nw == 16) r = fz_launch_fused_tw<SRC, 2, 16>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            else if (tw == 2 && nw == 12) r = fz_launch_fused_tw<SRC, 2, 12>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            else if (tw == 1 && nw == 4) r = fz_launch_fused_tw<SRC, 1, 4>(c, src, kv, n, M, ko, lmap, levid, pdfs);
            else return fail(-1, "FZ_FUSED_CFG=%d,%d is not an instantiated configuration", tw, nw);
        }
        if (r <= 0) { c->last_form = "k_fused"; return r; }
    }
    c->last_form = "k_stats + k_kde";
    if (c->omap) return 2;          // an object subset: only the fused kernel takes one; the caller redoes the whole chunk
    FZCHK(fz_launch_stats(c, src, n, M, 0, lmap, levid));
    return fz_launch_kde(c, src, n, M, 0, lmap, levid, ko, pdfs);
}
