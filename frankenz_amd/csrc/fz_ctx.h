// Host-side context shared by the translation units of libfrankenz_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/frankenz_hip.h"
#include "fz_device.h"
#include "fz_knn.h"

// ---- errors ------------------------------------------------------------------
std::string& fz_err_slot();                       // thread-local, defined in frankenz_hip.hip
inline int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    fz_err_slot() = buf;
    return code;
}
#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(-1, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define FZCHK(call) do { int r_ = (call); if (r_ != 0) return r_; } while (0)

// The model arrays are padded to whole LDS tiles of every kernel that stages them: k_fused up to 1024 models, k_hist 256 / 128 / 64
// or 384 (FZ_HIST_TILE384): the least common multiple
#define FZ_MP_ALIGN 3072

// ---- grow-only cached device allocation ----------------------------------------
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(-2, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// Test / tuning switches ("FZ_..." names): set by ONE call, fz_debug_opts (include/frankenz_hip.h) -- the library does not read the
// environment.  nullptr when the switch is not set (the meaning getenv had for the code that consults it).
const char* fz_dbg(const char* name);

struct fz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // where device-resident inputs come from (fz_set_producer_stream): 0 unknown -> device-wide wait, 1 `producer_stream` -> event wait,
    // 2 already complete -> no wait
    int producer_mode = 0; hipStream_t producer_stream = nullptr; hipEvent_t ev_producer = nullptr;
    // host-PDF pipeline of fit_predict: chunk k's rows leave on copy_stream while chunk k+1 is computed
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_done[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    // deferred kernel timing (no host block inside that pipeline): event pairs resolved by timer_flush
    bool defer_timing = false;
    struct PendingTime { hipEvent_t a, b; double* ms; int64_t* n; };
    std::vector<PendingTime> pending;
    std::vector<hipEvent_t> ev_pool;
    fz_timing tm{};
    int64_t ws_limit = (int64_t)32 << 30;
    int cu_count = 256;
    int force_twopass = 0;     // diagnostics: disable the single-pass fused kernel
    // share of (object, model) pairs within the weight threshold, sampled per fused launch and read back one launch later
    // (pinned host word + event: the choice of kernel form never waits for the device after the first launch of a model set)
    unsigned long long* h_probe = nullptr; hipEvent_t ev_probe = nullptr; bool probe_pending = false; double probe_share = -1.0;
    std::string last_form;     // which kernel form the last fused launch took (fz_last_form)
    int exact_evidence = 0;    // every weight of the fused path's ln-evidence in fp64 (fz_like_opts.exact_evidence of the call being served)

    // models (BruteForce.__init__)
    int64_t M = 0, Mp = 0; int B = 0, BT = 0;
    bool models_masked = false, models_real_masked = false, models_wild = false, models_err_const = false, models_big = false;
    int64_t mc_info[4] = {0, 0, 0, 0};   // fz_modec_info: ambiguous objects re-run, slowest object's iterations, path, block shape
    int mc_lnl_only = 0;           // mode C: the caller of run_modec wants the final ln-like plane only (request); set back to 0 by whoever cannot honour it
    double grid_step = 0.0;        // gauss_kde grid labels: spacing of an evenly spaced grid (checked on upload), else 0
    DevBuf d_y, d_ye2, d_ye, d_mbits, d_lgA, d_lgB, d_rec0, d_rec1, d_ye2c;
    // kde dictionary (PDFDict)
    int64_t G = 0, D = 0;       // G: grid length of the labels in force (dictionary or direct grid)
    int64_t dict_G = 0;         // grid length of the uploaded dictionary (restored by fz_labels_upload_dict)
    std::vector<int64_t> h_widths, h_offsets; std::vector<double> h_kcdf;
    DevBuf d_widths, d_offsets, d_kern;
    // labels
    int label_mode = 0;        // 0 none, 1 dict, 2 grid
    int64_t label_M = 0;
    bool single_cls = false; int32_t cls0 = 0, w0 = 0;
    DevBuf d_pos, d_cls, d_norm, d_normtab, d_ly, d_lstd, d_lo, d_hi, d_grid;
    // class-sorted copy of the fused kernel's model records + tables (many dictionary widths; fz_kernels.h, MC)
    bool mc_ok = false, mc_rec0_valid = false, mc_rec1_valid = false; int32_t mc_gp = 0, mc_w0 = 0;
    DevBuf d_mc_tag, d_mc_perm, d_mc_width, d_mc_off, d_mc_norm, d_mc_rnorm, d_rec0p, d_rec1p;
    // segmented model layout of the one-pass kernel (k_hist<..., SEG>: masked models, per-model errors against masked objects):
    // records sorted by (dictionary class, mask pattern), segments padded to whole 64-model groups.  Built on first use from the host
    // copies below (fz_segments); seg_state: 0 not built, 1 built, -1 this model / label set does not take the form
    std::vector<uint32_t> h_mbits; std::vector<int32_t> h_pos, h_cls;
    int seg_state = 0; int64_t seg_Ms = 0; int32_t seg_n = 0, seg_nrank = 0; bool seg_rec0_valid = false, seg_rec1_valid = false;
    DevBuf d_seg_tag, d_seg_perm, d_seg_mask, d_seg_rank, d_seg_start, d_seg_rec0, d_seg_rec1;
    // per-chunk object buffers
    DevBuf d_sx, d_sxe, d_sxm;       // a host call's whole object set, staged once (fz_fit_predict)
    DevBuf d_rx, d_rxe, d_rxm, d_ox, d_ov, d_obits, d_oslv, d_flags;
    DevBuf d_lmap, d_levid, d_pdfs, d_pdfs2;
    DevBuf d_pl[7];            // staging planes
    DevBuf d_mc[4], d_mcerr, d_mcfn, d_mcact, d_mccnt, d_mcniter; int64_t mc_niter_n = 0;
    DevBuf d_cand, d_kv;
    DevBuf d_net[10];                // staging of host arrays handed to the fz_net_* entry points
    DevBuf d_sgrid, d_sloss;         // pdfs_summarize: grid / loss matrix   // candidate lists / KDE table view / per-object statistics of the single-pass kernels
    // additive ln-prior of the chunk being processed (tab == nullptr: none); set by bind_prior
    fz::PriorView prior{};
    DevBuf d_ptab, d_prows, d_lrec;
    // object subset of the chunk being processed by the fused kernel (nullptr: all objects)
    const int* omap = nullptr;
    DevBuf d_omap, d_redo;     // d_redo: objects the weight-space body hands to the ln-space body (count, then indices)
    // knn
    int knn_K = 0, knn_F = 0; int64_t knn_M = 0; bool knn_mfma = false, knn_sorted = false; int knn_gsl = 0;
    DevBuf d_kgbox, d_ktbox;   // boxes of the scan's <= 128 tile groups per feature set, and of every 64-model tile
    DevBuf d_idxs;             // neighbour table of the chunk being searched and fitted (fz_knn_search_fit_predict)
    DevBuf d_trees, d_q, d_idx, d_nbr, d_nn, d_tnorm, d_kbmat, d_kcen, d_kpmax, d_kperm, d_ktab, d_kqperm, d_kqcnt;

    std::vector<DevBuf*> all_bufs() {
        std::vector<DevBuf*> v = {&d_y, &d_ye2, &d_ye, &d_rec0, &d_rec1, &d_ye2c, &d_mbits, &d_lgA, &d_lgB, &d_widths, &d_offsets, &d_kern, &d_pos,
                                  &d_cls, &d_norm, &d_normtab, &d_mc_tag, &d_mc_perm, &d_mc_width, &d_mc_off, &d_mc_norm, &d_mc_rnorm, &d_rec0p, &d_rec1p, &d_seg_tag, &d_seg_perm, &d_seg_mask, &d_seg_rank, &d_seg_start, &d_seg_rec0, &d_seg_rec1, &d_ly, &d_lstd, &d_lo, &d_hi, &d_grid, &d_sx, &d_sxe, &d_sxm, &d_rx, &d_rxe, &d_rxm, &d_ox,
                                  &d_ov, &d_obits, &d_oslv, &d_flags, &d_lmap, &d_levid, &d_pdfs, &d_pdfs2, &d_mcerr,
                                  &d_mcfn, &d_mcact, &d_mccnt, &d_mcniter, &d_cand, &d_kv, &d_omap, &d_redo, &d_sgrid, &d_sloss, &d_ptab, &d_prows, &d_lrec, &d_kgbox, &d_ktbox, &d_idxs, &d_trees, &d_q, &d_idx, &d_nbr, &d_nn, &d_tnorm, &d_kbmat, &d_kcen, &d_kpmax, &d_kperm, &d_ktab, &d_kqperm, &d_kqcnt};
        for (auto& b : d_pl) v.push_back(&b);
        for (auto& b : d_net) v.push_back(&b);
        for (auto& b : d_mc) v.push_back(&b);
        return v;
    }
};

inline bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// HIP-event bracket on the ctx stream, accumulated per kernel family.  Normally the destructor waits
// for the kernel (the figure is needed at once and the next host step would wait anyway); with
// c->defer_timing the pair is queued and resolved later by timer_flush, so the host can go on.
struct Timer {
    fz_ctx* c; double* ms; int64_t* n;
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t take(fz_ctx* c) {
        if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
        hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
    }
    Timer(fz_ctx* c_, double* ms_, int64_t* n_) : c(c_), ms(ms_), n(n_) {
        if (c->defer_timing) { a = take(c); b = take(c); (void)hipEventRecord(a, c->stream); }
        else (void)hipEventRecord(c->ev0, c->stream);
    }
    ~Timer() {
        if (a) { (void)hipEventRecord(b, c->stream); c->pending.push_back({a, b, ms, n}); return; }
        (void)hipEventRecord(c->ev1, c->stream);
        (void)hipEventSynchronize(c->ev1);
        float t = 0; (void)hipEventElapsedTime(&t, c->ev0, c->ev1);
        *ms += t; *n += 1;
    }
};
inline void timer_flush(fz_ctx* c) {
    for (auto& p : c->pending) {
        (void)hipEventSynchronize(p.b);
        float t = 0; (void)hipEventElapsedTime(&t, p.a, p.b);
        *p.ms += t; *p.n += 1;
        c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b);
    }
    c->pending.clear();
}

// copies with either side on host or device, ordered on the ctx stream
inline int copy_in(fz_ctx* c, void* dst_dev, const void* src, size_t bytes) {
    if (!bytes) return 0;
    HIPCHK(hipMemcpyAsync(dst_dev, src, bytes, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
inline int copy_out(fz_ctx* c, void* dst, const void* src_dev, size_t bytes) {
    if (!bytes) return 0;
    HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// record width (doubles) of the array-of-records model copies: >= nval and 2 mod 4
inline int fz_rec_width(int nval) { int w = nval; while (w % 4 != 2) ++w; return w; }

// ---- device views ------------------------------------------------------------
inline fz::ModelView model_view(fz_ctx* c) {
    fz::ModelView v; v.y = c->d_y.as<double>(); v.ye2 = c->d_ye2.as<double>(); v.ye = c->d_ye.as<double>();
    v.rec0 = c->d_rec0.as<double>(); v.rec1 = c->d_rec1.as<double>();
    v.bits = c->d_mbits.as<uint32_t>(); v.M = c->M; v.Mp = c->Mp; return v;
}
inline fz::ObjView obj_view(fz_ctx* c) {
    fz::ObjView v; v.x = c->d_ox.as<double>(); v.v = c->d_ov.as<double>();
    v.bits = c->d_obits.as<uint32_t>(); v.slv = c->d_oslv.as<double>(); return v;
}
inline fz::LikeParams like_params(fz_ctx* c, int mode, int dim_prior) {
    fz::LikeParams lp; lp.dim_prior = dim_prior; lp.nband = c->B;
    lp.lgtab = (mode == 2) ? c->d_lgB.as<double>() : c->d_lgA.as<double>();
    const double a = (mode == 2) ? 0.5 * (c->B - 1) : 0.5 * c->B;
    lp.lg_full = std::lgamma(a) + a * FZ_LN2;
    return lp;
}
inline int like_mode(const fz_like_opts* o) {      // 0 A, 1 Ai, 2 B, 3 C
    if (!o->free_scale) return o->ignore_model_err ? 1 : 0;
    return o->ignore_model_err ? 2 : 3;
}
// Mode A (fixed scale, model errors kept) with band-constant model errors is mode Ai on the
// variances xe^2 + ye^2[b], which then belong to the object: same chi2, N_dim and ln-like
// (pdf.py:76-98) to the rounding of one reciprocal, at the cost of the cheaper kernel.
inline int eff_mode(const fz_ctx* c, int mode) { return (mode == 0 && c->models_err_const) ? 1 : mode; }
inline int obj_vmode(const fz_ctx* c, int mode) {      // what k_prep_objects derives (see there)
    if (mode == 0) return c->models_err_const ? 2 : 0;
    return (mode == 1 || mode == 2) ? 1 : 0;
}
// arithmetic variant of a chunk: see VAR_* in fz_device.h
inline int pick_var(fz_ctx* c, int obj_flags) {
    if (c->models_wild || (obj_flags & 4)) return fz::VAR_SAFE;
    if (c->models_real_masked || (obj_flags & 1)) return fz::VAR_MASKED;
    if (c->BT != c->B) return c->BT > 8 ? fz::VAR_PAD : fz::VAR_MASKED;
    return fz::VAR_FAST;
}

// segmented layout (frankenz_hip.hip): 0 built (records of the asked kind valid), +1 not applicable, < 0 error
int fz_segments(fz_ctx* c, bool rec0);

// ---- per-band-count launchers (fz_inst.hip, one translation unit per BT) ------
#define FZ_DECL_BT(N)                                                                                     \
    int fz_planes_bt##N(fz_ctx* c, int mode, int var, int dim_prior, int64_t n, double* lnl, double* chi2, \
                        int64_t* ndim, double* scale, double* serr);                                      \
    int fz_fitpredict_bt##N(fz_ctx* c, int mode, int var, int dim_prior, int64_t n, const fz_kde_opts* ko, \
                            double* lmap, double* levid, double* pdfs);                                   \
    int fz_modec_bt##N(fz_ctx* c, int var, int64_t n, const fz_like_opts* o, const int64_t* nbr,           \
                       const int64_t* nnb, int W);                             \
    int fz_knnsubset_bt##N(fz_ctx* c, int mode, int var, int dim_prior, int64_t n, const int64_t* idx, int W, \
                           const fz_kde_opts* ko, const fz::KnnOut* out, int* errflag);                   \
    int fz_knnquery_bt##N(fz_ctx* c, const double* q, int64_t n, int k, double bound2, int64_t* idx, int pnorm);
FZ_DECL_BT(4)
FZ_DECL_BT(5)
FZ_DECL_BT(6)
FZ_DECL_BT(7)
FZ_DECL_BT(8)
FZ_DECL_BT(12)
FZ_DECL_BT(16)
FZ_DECL_BT(24)
FZ_DECL_BT(32)
#undef FZ_DECL_BT
