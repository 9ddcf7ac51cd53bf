// libfrankenz_hip.so -- C ABI (include/frankenz_hip.h) over the gfx950 kernels.
// This translation unit: context, uploads, object preparation, the ABI entry points
// and the kernels that do not depend on the band count.  The photometric kernels
// are instantiated per band count in fz_inst.hip.
#include <mutex>
#include "fz_ctx.h"
#include "fz_kernels.h"
#include "fz_cdf.h"
#include "fz_knn.h"
#include "fz_knn_mfma.h"
#include "fz_launch.h"
#include "fz_modec.h"
#include "fz_summary.h"
#include "fz_net.h"

using namespace fz;

std::string& fz_err_slot() { static thread_local std::string e; return e; }
extern "C" const char* fz_last_error(void) { return fz_err_slot().c_str(); }

// ---- test / tuning switches: one call sets them, nothing reads the environment ----
namespace {
struct DbgOpts { std::mutex mu; std::vector<std::pair<std::string, std::string>> kv; };
DbgOpts& dbg_opts() { static DbgOpts d; return d; }
}
const char* fz_dbg(const char* name) {
    DbgOpts& d = dbg_opts();
    std::lock_guard<std::mutex> lk(d.mu);
    for (auto& e : d.kv) if (e.first == name) return e.second.c_str();      // (stable until the next fz_debug_opts)
    return nullptr;
}
extern "C" int fz_debug_opts(const char* spec) {
    DbgOpts& d = dbg_opts();
    std::lock_guard<std::mutex> lk(d.mu);
    d.kv.clear();
    if (!spec) return 0;
    const char* p = spec;
    while (*p) {
        const char* e = p; while (*e && *e != ';') ++e;
        const char* q = p; while (q < e && *q != '=') ++q;
        if (q > p) d.kv.emplace_back(std::string(p, q), q < e ? std::string(q + 1, e) : std::string());
        p = *e ? e + 1 : e;
    }
    return 0;
}
extern "C" const char* fz_last_form(fz_ctx* c) { return c ? c->last_form.c_str() : ""; }

extern "C" int fz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int fz_ctx_create(int device, fz_ctx** out) {
    if (!out) return fail(-1, "fz_ctx_create: out is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(-1, "fz_ctx_create: device %d not in [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    fz_ctx* c = new fz_ctx();
    c->device = device;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_copied[k], hipEventDisableTiming));
    }
    hipDeviceProp_t pr;
    HIPCHK(hipGetDeviceProperties(&pr, device));
    c->cu_count = pr.multiProcessorCount;
    // workspace budget (candidate lists, staging planes): 45 % of the device memory, at least the
    // old fixed 32 GiB figure where the part allows it -- 130 GB on a 288-GB MI355X, which is what
    // the single-pass kernel wants for a 1e6-model set.  Allocation stays on demand.
    {
        const int64_t total = (int64_t)pr.totalGlobalMem;
        c->ws_limit = std::max<int64_t>(std::min<int64_t>((int64_t)32 << 30, total / 2), (int64_t)(0.45 * (double)total));
    }
    *out = c;
    return 0;
}

extern "C" void fz_ctx_destroy(fz_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : c->all_bufs()) b->release();
    (void)hipStreamSynchronize(c->copy_stream);
    timer_flush(c);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    for (int k = 0; k < 2; ++k) { (void)hipEventDestroy(c->ev_done[k]); (void)hipEventDestroy(c->ev_copied[k]); }
    if (c->ev_probe) (void)hipEventDestroy(c->ev_probe);
    if (c->ev_producer) (void)hipEventDestroy(c->ev_producer);
    if (c->h_probe) (void)hipHostFree(c->h_probe);
    (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int fz_sync(fz_ctx* c) {
    if (!c) return fail(-1, "fz_sync: ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int fz_timing_reset(fz_ctx* c) { if (!c) return fail(-1, "ctx is NULL"); c->tm = fz_timing{}; for (auto& v : c->mc_info) v = 0; return 0; }
extern "C" int fz_timing_get(fz_ctx* c, fz_timing* out) {
    if (!c || !out) return fail(-1, "fz_timing_get: NULL argument");
    *out = c->tm; return 0;
}
// page-locked host memory for results (the drop-in classes return PDFs in it: a device-to-host copy into pageable memory is
// staged by the runtime and runs at a third of the link rate)
extern "C" int fz_host_alloc(int64_t bytes, void** out) {
    if (!out || bytes <= 0) return fail(-1, "fz_host_alloc: bad argument");
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(-2, "hipHostMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e)); }
    *out = p;
    return 0;
}
extern "C" int fz_host_free(void* p) {
    if (!p) return 0;
    HIPCHK(hipHostFree(p));
    return 0;
}
// mode C since the last fz_timing_reset: {objects re-run by the IEEE solve because their error came within rounding of ltol, iterations of the
// slowest object, 1 = one block per object (k_modec_persist) / 2 = state planes (k_modec_step / _check), threads per block of the
// persistent shape}
extern "C" int fz_modec_info(fz_ctx* c, int64_t* out4) {
    if (!c || !out4) return fail(-1, "fz_modec_info: NULL argument");
    for (int k = 0; k < 4; ++k) out4[k] = c->mc_info[k];
    return 0;
}
// iterations of pdf.py:199's loop each object of the LAST mode-C chunk took (the reference has no such output; tests compare it with
// the restated loop's count): n <= the chunk's objects, out int32 (host)
extern "C" int fz_modec_niter(fz_ctx* c, int64_t n, int32_t* out) {
    if (!c || !out) return fail(-1, "fz_modec_niter: NULL argument");
    if (n < 0 || n > c->mc_niter_n) return fail(-1, "fz_modec_niter: %lld objects asked, the last mode-C chunk held %lld", (long long)n, (long long)c->mc_niter_n);
    if (!n) return 0;
    HIPCHK(hipSetDevice(c->device));
    return copy_out(c, out, c->d_mcniter.p, (size_t)n * 4);
}
extern "C" int fz_set_workspace_limit(fz_ctx* c, int64_t bytes) {
    if (!c || bytes < (1 << 20)) return fail(-1, "fz_set_workspace_limit: need >= 1 MiB");
    c->ws_limit = bytes; return 0;
}

// ---------------------------------------------------------------------------
// models  (BruteForce.__init__, bruteforce.py:36-64)
// ---------------------------------------------------------------------------
// flags: bit0 = some mask entry is 0, bit1 = some mask entry is neither 0 nor 1,
// bit2 = some value is outside the range the reciprocal-based fast arithmetic accepts,
// bit3 = the model errors of some band differ between models, bit4 = some observed model flux is beyond 1e9
__global__ void k_prep_models(const double* y, const double* ye, const double* ym, int64_t M, int64_t Mp,
                              int B, int BT, double* sy, double* sye2, double* sye, uint32_t* bits, int* flags,
                              double* rec0, int rw0, double* rec1, int rw1) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Mp) return;
    uint32_t bt = 0; int fl = 0;
    double ysum = 0.0;
    for (int b = 0; b < BT; ++b) {
        double vy = 0.0, ve2 = 0.0, ve = 0.0;
        if (j < M && b < B) {
            vy = y[j * B + b];
            ysum += fabs(vy);
            const double e = ye[j * B + b];
            ve2 = e * e; ve = e;
            if (!(e == ye[b])) fl |= 8;                        // vs model 0 (nan counts as different)
            const double mk = ym[j * B + b];
            if (mk != 0.0) bt |= 1u << b; else fl |= 1;
            if (mk != 0.0 && mk != 1.0) fl |= 2;
            if (!(ve2 == 0.0 || (ve2 > 1e-30 && ve2 < 1e30)) || !(fabs(vy) < 1e30)) fl |= 4;
            if (mk != 0.0 && !(fabs(vy) < 1e9)) fl |= 16;           // (the segmented kernel's bound on an unobserved object band's term, fz_hist.h)
        } else if (j >= M) { vy = 1.0; ve2 = 1.0; ve = 1.0; }
        sy[(int64_t)b * Mp + j] = vy;
        sye2[(int64_t)b * Mp + j] = ve2;
        sye[(int64_t)b * Mp + j] = ve;
        rec0[j * rw0 + b] = vy; rec0[j * rw0 + BT + b] = ve2;
        rec1[j * rw1 + b] = vy;
    }
    for (int b = 2 * BT; b < rw0; ++b) rec0[j * rw0 + b] = 0.0;
    for (int b = BT; b < rw1; ++b) rec1[j * rw1 + b] = 0.0;
    bits[j] = bt;
    // an all-zero model makes the free-scale solve 0/0 (nan scale and ln-like, as in the reference):
    // such sets take the IEEE variant, which keeps the nan bookkeeping the fast variants drop
    if (j < M && !(ysum > 0.0)) fl |= 4;
    if (fl) atomicOr(flags, fl);
}

static int pick_bt(int B) { return (B >= 4 && B <= 8) ? B : (B < 4 ? 4 : (B <= 12 ? 12 : (B <= 16 ? 16 : (B <= 24 ? 24 : (B <= 32 ? 32 : 0))))); }

extern "C" int fz_models_upload(fz_ctx* c, const double* y, const double* ye, const double* ym, int64_t M, int32_t B) {
    if (!c || !y || !ye || !ym) return fail(-1, "fz_models_upload: NULL argument");
    if (M <= 0 || B <= 0) return fail(-1, "fz_models_upload: bad shape (%lld,%d)", (long long)M, B);
    const int BT = pick_bt(B);
    if (!BT) return fail(-5, "fz_models_upload: %d bands unsupported (max 32)", B);
    HIPCHK(hipSetDevice(c->device));
    c->probe_share = -1.0; c->probe_pending = false;  // what was sampled belongs to the previous model set
    const int64_t Mp = (M + FZ_MP_ALIGN - 1) / FZ_MP_ALIGN * FZ_MP_ALIGN;      // whole tiles of every kernel (fz_ctx.h)
    const size_t raw = (size_t)M * B * sizeof(double);
    FZCHK(c->d_rx.ensure(raw)); FZCHK(c->d_rxe.ensure(raw)); FZCHK(c->d_rxm.ensure(raw));
    FZCHK(copy_in(c, c->d_rx.p, y, raw)); FZCHK(copy_in(c, c->d_rxe.p, ye, raw)); FZCHK(copy_in(c, c->d_rxm.p, ym, raw));
    FZCHK(c->d_y.ensure((size_t)BT * Mp * 8)); FZCHK(c->d_ye2.ensure((size_t)BT * Mp * 8)); FZCHK(c->d_ye.ensure((size_t)BT * Mp * 8));
    FZCHK(c->d_mbits.ensure((size_t)Mp * 4)); FZCHK(c->d_flags.ensure(64));
    const int rw0 = fz_rec_width(2 * BT), rw1 = fz_rec_width(BT);
    FZCHK(c->d_rec0.ensure((size_t)Mp * rw0 * 8)); FZCHK(c->d_rec1.ensure((size_t)Mp * rw1 * 8));
    HIPCHK(hipMemsetAsync(c->d_flags.p, 0, 64, c->stream));
    {
        Timer t(c, &c->tm.ms_other, &c->tm.n_other);
        hipLaunchKernelGGL(k_prep_models, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_rx.as<double>(), c->d_rxe.as<double>(), c->d_rxm.as<double>(), M, Mp, (int)B, BT,
                           c->d_y.as<double>(), c->d_ye2.as<double>(), c->d_ye.as<double>(), c->d_mbits.as<uint32_t>(), c->d_flags.as<int>(),
                           c->d_rec0.as<double>(), rw0, c->d_rec1.as<double>(), rw1);
    }
    HIPCHK(hipGetLastError());
    int fl = 0;
    FZCHK(copy_out(c, &fl, c->d_flags.p, sizeof fl));
    if (fl & 2) return fail(-4, "models_mask must be binary (0/1)");
    c->M = M; c->Mp = Mp; c->B = B; c->BT = BT;
    c->mc_rec0_valid = c->mc_rec1_valid = false;
    c->seg_state = 0; c->seg_rec0_valid = c->seg_rec1_valid = false;
    c->h_mbits.resize((size_t)M);
    FZCHK(copy_out(c, c->h_mbits.data(), c->d_mbits.p, (size_t)M * 4));
    c->models_masked = (fl & 1) || (BT != B);
    c->models_real_masked = (fl & 1) != 0;
    c->models_wild = (fl & 4) != 0;
    c->models_big = (fl & 16) != 0;
    // band-constant model errors (zeros for a template grid, a common floor, the SURVEY 8d
    // configurations): xe^2 + ye^2 does not depend on the model, so it is formed once per object
    // and mode A runs on the kernels of mode Ai (see obj_vmode); FZ_NO_ERRCONST=1 disables this
    c->models_err_const = !(fl & 8) && !fz_dbg("FZ_NO_ERRCONST");
    {
        std::vector<double> e0(B), e2(BT, 0.0);
        FZCHK(copy_out(c, e0.data(), c->d_rxe.p, (size_t)B * 8));
        for (int b = 0; b < B; ++b) e2[b] = e0[b] * e0[b];
        FZCHK(c->d_ye2c.ensure((size_t)BT * 8));
        FZCHK(copy_in(c, c->d_ye2c.p, e2.data(), (size_t)BT * 8));
    }
    // gammaln(a) + a ln2 tables: a = n/2 (pdf.py:91-93) and a = (n-1)/2 (pdf.py:227-229)
    std::vector<double> ta(BT + 1), tb(BT + 1);
    for (int n = 0; n <= BT; ++n) {
        const double a = 0.5 * n, b = 0.5 * (n - 1);
        ta[n] = std::lgamma(a) + a * FZ_LN2;
        tb[n] = std::lgamma(b) + b * FZ_LN2;
    }
    FZCHK(c->d_lgA.ensure((BT + 1) * 8)); FZCHK(c->d_lgB.ensure((BT + 1) * 8));
    FZCHK(copy_in(c, c->d_lgA.p, ta.data(), (BT + 1) * 8));
    FZCHK(copy_in(c, c->d_lgB.p, tb.data(), (BT + 1) * 8));
    c->label_mode = 0;      // labels belong to a model set
    return 0;
}

// ---------------------------------------------------------------------------
// KDE dictionary and labels  (pdf.py:800-819, 843-852)
// ---------------------------------------------------------------------------
extern "C" int fz_kdedict_upload(fz_ctx* c, int64_t G, int64_t D, const int64_t* widths, const int64_t* offsets,
                                 const double* kern, const double* kcdf) {
    if (!c || !widths || !offsets || !kern || !kcdf) return fail(-1, "fz_kdedict_upload: NULL argument");
    if (G <= 0 || D <= 0) return fail(-1, "fz_kdedict_upload: bad sizes");
    HIPCHK(hipSetDevice(c->device));
    c->h_widths.assign(widths, widths + D);
    c->h_offsets.assign(offsets, offsets + D + 1);
    const int64_t tot = offsets[D];
    c->h_kcdf.assign(kcdf, kcdf + tot);
    FZCHK(c->d_widths.ensure(D * 8)); FZCHK(c->d_offsets.ensure((D + 1) * 8)); FZCHK(c->d_kern.ensure(tot * 8));
    FZCHK(copy_in(c, c->d_widths.p, widths, D * 8));
    FZCHK(copy_in(c, c->d_offsets.p, offsets, (D + 1) * 8));
    FZCHK(copy_in(c, c->d_kern.p, kern, tot * 8));
    c->G = G; c->dict_G = G; c->D = D;
    c->label_mode = 0;
    return 0;
}

extern "C" int fz_labels_upload_dict(fz_ctx* c, const int64_t* y_idx, const int64_t* y_std_idx, int64_t M) {
    if (!c || !y_idx || !y_std_idx) return fail(-1, "fz_labels_upload_dict: NULL argument");
    if (!c->D) return fail(-1, "fz_labels_upload_dict: upload the dictionary first");
    if (M <= 0) return fail(-1, "fz_labels_upload_dict: M <= 0");
    HIPCHK(hipSetDevice(c->device));
    c->probe_share = -1.0; c->probe_pending = false;  // what was sampled belongs to the previous model set
    const int64_t Mp = (M + FZ_MP_ALIGN - 1) / FZ_MP_ALIGN * FZ_MP_ALIGN;      // whole tiles of every kernel (fz_ctx.h)
    std::vector<int64_t> hy(M), hs(M);
    if (is_device_ptr(y_idx)) {
        HIPCHK(hipMemcpy(hy.data(), y_idx, M * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hs.data(), y_std_idx, M * 8, hipMemcpyDeviceToHost));
    } else { memcpy(hy.data(), y_idx, M * 8); memcpy(hs.data(), y_std_idx, M * 8); }
    std::vector<int32_t> pos(Mp, 0), cls(Mp, 0);
    std::vector<double> nrm(Mp, 1.0);
    // the grid length is the DICTIONARY's: fz_labels_upload_grid leaves its own grid's length in c->G
    const int64_t G = c->G = c->dict_G;
    bool single = true;
    for (int64_t j = 0; j < M; ++j) {
        const int64_t d = hs[j], p = hy[j];
        if (d < 0 || d >= c->D) return fail(-3, "label %lld: y_std_idx %lld outside the dictionary", (long long)j, (long long)d);
        const int64_t w = c->h_widths[d], len = c->h_offsets[d + 1] - c->h_offsets[d];
        if (len != 2 * w + 1)
            return fail(-4, "label %lld maps onto dictionary entry %lld whose kernel is malformed "
                        "(width %lld > Ngrid//2, pdf.py:814-816)", (long long)j, (long long)d, (long long)w);
        // window [p-w, p+w] must overlap the grid (else pdf.py:612-620 raises IndexError / shape errors)
        if (p + w < 0 || p - w > G - 1)
            return fail(-3, "label %lld: grid index %lld with kernel half-width %lld lies off the grid", (long long)j, (long long)p, (long long)w);
        const int64_t lo = std::max<int64_t>(p - w, 0), hi = std::min<int64_t>(p + w + 1, G);
        const int64_t lpad = lo - (p - w), hpad = hi - (p + w + 1);
        const double* cdf = c->h_kcdf.data() + c->h_offsets[d];
        // kcdf[hpad-1] with Python negative indexing -> element len+hpad-1   (pdf.py:613-617)
        double mass = cdf[len + hpad - 1];
        if (lpad != 0) mass -= cdf[lpad - 1];
        pos[j] = (int32_t)p; cls[j] = (int32_t)d; nrm[j] = mass;
        if (d != hs[0]) single = false;
    }
    FZCHK(c->d_pos.ensure(Mp * 4)); FZCHK(c->d_cls.ensure(Mp * 4)); FZCHK(c->d_norm.ensure(Mp * 8));
    FZCHK(copy_in(c, c->d_pos.p, pos.data(), Mp * 4));
    FZCHK(copy_in(c, c->d_cls.p, cls.data(), Mp * 4));
    FZCHK(copy_in(c, c->d_norm.p, nrm.data(), Mp * 8));
    c->single_cls = single; c->cls0 = (int32_t)hs[0]; c->w0 = (int32_t)c->h_widths[hs[0]];
    if (single) {
        // one dictionary kernel for every label: the edge-truncated mass (pdf.py:613-617) is a function of the
        // grid index alone -- a table over the padded histogram row [0, G + 2 w0), entry q <-> index q - w0
        const int64_t d = hs[0], w = c->h_widths[d], len = c->h_offsets[d + 1] - c->h_offsets[d];
        const double* cdf = c->h_kcdf.data() + c->h_offsets[d];
        std::vector<double> tab((size_t)(G + 2 * w), 1.0);
        for (int64_t q = 0; q < G + 2 * w; ++q) {
            const int64_t pp = q - w;
            if (pp + w < 0 || pp - w > G - 1) continue;
            const int64_t lo = std::max<int64_t>(pp - w, 0), hi = std::min<int64_t>(pp + w + 1, G);
            const int64_t lpad = lo - (pp - w), hpad = hi - (pp + w + 1);
            double mass = cdf[len + hpad - 1];
            if (lpad != 0) mass -= cdf[lpad - 1];
            tab[(size_t)q] = (mass > 0.0) ? mass : 1.0;
        }
        FZCHK(c->d_normtab.ensure(tab.size() * 8));
        FZCHK(copy_in(c, c->d_normtab.p, tab.data(), tab.size() * 8));
    }
    // many dictionary widths: tables of the class-sorted stack (k_fused MC) -- the kernel's copy of the model records
    // is ordered by class, so every object's candidate list comes out grouped by class and the PDF stage can keep ONE
    // histogram, convolved with its class's kernel whenever the class changes.  Limits: half-widths <= 63 (the taps
    // sit in two registers across the wave), G <= 768 (the result row sits in 12 registers per lane).
    c->mc_ok = false; c->mc_rec0_valid = c->mc_rec1_valid = false;
    if (!single) {
        std::vector<int32_t> present((size_t)c->D, 0);
        int64_t W0 = 0;
        for (int64_t j = 0; j < M; ++j) { present[hs[j]] = 1; W0 = std::max<int64_t>(W0, c->h_widths[hs[j]]); }
        const int64_t Gp = G + 2 * W0;
        if (W0 <= 63 && G <= 768 && Gp < 1024) {
            std::vector<int32_t> rank((size_t)c->D, -1), rwidth; std::vector<int64_t> roff, start;
            for (int64_t d = 0; d < c->D; ++d) if (present[d]) { rank[d] = (int32_t)rwidth.size(); rwidth.push_back((int32_t)c->h_widths[d]); roff.push_back(c->h_offsets[d]); }
            const size_t C = rwidth.size();
            start.assign(C + 1, 0);
            for (int64_t j = 0; j < M; ++j) ++start[rank[hs[j]] + 1];
            for (size_t r = 0; r < C; ++r) start[r + 1] += start[r];
            std::vector<int32_t> perm((size_t)M), tag((size_t)Mp, 0);
            for (int64_t j = 0; j < M; ++j) {                    // stable: models keep their order inside a class
                const int32_t r = rank[hs[j]];
                const int64_t nj = start[r]++;
                perm[nj] = (int32_t)j; tag[nj] = (r << 10) | (int32_t)(hy[j] + W0);
            }
            std::vector<double> ntab(C * (size_t)Gp, 1.0);
            size_t r = 0;
            for (int64_t d = 0; d < c->D; ++d) {
                if (!present[d]) continue;
                const int64_t w = c->h_widths[d], len = c->h_offsets[d + 1] - c->h_offsets[d];
                const double* cdf = c->h_kcdf.data() + c->h_offsets[d];
                for (int64_t q = 0; q < Gp; ++q) {
                    const int64_t pp = q - W0;
                    if (pp + w < 0 || pp - w > G - 1) continue;
                    const int64_t lo = std::max<int64_t>(pp - w, 0), hi = std::min<int64_t>(pp + w + 1, G);
                    const int64_t lpad = lo - (pp - w), hpad = hi - (pp + w + 1);
                    double mass = cdf[len + hpad - 1];
                    if (lpad != 0) mass -= cdf[lpad - 1];
                    ntab[r * Gp + q] = mass > 0.0 ? mass : 1.0;       // a zero / underflowed mass: no model of this class stacks at q (the labels were checked), the row entry stays 0
                }
                ++r;
            }
            FZCHK(c->d_mc_tag.ensure(Mp * 4)); FZCHK(c->d_mc_perm.ensure(M * 4)); FZCHK(c->d_mc_width.ensure(C * 4));
            FZCHK(c->d_mc_off.ensure(C * 8)); FZCHK(c->d_mc_norm.ensure(ntab.size() * 8));
            FZCHK(copy_in(c, c->d_mc_tag.p, tag.data(), Mp * 4)); FZCHK(copy_in(c, c->d_mc_perm.p, perm.data(), M * 4));
            FZCHK(copy_in(c, c->d_mc_width.p, rwidth.data(), C * 4)); FZCHK(copy_in(c, c->d_mc_off.p, roff.data(), C * 8));
            FZCHK(copy_in(c, c->d_mc_norm.p, ntab.data(), ntab.size() * 8));
            {   // ... and its reciprocals (k_hist's class flush multiplies, fz_hist.h)
                std::vector<double> rtab(ntab.size());
                for (size_t q = 0; q < ntab.size(); ++q) rtab[q] = 1.0 / ntab[q];
                FZCHK(c->d_mc_rnorm.ensure(rtab.size() * 8));
                FZCHK(copy_in(c, c->d_mc_rnorm.p, rtab.data(), rtab.size() * 8));
            }
            c->mc_ok = true; c->mc_gp = (int32_t)Gp; c->mc_w0 = (int32_t)W0;
        }
    }
    c->label_mode = 1; c->label_M = M;
    c->h_pos.assign(pos.begin(), pos.begin() + M); c->h_cls.assign(cls.begin(), cls.begin() + M);
    c->seg_state = 0; c->seg_rec0_valid = c->seg_rec1_valid = false;
    return 0;
}

// per-model window and in-window Gaussian mass of the direct KDE (pdf.py:499-502, 520-522)
// lrec (even grids): per model {label, 1 / std, 1 / (mass sqrt(2 pi) std) (0: empty window), e^{-(step / std)^2}, lo | hi << 32, -}: what a
// window add needs, in one 48-byte record (three 16-byte loads, no divisions and one exponential less per stacked model)
__global__ void k_prep_grid_labels(const double* y, const double* ystd, int64_t M, const double* grid, int G,
                                   double dx, double sig, int32_t* lo, int32_t* hi, double* nrm, int* flags, double gstep, double* lrec) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double cd = (y[j] - grid[0]) / dx, od = sig * ystd[j] / dx;
    // np.array(..., dtype='int') truncates toward zero
    const long long cen = (long long)cd, off = (long long)od;
    long long up = cen + off, dn = cen - off;
    if (!(cd == cd) || !(od == od) || fabs(cd) > 1e15 || fabs(od) > 1e15) { atomicOr(flags, 1); up = 0; dn = 0; }
    if (up > G) up = G;
    if (dn < 0) dn = 0;
    if (up < 0) { atomicOr(flags, 2); }                // x[dn:up] with a negative upper bound wraps around in Python: not reproduced
    if (dn > G) dn = G;                                // window wholly above the grid: an empty slice, sum 0, kernel skipped (pdf.py:519-523)
    if (up < dn) up = dn;
    lo[j] = (int32_t)dn; hi[j] = (int32_t)up;
    const double mu = y[j], sd = ystd[j], gn = 2.5066282746310002 * sd;
    double tot = 0.0;                                  // Python sum(): sequential
    for (long long t = dn; t < up; ++t) {
        const double z = (grid[t] - mu) / sd;
        tot += exp(-0.5 * (z * z)) / gn;
    }
    nrm[j] = tot;
    if (gstep > 0.0) {
        const double isd = 1.0 / sd, h = gstep * isd;
        double* r = lrec + 6 * j;
        r[0] = mu; r[1] = isd; r[2] = (tot != 0.0) ? 1.0 / (tot * gn) : 0.0; r[3] = exp(-(h * h));
        r[4] = __hiloint2double((int)up, (int)dn); r[5] = 0.0;
    }
}

extern "C" int fz_labels_upload_grid(fz_ctx* c, const double* y, const double* ystd, int64_t M, const double* grid,
                                     int64_t G, double dx, double sig_thresh) {
    if (!c || !y || !ystd || !grid) return fail(-1, "fz_labels_upload_grid: NULL argument");
    if (M <= 0 || G <= 1) return fail(-1, "fz_labels_upload_grid: bad sizes");
    HIPCHK(hipSetDevice(c->device));
    c->probe_share = -1.0; c->probe_pending = false;  // what was sampled belongs to the previous model set
    const int64_t Mp = (M + FZ_MP_ALIGN - 1) / FZ_MP_ALIGN * FZ_MP_ALIGN;      // whole tiles of every kernel (fz_ctx.h)
    FZCHK(c->d_ly.ensure(Mp * 8)); FZCHK(c->d_lstd.ensure(Mp * 8)); FZCHK(c->d_grid.ensure(G * 8));
    FZCHK(c->d_lo.ensure(Mp * 4)); FZCHK(c->d_hi.ensure(Mp * 4)); FZCHK(c->d_norm.ensure(Mp * 8));
    FZCHK(c->d_flags.ensure(64));
    HIPCHK(hipMemsetAsync(c->d_flags.p, 0, 64, c->stream));
    HIPCHK(hipMemsetAsync(c->d_lo.p, 0, Mp * 4, c->stream));
    HIPCHK(hipMemsetAsync(c->d_hi.p, 0, Mp * 4, c->stream));
    HIPCHK(hipMemsetAsync(c->d_norm.p, 0, Mp * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_ly.p, 0, Mp * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_lstd.p, 0, Mp * 8, c->stream));
    FZCHK(copy_in(c, c->d_ly.p, y, M * 8)); FZCHK(copy_in(c, c->d_lstd.p, ystd, M * 8));
    FZCHK(copy_in(c, c->d_grid.p, grid, G * 8));
    // an evenly spaced grid (every point within 1e-9 of a step of x0 + t step) lets the window adds run a two-multiplication
    // recurrence per point (kde_scatter); anything else keeps the table exponential per point
    c->grid_step = 0.0;
    {
        std::vector<double> hg((size_t)G);
        FZCHK(copy_out(c, hg.data(), c->d_grid.p, (size_t)G * 8));
        const double step = (hg[(size_t)G - 1] - hg[0]) / (double)(G - 1);
        bool even = step > 0.0 && (step - step == 0.0);
        for (int64_t t = 0; t < G && even; ++t) even = fabs(hg[(size_t)t] - (hg[0] + (double)t * step)) <= 1e-9 * step;
        // (the recurrence seeds a window with e^{-z^2/2} and e^{-z h - h^2/2}, both clamped at e^{+-700}: with windows of more than ~30
        //  label errors the seed point lies so far in the tail that the product of the two clamped values is O(1) instead of ~0 --
        //  such calls keep the per-point exponential)
        if (even && sig_thresh <= 30.0) c->grid_step = step;
    }
    FZCHK(c->d_lrec.ensure((size_t)Mp * 48));
    HIPCHK(hipMemsetAsync(c->d_lrec.p, 0, (size_t)Mp * 48, c->stream));
    {
        Timer t(c, &c->tm.ms_other, &c->tm.n_other);
        hipLaunchKernelGGL(k_prep_grid_labels, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_ly.as<double>(), c->d_lstd.as<double>(), M, c->d_grid.as<double>(), (int)G, dx,
                           sig_thresh, c->d_lo.as<int32_t>(), c->d_hi.as<int32_t>(), c->d_norm.as<double>(),
                           c->d_flags.as<int>(), c->grid_step, c->d_lrec.as<double>());
    }
    HIPCHK(hipGetLastError());
    int fl = 0;
    FZCHK(copy_out(c, &fl, c->d_flags.p, sizeof fl));
    if (fl & 1) return fail(-4, "gauss_kde labels: non-finite label or label error");
    if (fl & 2) return fail(-3, "gauss_kde labels: a label's window lies wholly below the grid "
                                "(the reference's negative-index slicing there is not reproduced)");
    c->G = G; c->label_mode = 2; c->label_M = M; c->mc_ok = false;
    c->seg_state = 0; c->seg_rec0_valid = c->seg_rec1_valid = false;
    return 0;
}

// ---------------------------------------------------------------------------
// segmented model layout of the one-pass kernel (fz_hist.h, SEG): pdf.py:76-87 with models_mask
// ---------------------------------------------------------------------------
// segment-ordered copy of the model records: row j' <- row perm[j'], pad slots (perm < 0) benign (y = ye^2 = 1)
// and the masked bands of a model ZERO (flux and error): the mask-free arithmetic of the segmented kernel then finds d = 0 - 0 there
static __global__ void k_seg_records(const double* __restrict__ in, const int* __restrict__ perm, const uint32_t* __restrict__ mbits, int64_t Ms,
                                     int rw, int nval, int BT, double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Ms * rw) return;
    const int64_t j = e / rw; const int r = (int)(e - j * rw);
    const int src = perm[j];
    double v = src >= 0 ? in[(int64_t)src * rw + r] : (r < nval ? 1.0 : 0.0);
    if (src >= 0 && r < nval && !((mbits[src] >> (r % BT)) & 1u)) v = 0.0;
    out[e] = v;
}
int fz_segments(fz_ctx* c, bool rec0) {
    if (c->seg_state < 0) return 1;
    if (c->seg_state == 0) {
        c->seg_state = -1;
        const int64_t M = c->M;
        if (c->label_mode != 1 || c->label_M != M || (int64_t)c->h_mbits.size() != M || (int64_t)c->h_cls.size() != M) return 1;
        // dictionary classes present, in dictionary order (the ranks of the class-sorted stack, fz_labels_upload_dict)
        std::vector<int32_t> rank((size_t)c->D, -1);
        int32_t C = 0;
        {
            std::vector<char> present((size_t)c->D, 0);
            for (int64_t j = 0; j < M; ++j) present[c->h_cls[j]] = 1;
            for (int64_t d = 0; d < c->D; ++d) if (present[d]) rank[d] = C++;
        }
        if (C > 1 && !c->mc_ok) return 1;                        // (tables of the class-sorted stack: half-widths <= 63, G <= 768)
        const int64_t W0 = C > 1 ? c->mc_w0 : c->w0;
        if (c->G + 2 * W0 > 65535) return 1;                     // the candidate buffers hold 16-bit label indices
        std::vector<int32_t> order((size_t)M);
        for (int64_t j = 0; j < M; ++j) order[j] = (int32_t)j;
        auto key = [&](int32_t j) { return ((uint64_t)(uint32_t)rank[c->h_cls[j]] << 32) | c->h_mbits[j]; };
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key(a) < key(b); });
        std::vector<uint32_t> smask; std::vector<int32_t> srank, sstart, perm, tag;
        perm.reserve((size_t)M + 4096); tag.reserve((size_t)M + 4096);
        for (int64_t a = 0; a < M;) {
            int64_t b = a;
            while (b < M && key(order[b]) == key(order[a])) ++b;
            const int32_t sg = (int32_t)smask.size();
            if (sg >= 16384) return 1;                           // (14 bits of the tag word)
            smask.push_back(c->h_mbits[order[a]]); srank.push_back(rank[c->h_cls[order[a]]]); sstart.push_back((int32_t)perm.size());
            for (int64_t k = a; k < b; ++k) { perm.push_back(order[k]); tag.push_back((int32_t)((c->h_pos[order[k]] + W0) | (sg << 16))); }
            if (perm.size() % 64) {                                  // pad slots: sign bit; every slot of their group: bit 30 (the kernel's tile test reads one)
                while (perm.size() % 64) { perm.push_back(-1); tag.push_back((int32_t)((uint32_t)(sg << 16) | 0x80000000u)); }
                for (size_t k = perm.size() - 64; k < perm.size(); ++k) tag[k] |= 0x40000000;
            }
            a = b;
        }
        sstart.push_back((int32_t)perm.size());
        const int32_t last = (int32_t)smask.size() - 1;
        while (perm.size() % FZ_MP_ALIGN) { perm.push_back(-1); tag.push_back((int32_t)((uint32_t)(last << 16) | 0xc0000000u)); }   // whole tiles (FZ_MAX_TILE)
        const int64_t Ms = (int64_t)perm.size();
        if (Ms > M + M / 4 + 4096) return 1;                     // mask patterns as many as models (wide band sets): padding would dominate
        HIPCHK(hipSetDevice(c->device));
        FZCHK(c->d_seg_perm.ensure((size_t)Ms * 4)); FZCHK(c->d_seg_tag.ensure((size_t)Ms * 4));
        FZCHK(c->d_seg_mask.ensure(smask.size() * 4)); FZCHK(c->d_seg_rank.ensure(srank.size() * 4));
        FZCHK(copy_in(c, c->d_seg_perm.p, perm.data(), (size_t)Ms * 4)); FZCHK(copy_in(c, c->d_seg_tag.p, tag.data(), (size_t)Ms * 4));
        FZCHK(copy_in(c, c->d_seg_mask.p, smask.data(), smask.size() * 4)); FZCHK(copy_in(c, c->d_seg_rank.p, srank.data(), srank.size() * 4));
        FZCHK(c->d_seg_start.ensure(sstart.size() * 4)); FZCHK(copy_in(c, c->d_seg_start.p, sstart.data(), sstart.size() * 4));
        c->seg_Ms = Ms; c->seg_n = (int32_t)smask.size(); c->seg_nrank = C;
        c->seg_rec0_valid = c->seg_rec1_valid = false;
        c->seg_state = 1;
    }
    bool& valid = rec0 ? c->seg_rec0_valid : c->seg_rec1_valid;
    if (!valid) {
        const int rw = rec0 ? fz_rec_width(2 * c->BT) : fz_rec_width(c->BT);
        DevBuf& dst = rec0 ? c->d_seg_rec0 : c->d_seg_rec1;
        FZCHK(dst.ensure((size_t)c->seg_Ms * rw * 8));
        const int64_t tot = c->seg_Ms * rw;
        hipLaunchKernelGGL(k_seg_records, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, (rec0 ? c->d_rec0 : c->d_rec1).as<double>(),
                           c->d_seg_perm.as<int>(), c->d_mbits.as<uint32_t>(), c->seg_Ms, rw, rec0 ? 2 * c->BT : c->BT, c->BT, dst.as<double>());
        HIPCHK(hipGetLastError());
        valid = true;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// objects: clean (pdf.py:309-311) + per-chunk derived arrays
// ---------------------------------------------------------------------------
// vmode 0: v = xe^2 (modes A, C) ; 1: v = 1/xe^2 (modes Ai, B) ; 2: v = 1/(xe^2 + ye2c[b]) (mode A with
// band-constant model errors, evaluated by the mode Ai kernels; slv = sum log(xe^2 + ye2c))
// flags: bit0 some data mask is 0 after cleaning, bit1 some mask non-binary,
// bit2 some value outside the fast arithmetic's range, bit3 some entry was rewritten by the clean
__global__ void k_prep_objects(double* x, double* xe, double* xm, int64_t N, int B, int BT, int vmode,
                               int derive, double* ox, double* ov, uint32_t* bits, double* slv,
                               int* flags, const double* __restrict__ ye2c) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint32_t bt = 0; int fl = 0; double sl = 0.0;
    for (int b = 0; b < BT; ++b) {
        double fx = 0.0, v = 1.0;
        if (b < B) {
            double f = x[i * B + b], e = xe[i * B + b], mk = xm[i * B + b];
            const bool clean = (f - f == 0.0) && (e - e == 0.0) && (e > 0.0);   // isfinite & isfinite & >0
            if (!clean) { f = 0.0; e = 1.0; mk = 0.0; x[i * B + b] = f; xe[i * B + b] = e; xm[i * B + b] = mk; fl |= 8; }
            if (mk != 0.0) bt |= 1u << b; else fl |= 1;
            if (mk != 0.0 && mk != 1.0) fl |= 2;
            const double e2 = (vmode == 2) ? e * e + ye2c[b] : e * e;       // pdf.py:77 tot_var
            if (!(e2 > 1e-30 && e2 < 1e30) || !(fabs(f) < 1e30)) fl |= 4;
            sl += log(e2);
            fx = f;
            // modes Ai / B: an unobserved band carries inverse variance 0 -- it then adds exactly nothing to chi2, inter and shape in the
            // mask-free arithmetic as well (the masked kernels multiply the term by the mask anyway)
            v = (vmode == 0) ? e2 : (mk != 0.0 ? 1.0 / e2 : 0.0);
        }
        if (derive) { ox[i * BT + b] = fx; ov[i * BT + b] = v; }
    }
    if (derive) { bits[i] = bt; slv[i] = sl; }
    if (fl) atomicOr(flags, fl);
}

// Stream-ordering contract of the ABI (include/frankenz_hip.h): the library works on its own
// non-blocking stream.  Arguments in device memory may have been produced by kernels the caller
// queued on ANY stream, so by default every entry point that accepts device pointers first waits
// for the device (hipDeviceSynchronize: ~10 us when idle).  That wait also drains work that has
// nothing to do with the call -- an RCCL all-gather of the previous round's rows in flight on
// another stream -- so a caller that knows where its inputs come from says so
// (fz_set_producer_stream): the library's stream then waits for an event recorded on THAT stream
// (no host wait, nothing else drained), or for nothing at all when the caller has synchronised.
static int wait_for_producers(fz_ctx* c, std::initializer_list<const void*> ptrs) {
    bool dev = false;
    for (const void* p : ptrs)
        if (is_device_ptr(p)) { dev = true; break; }
    if (!dev) return 0;
    if (c->producer_mode == 2) return 0;                          // inputs complete: the caller's word
    if (c->producer_mode == 1) {
        if (!c->ev_producer) HIPCHK(hipEventCreateWithFlags(&c->ev_producer, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->ev_producer, c->producer_stream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_producer, 0));
        return 0;
    }
    HIPCHK(hipDeviceSynchronize());
    return 0;
}
extern "C" int fz_set_producer_stream(fz_ctx* c, void* stream, int32_t mode) {
    if (!c) return fail(-1, "fz_set_producer_stream: ctx is NULL");
    if (mode < 0 || mode > 2) return fail(-1, "fz_set_producer_stream: mode %d not in {0, 1, 2}", mode);
    c->producer_mode = mode;
    c->producer_stream = (mode == 1) ? (hipStream_t)stream : nullptr;
    return 0;
}

// stage a chunk of raw objects on the device (no-op views for device pointers),
// clean + derive, copy the cleaned rows back.  Returns the chunk's mask flags.
struct ObjChunk { double *x, *xe, *xm; bool staged; };
static int prep_chunk(fz_ctx* c, double* x, double* xe, double* xm, int64_t i0, int64_t n, int vmode, bool derive,
                      ObjChunk& ch, int& flags) {
    const int B = c->B, BT = c->BT;
    const size_t raw = (size_t)n * B * 8;
    const bool dev = is_device_ptr(x);
    if (dev != is_device_ptr(xe) || dev != is_device_ptr(xm)) return fail(-1, "x, xe, xm must live in the same memory space");
    if (dev) { ch.x = x + i0 * B; ch.xe = xe + i0 * B; ch.xm = xm + i0 * B; ch.staged = false; }
    else {
        FZCHK(c->d_rx.ensure(raw)); FZCHK(c->d_rxe.ensure(raw)); FZCHK(c->d_rxm.ensure(raw));
        ch.x = c->d_rx.as<double>(); ch.xe = c->d_rxe.as<double>(); ch.xm = c->d_rxm.as<double>(); ch.staged = true;
        FZCHK(copy_in(c, ch.x, x + i0 * B, raw)); FZCHK(copy_in(c, ch.xe, xe + i0 * B, raw)); FZCHK(copy_in(c, ch.xm, xm + i0 * B, raw));
    }
    if (derive) {
        const size_t d = (size_t)n * BT * 8;
        FZCHK(c->d_ox.ensure(d)); FZCHK(c->d_ov.ensure(d));
        FZCHK(c->d_obits.ensure((size_t)n * 4)); FZCHK(c->d_oslv.ensure((size_t)n * 8));
    }
    FZCHK(c->d_flags.ensure(64));
    HIPCHK(hipMemsetAsync(c->d_flags.p, 0, 64, c->stream));
    {
        Timer t(c, &c->tm.ms_other, &c->tm.n_other);
        hipLaunchKernelGGL(k_prep_objects, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ch.x, ch.xe, ch.xm,
                           n, B, BT, vmode, derive ? 1 : 0, c->d_ox.as<double>(), c->d_ov.as<double>(),
                           c->d_obits.as<uint32_t>(), c->d_oslv.as<double>(), c->d_flags.as<int>(), c->d_ye2c.as<double>());
    }
    HIPCHK(hipGetLastError());
    FZCHK(copy_out(c, &flags, c->d_flags.p, sizeof(int)));
    if (ch.staged) {
        FZCHK(copy_out(c, x + i0 * B, ch.x, raw)); FZCHK(copy_out(c, xe + i0 * B, ch.xe, raw)); FZCHK(copy_out(c, xm + i0 * B, ch.xm, raw));
    }
    if (flags & 2) return fail(-4, "data_mask must be binary (0/1)");
    return 0;
}

extern "C" int fz_clean(fz_ctx* c, double* x, double* xe, double* xm, int64_t N, int32_t B) {
    if (!c || !x || !xe || !xm) return fail(-1, "fz_clean: NULL argument");
    if (N <= 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    FZCHK(wait_for_producers(c, {x, xe, xm}));
    const int sB = c->B, sBT = c->BT;
    c->B = B; c->BT = B;
    ObjChunk ch; int fl = 0; int r = 0;
    const int64_t step = 1 << 20;
    for (int64_t i0 = 0; i0 < N && r == 0; i0 += step) r = prep_chunk(c, x, xe, xm, i0, std::min(step, N - i0), 0, false, ch, fl);
    c->B = sB; c->BT = sBT;
    return r;
}

// ---------------------------------------------------------------------------
// dispatch on the band-count template (one translation unit each, fz_inst.hip)
// ---------------------------------------------------------------------------
static int run_planes(fz_ctx* c, int mode, int var, int dp, int64_t n, double* lnl, double* chi2, int64_t* ndim,
                      double* scale, double* serr) {
    switch (c->BT) {
        case 4: return fz_planes_bt4(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
        case 5: return fz_planes_bt5(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
        case 6: return fz_planes_bt6(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
        case 7: return fz_planes_bt7(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
        case 8: return fz_planes_bt8(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
        case 12: return fz_planes_bt12(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
        case 16: return fz_planes_bt16(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
        case 24: return fz_planes_bt24(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
        default: return fz_planes_bt32(c, mode, var, dp, n, lnl, chi2, ndim, scale, serr);
    }
}
static int run_fitpredict(fz_ctx* c, int mode, int var, int dp, int64_t n, const fz_kde_opts* ko, double* lmap, double* levid,
                          double* pdfs) {
    switch (c->BT) {
        case 4: return fz_fitpredict_bt4(c, mode, var, dp, n, ko, lmap, levid, pdfs);
        case 5: return fz_fitpredict_bt5(c, mode, var, dp, n, ko, lmap, levid, pdfs);
        case 6: return fz_fitpredict_bt6(c, mode, var, dp, n, ko, lmap, levid, pdfs);
        case 7: return fz_fitpredict_bt7(c, mode, var, dp, n, ko, lmap, levid, pdfs);
        case 8: return fz_fitpredict_bt8(c, mode, var, dp, n, ko, lmap, levid, pdfs);
        case 12: return fz_fitpredict_bt12(c, mode, var, dp, n, ko, lmap, levid, pdfs);
        case 16: return fz_fitpredict_bt16(c, mode, var, dp, n, ko, lmap, levid, pdfs);
        case 24: return fz_fitpredict_bt24(c, mode, var, dp, n, ko, lmap, levid, pdfs);
        default: return fz_fitpredict_bt32(c, mode, var, dp, n, ko, lmap, levid, pdfs);
    }
}
static int run_modec(fz_ctx* c, int var, int64_t n, const fz_like_opts* o, const int64_t* nbr = nullptr,
                     const int64_t* nnb = nullptr, int W = 0) {
    switch (c->BT) {
        case 4: return fz_modec_bt4(c, var, n, o, nbr, nnb, W);
        case 5: return fz_modec_bt5(c, var, n, o, nbr, nnb, W);
        case 6: return fz_modec_bt6(c, var, n, o, nbr, nnb, W);
        case 7: return fz_modec_bt7(c, var, n, o, nbr, nnb, W);
        case 8: return fz_modec_bt8(c, var, n, o, nbr, nnb, W);
        case 12: return fz_modec_bt12(c, var, n, o, nbr, nnb, W);
        case 16: return fz_modec_bt16(c, var, n, o, nbr, nnb, W);
        case 24: return fz_modec_bt24(c, var, n, o, nbr, nnb, W);
        default: return fz_modec_bt32(c, var, n, o, nbr, nnb, W);
    }
}

static int check_kde_opts(const fz_kde_opts* ko) {
    if (!ko) return fail(-1, "kde options are NULL");
    if (!ko->use_wt_thresh && !(ko->cdf_thresh > 0.0 && ko->cdf_thresh < 1.0))
        return fail(-4, "cdf_thresh must lie in (0, 1)");
    return 0;
}

// CDF-threshold KDE (pdf.py:513-516 / 593-597) from device-resident rows of (ln-)weights
static int run_cdf(fz_ctx* c, int64_t n, int L, int64_t M, const double* rows, const int64_t* nbr, const int64_t* nnb,
                   int is_log, const fz_kde_opts* ko, double* dp, double* dm, double* de) {
    const int64_t savedM = c->M; c->M = M;
    KdeView kv; const int rc = fz_kde_view(c, kv);
    c->M = savedM;
    if (rc) return rc;
    FZCHK(c->d_kv.ensure(sizeof(KdeView)));
    FZCHK(copy_in(c, c->d_kv.p, &kv, sizeof(KdeView)));
    const size_t per_wave = (size_t)std::max(kv.acc_stride, 512) * 8;      // (the selection's 2 x 256-bucket histogram shares the row)
    int wpb = 4;
    while (wpb > 1 && per_wave * wpb > 64 * 1024) wpb >>= 1;
    const size_t lds = per_wave * wpb;
    HIPCHK(hipFuncSetAttribute((const void*)k_kde_cdf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FZCHK(c->d_flags.ensure(64));
    HIPCHK(hipMemsetAsync(c->d_flags.p, 0, 64, c->stream));
    {
        Timer t(c, &c->tm.ms_kde, &c->tm.n_kde);
        hipLaunchKernelGGL(k_kde_cdf, dim3((unsigned)((n + wpb - 1) / wpb)), dim3(wpb * 64), lds, c->stream,
                           c->d_kv.as<KdeView>(), kv.acc_stride, n, L, (int)M, rows, nbr, nnb, is_log, ko->cdf_thresh,
                           ko->normalize, dp, dm, de, c->d_flags.as<int>());
    }
    HIPCHK(hipGetLastError());
    int ef = 0;
    FZCHK(copy_out(c, &ef, c->d_flags.p, sizeof ef));
    if (ef) return fail(-3, "neighbour table entry outside [0, Nmodel) or Nneighbors outside [0, K*k]");
    return 0;
}

static int modec_final(fz_ctx* c, int64_t n, bool masked, const fz_like_opts* o, double* lnl, double* chi2, int64_t* ndim,
                       double* scale, double* serr, const int64_t* nbr = nullptr, const int64_t* nnb = nullptr, int W = 0) {
    SubsetView sub; sub.nbr = nbr; sub.nnb = nnb; sub.W = W;
    const int64_t Mloc = nbr ? W : c->M;
    ModeCState st; st.s = c->d_mc[0].as<double>(); st.l = c->d_mc[1].as<double>(); st.c = c->d_mc[2].as<double>();
    st.sh = c->d_mc[3].as<double>(); st.niter = nullptr; st.err = nullptr; st.firstnan = nullptr; st.list = nullptr; st.list_next = nullptr; st.nactive = nullptr; st.ncur = nullptr; st.last_iter = nullptr;
    const int64_t tot = n * Mloc;
    Timer t(c, &c->tm.ms_modec, &c->tm.n_modec);
    hipLaunchKernelGGL(k_modec_final, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, st, model_view(c), sub,
                       c->d_obits.as<uint32_t>(), masked ? 1 : 0, c->B, o->dim_prior, c->d_lgB.as<double>(), n, Mloc, lnl, chi2,
                       ndim, scale, serr);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// additive ln-prior (fz_prior): the lnprior a user lprob_func returns per object
// (bruteforce.py:193-199), as rows of a (P,M) table
// ---------------------------------------------------------------------------
__global__ void k_prior_check(const int64_t* rows, int64_t n, int64_t P, int* flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (rows[i] < 0 || rows[i] >= P)) atomicExch(flag, 1);
}
// planes: lnprior[i][s] = tab[row(i)][col], lnprob = lnl + lnprior, col = s or nbr[i][s]
// (k-NN subsets, padded entries s >= nnb[i] get -inf like knn.py:815-817).  lnprob may alias lnl.
__global__ void k_prior_add(PriorView pv, const double* lnl, int64_t n, int64_t L, const int64_t* nbr, const int64_t* nnb,
                            double* lnprior, double* lnprob) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n * L) return;
    const int64_t i = k / L, s = k - i * L;
    double p = 0.0;
    if (nbr) {
        if (s < nnb[i]) { if (pv.tab) p = pv.tab[pv.row(i) * pv.ld + nbr[k]]; }
        else p = -INFINITY;
    } else if (pv.tab) p = pv.tab[pv.row(i) * pv.ld + s];
    if (lnprior) lnprior[k] = p;
    if (lnprob) lnprob[k] = lnl[k] + p;
}

// objects of a chunk split by "every band observed" (all B mask bits set) or not
__global__ void k_partition_masked(const uint32_t* __restrict__ bits, int64_t n, uint32_t full, int* __restrict__ fast,
                                   int* __restrict__ masked, int* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if ((bits[i] & full) == full) fast[atomicAdd(&counts[0], 1)] = (int)i;
    else masked[atomicAdd(&counts[1], 1)] = (int)i;
}
struct OmapGuard { fz_ctx* c; ~OmapGuard() { c->omap = nullptr; } };

struct PriorBind {
    const fz_prior* pr = nullptr;
    int kind = 0;              // 0 none, 1 one row for all, 2 row i for object i, 3 rows[i]
    bool tab_dev = false, rows_dev = false;
    const double* tab = nullptr;           // device table when it is resident for the whole call
    int64_t chunk_bytes_per_obj = 0;       // staging a chunk needs per object (kind 2 from host memory)
};
struct PriorGuard { fz_ctx* c; ~PriorGuard() { c->prior = PriorView{}; } };

static int prior_begin(fz_ctx* c, const fz_prior* pr, int64_t N, int64_t M, PriorBind& pb) {
    c->prior = PriorView{};
    if (!pr || !pr->table) return 0;
    if (pr->P <= 0 || pr->P >= ((int64_t)1 << 31)) return fail(-4, "ln-prior table: P = %lld rows is out of range", (long long)pr->P);
    pb.pr = pr; pb.tab_dev = is_device_ptr(pr->table); pb.rows_dev = is_device_ptr(pr->rows);
    if (pr->rows) pb.kind = 3;
    else if (pr->P == 1) pb.kind = 1;
    else if (pr->P == N) pb.kind = 2;
    else return fail(-4, "ln-prior table has %lld rows for %lld objects and no row index", (long long)pr->P, (long long)N);
    if (pb.tab_dev) { pb.tab = pr->table; return 0; }
    if (pb.kind == 2) { pb.chunk_bytes_per_obj = M * 8; return 0; }
    const size_t bytes = (size_t)pr->P * M * 8;
    if ((int64_t)bytes > c->ws_limit) return fail(-2, "ln-prior table of %zu bytes exceeds the workspace limit", bytes);
    FZCHK(c->d_ptab.ensure(bytes));
    FZCHK(copy_in(c, c->d_ptab.p, pr->table, bytes));
    pb.tab = c->d_ptab.as<double>();
    return 0;
}
// point c->prior at objects [i0, i0+n)
static int prior_chunk(fz_ctx* c, const PriorBind& pb, int64_t i0, int64_t n, int64_t M) {
    if (!pb.kind) return 0;
    PriorView v{}; v.ld = M;
    if (pb.kind == 1) { v.tab = pb.tab; v.ident = 0; }
    else if (pb.kind == 2) {
        v.ident = 1;
        if (pb.tab_dev) v.tab = pb.tab + i0 * M;
        else { FZCHK(c->d_ptab.ensure((size_t)n * M * 8)); FZCHK(copy_in(c, c->d_ptab.p, pb.pr->table + i0 * M, (size_t)n * M * 8)); v.tab = c->d_ptab.as<double>(); }
    } else {
        v.tab = pb.tab;
        if (pb.rows_dev) v.rows = pb.pr->rows + i0;
        else { FZCHK(c->d_prows.ensure((size_t)n * 8)); FZCHK(copy_in(c, c->d_prows.p, pb.pr->rows + i0, (size_t)n * 8)); v.rows = c->d_prows.as<int64_t>(); }
        FZCHK(c->d_flags.ensure(64));
        HIPCHK(hipMemsetAsync(c->d_flags.p, 0, 64, c->stream));
        hipLaunchKernelGGL(k_prior_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, v.rows, n, pb.pr->P, c->d_flags.as<int>());
        HIPCHK(hipGetLastError());
        int ef = 0;
        FZCHK(copy_out(c, &ef, c->d_flags.p, sizeof ef));
        if (ef) return fail(-3, "ln-prior row index outside [0, %lld)", (long long)pb.pr->P);
    }
    c->prior = v;
    return 0;
}
static int prior_add(fz_ctx* c, const double* lnl, int64_t n, int64_t L, const int64_t* nbr, const int64_t* nnb, double* lnprior,
                     double* lnprob) {
    if (!lnprior && !lnprob) return 0;
    const int64_t tot = n * L;
    Timer t(c, &c->tm.ms_other, &c->tm.n_other);
    hipLaunchKernelGGL(k_prior_add, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->prior, lnl, n, L, nbr, nnb,
                       lnprior, lnprob);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// fz_fit : BruteForce._fit (bruteforce.py:127-205)
// ---------------------------------------------------------------------------
extern "C" int fz_fit_prior(fz_ctx* c, double* x, double* xe, double* xm, int64_t N, const fz_like_opts* o, const fz_prior* pr,
                            double* lnprior, double* lnlike, double* lnprob, double* chi2, int64_t* ndim, double* scale,
                            double* scale_err) {
    if (!c || !x || !xe || !xm || !o) return fail(-1, "fz_fit: NULL argument");
    if (!c->M) return fail(-1, "fz_fit: models have not been uploaded");
    if (N <= 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    FZCHK(wait_for_producers(c, {x, xe, xm, pr ? pr->table : nullptr, pr ? pr->rows : nullptr}));
    const int mode = eff_mode(c, like_mode(o));
    const int64_t M = c->M;
    PriorBind pb; PriorGuard guard{c};
    FZCHK(prior_begin(c, pr, N, M, pb));
    constexpr int NO = 7;
    void* outs[NO] = {lnlike, chi2, ndim, scale, scale_err, lnprior, lnprob};
    const bool need_lnl = lnlike || lnprob;
    bool dev_out[NO]; int nstage = 0;
    for (int k = 0; k < NO; ++k) { dev_out[k] = is_device_ptr(outs[k]); if (outs[k] && !dev_out[k]) ++nstage; }
    if (!lnlike && lnprob) ++nstage;               // scratch ln-like plane behind lnprob
    // chunk so that staging planes (+ mode C state, + a host prior's rows) fit the workspace budget
    const int64_t row = M * 8;
    const int64_t per_obj = row * (nstage + (mode == 3 ? 4 : 0)) + pb.chunk_bytes_per_obj;
    int64_t nc = per_obj ? std::max<int64_t>(1, c->ws_limit / per_obj) : N;
    nc = std::min<int64_t>(std::min<int64_t>(nc, N), 1 << 18);
    for (int64_t i0 = 0; i0 < N; i0 += nc) {
        const int64_t n = std::min(nc, N - i0);
        ObjChunk ch; int fl = 0;
        FZCHK(prep_chunk(c, x, xe, xm, i0, n, obj_vmode(c, like_mode(o)), true, ch, fl));
        FZCHK(prior_chunk(c, pb, i0, n, M));
        const int var = pick_var(c, fl);
        const bool masked = var != VAR_FAST;
        void* dst[NO];
        for (int k = 0; k < NO; ++k) {
            if (!outs[k] && !(k == 0 && need_lnl)) dst[k] = nullptr;
            else if (outs[k] && dev_out[k]) dst[k] = (char*)outs[k] + (size_t)i0 * row;
            else { FZCHK(c->d_pl[k].ensure((size_t)n * row)); dst[k] = c->d_pl[k].p; }
        }
        if (mode == 3) {
            FZCHK(run_modec(c, var, n, o));
            FZCHK(modec_final(c, n, masked, o, (double*)dst[0], (double*)dst[1], (int64_t*)dst[2], (double*)dst[3], (double*)dst[4]));
        } else {
            FZCHK(run_planes(c, mode, var, o->dim_prior, n, (double*)dst[0], (double*)dst[1], (int64_t*)dst[2], (double*)dst[3],
                             (double*)dst[4]));
        }
        FZCHK(prior_add(c, (const double*)dst[0], n, M, nullptr, nullptr, (double*)dst[5], (double*)dst[6]));
        for (int k = 0; k < NO; ++k)
            if (outs[k] && !dev_out[k]) FZCHK(copy_out(c, (char*)outs[k] + (size_t)i0 * row, dst[k], (size_t)n * row));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int fz_fit(fz_ctx* c, double* x, double* xe, double* xm, int64_t N, const fz_like_opts* o, double* lnlike,
                      double* chi2, int64_t* ndim, double* scale, double* scale_err) {
    return fz_fit_prior(c, x, xe, xm, N, o, nullptr, nullptr, lnlike, nullptr, chi2, ndim, scale, scale_err);
}

// ---------------------------------------------------------------------------
// fz_fit_predict : BruteForce._fit_predict, save_fits=False (bruteforce.py:505-631)
// ---------------------------------------------------------------------------
extern "C" int fz_fit_predict(fz_ctx* c, double* x, double* xe, double* xm, int64_t N, const fz_like_opts* o,
                              const fz_kde_opts* ko, double* pdfs, double* lmap, double* levid) {
    return fz_fit_predict_prior(c, x, xe, xm, N, o, ko, nullptr, pdfs, lmap, levid);
}
static int fit_predict_impl(fz_ctx* c, double* x, double* xe, double* xm, int64_t N, const fz_like_opts* o,
                            const fz_kde_opts* ko, const fz_prior* pr, double* pdfs, double* lmap, double* levid);
extern "C" int fz_fit_predict_prior(fz_ctx* c, double* x, double* xe, double* xm, int64_t N, const fz_like_opts* o,
                                    const fz_kde_opts* ko, const fz_prior* pr, double* pdfs, double* lmap, double* levid) {
    if (!c || !x || !xe || !xm || !o || !pdfs) return fail(-1, "fz_fit_predict: NULL argument");
    if (!c->M) return fail(-1, "fz_fit_predict: models have not been uploaded");
    FZCHK(check_kde_opts(ko));
    if (N <= 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    FZCHK(wait_for_producers(c, {x, xe, xm, pr ? pr->table : nullptr, pr ? pr->rows : nullptr}));
    // Host objects: the whole (N, B) x 3 set goes to the device ONCE (120 MB at 1e6 x 5), is cleaned there, and comes back only if the
    // clean changed anything -- staged chunk by chunk, every chunk's six small synchronous copies waited for the previous chunk's kernel
    // and left the GPU idle meanwhile (~5 ms per chunk)
    const size_t raw = (size_t)N * c->B * 8;
    if (!is_device_ptr(x) && !is_device_ptr(xe) && !is_device_ptr(xm) && raw <= ((size_t)1 << 30) && N >= 4096) {
        FZCHK(c->d_sx.ensure(raw)); FZCHK(c->d_sxe.ensure(raw)); FZCHK(c->d_sxm.ensure(raw));
        HIPCHK(hipMemcpyAsync(c->d_sx.p, x, raw, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_sxe.p, xe, raw, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_sxm.p, xm, raw, hipMemcpyHostToDevice, c->stream));
        FZCHK(c->d_flags.ensure(64));
        HIPCHK(hipMemsetAsync(c->d_flags.p, 0, 64, c->stream));
        hipLaunchKernelGGL(k_prep_objects, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, c->d_sx.as<double>(), c->d_sxe.as<double>(),
                           c->d_sxm.as<double>(), N, c->B, c->BT, 0, 0, (double*)nullptr, (double*)nullptr, (uint32_t*)nullptr, (double*)nullptr,
                           c->d_flags.as<int>(), c->d_ye2c.as<double>());
        HIPCHK(hipGetLastError());
        int fl = 0;
        FZCHK(copy_out(c, &fl, c->d_flags.p, sizeof fl));
        if (fl & 2) return fail(-4, "data_mask must be binary (0/1)");
        if (fl & 8) {                                            // pdf.py:310-311 rewrote something: the caller's arrays see it
            FZCHK(copy_out(c, x, c->d_sx.p, raw)); FZCHK(copy_out(c, xe, c->d_sxe.p, raw)); FZCHK(copy_out(c, xm, c->d_sxm.p, raw));
        }
        const int pm = c->producer_mode; c->producer_mode = 2;    // the staged copies are this stream's own work
        const int r = fit_predict_impl(c, c->d_sx.as<double>(), c->d_sxe.as<double>(), c->d_sxm.as<double>(), N, o, ko, pr, pdfs, lmap, levid);
        c->producer_mode = pm;
        return r;
    }
    return fit_predict_impl(c, x, xe, xm, N, o, ko, pr, pdfs, lmap, levid);
}
static int fit_predict_impl(fz_ctx* c, double* x, double* xe, double* xm, int64_t N, const fz_like_opts* o,
                            const fz_kde_opts* ko, const fz_prior* pr, double* pdfs, double* lmap, double* levid) {
    const int mode = eff_mode(c, like_mode(o));
    const int64_t M = c->M, G = c->G;
    if (c->label_mode == 0) return fail(-1, "fz_fit_predict: labels have not been uploaded");
    c->exact_evidence = (o->exact_evidence != 0) || (ko->exact_evidence != 0);
    const bool pdf_dev = is_device_ptr(pdfs), lm_dev = is_device_ptr(lmap), le_dev = is_device_ptr(levid);
    const bool cdf = !ko->use_wt_thresh;          // reference CDF rule: materialise the chunk's ln-like rows
    PriorBind pb; PriorGuard guard{c};
    FZCHK(prior_begin(c, pr, N, M, pb));
    int64_t nc = std::min<int64_t>(N, fz_dbg("FZ_CHUNK") ? atoll(fz_dbg("FZ_CHUNK")) : (1 << 20));   // the fused kernel's workspace does not grow with the chunk
    const int64_t per_obj = M * 8 * (mode == 3 ? 4 : (cdf ? 1 : 0)) + pb.chunk_bytes_per_obj;
    if (per_obj) nc = std::min<int64_t>(nc, std::max<int64_t>(1, c->ws_limit / per_obj));
    // Host PDFs are the bulk of the PCIe traffic of the drop-in call (5.6 GB at 1e6 objects, longer
    // than the kernel).  Pipeline: chunks of 2^17 objects, two device staging buffers, chunk k's rows
    // leave on a second stream while chunk k+1 is being computed; kernel timing is deferred so that
    // the host does not wait on a kernel before it has queued the previous chunk's copy.
    const bool pipe = !pdf_dev && mode != 3 && !cdf && N >= (3 << 17) && !fz_dbg("FZ_NO_PIPELINE");
    if (pipe) nc = std::min<int64_t>(nc, 1 << 17);               // (the last chunk's rows leave with nothing to hide behind: keep it small)
    struct PipeGuard {
        fz_ctx* c; bool on;
        ~PipeGuard() { if (on) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamSynchronize(c->stream); c->defer_timing = false; timer_flush(c); } }
    } pguard{c, pipe};
    if (pipe) {
        c->defer_timing = true;
        if (lmap && !lm_dev) FZCHK(c->d_lmap.ensure((size_t)N * 8));          // whole-run gof buffers: copied once at the end
        if (levid && !le_dev) FZCHK(c->d_levid.ensure((size_t)N * 8));
    }
    int64_t prev_i0 = -1, prev_n = 0; int prev_b = 0;
    auto ship_prev = [&]() -> int {                                // queue the previous chunk's rows behind its kernel
        if (prev_i0 < 0) return 0;
        DevBuf& st = prev_b ? c->d_pdfs2 : c->d_pdfs;
        HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_done[prev_b], 0));
        HIPCHK(hipMemcpyAsync(pdfs + prev_i0 * G, st.p, (size_t)prev_n * G * 8, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(hipEventRecord(c->ev_copied[prev_b], c->copy_stream));
        prev_i0 = -1;
        return 0;
    };
    int64_t kchunk = 0;
    for (int64_t i0 = 0; i0 < N; i0 += nc, ++kchunk) {
        const int64_t n = std::min(nc, N - i0);
        const int bsel = pipe ? (int)(kchunk & 1) : 0;
        ObjChunk ch; int fl = 0;
        FZCHK(prep_chunk(c, x, xe, xm, i0, n, obj_vmode(c, like_mode(o)), true, ch, fl));
        FZCHK(prior_chunk(c, pb, i0, n, M));
        const int var = pick_var(c, fl);
        const bool masked = var != VAR_FAST;
        double* d_pdf; double* d_lm; double* d_le;
        if (pdf_dev) d_pdf = pdfs + i0 * G;
        else {
            DevBuf& st = bsel ? c->d_pdfs2 : c->d_pdfs;
            if (pipe && kchunk >= 2) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_copied[bsel], 0));     // its previous rows have left
            if (st.cap < (size_t)n * G * 8) { if (pipe) HIPCHK(hipStreamSynchronize(c->copy_stream)); FZCHK(st.ensure((size_t)n * G * 8)); }
            d_pdf = st.as<double>();
        }
        if (pipe) {
            d_lm = (lmap && lm_dev) ? lmap + i0 : c->d_lmap.as<double>() + ((lmap && !lm_dev) ? i0 : 0);
            d_le = (levid && le_dev) ? levid + i0 : c->d_levid.as<double>() + ((levid && !le_dev) ? i0 : 0);
            if (!lmap) { FZCHK(c->d_lmap.ensure((size_t)n * 8)); d_lm = c->d_lmap.as<double>(); }
            if (!levid) { FZCHK(c->d_levid.ensure((size_t)n * 8)); d_le = c->d_levid.as<double>(); }
        } else {
        if (lmap && lm_dev) d_lm = lmap + i0; else { FZCHK(c->d_lmap.ensure(n * 8)); d_lm = c->d_lmap.as<double>(); }
        if (levid && le_dev) d_le = levid + i0; else { FZCHK(c->d_levid.ensure(n * 8)); d_le = c->d_levid.as<double>(); }
        }
        if (mode == 3) {
            c->mc_lnl_only = 1;                               // only the final ln-like plane is needed here
            FZCHK(run_modec(c, var, n, o));
            double* lpl = c->d_mc[1].as<double>();
            if (!c->mc_lnl_only) FZCHK(modec_final(c, n, masked, o, lpl, nullptr, nullptr, nullptr, nullptr));   // in place: lnl plane (the one-block-per-object kernel writes it in its final form itself)
            else c->tm.n_modec += 1;                          // (timing counters: the final pass ran inside the iteration kernel; bench.py subtracts two scopes)
            c->mc_lnl_only = 0;
            if (c->prior.tab) FZCHK(prior_add(c, lpl, n, M, nullptr, nullptr, nullptr, lpl));
            if (cdf) {
                FZCHK(run_cdf(c, n, (int)M, M, lpl, nullptr, nullptr, 1, ko, d_pdf, d_lm, d_le));
            } else {
                FZCHK(fz_launch_plane_predict(c, lpl, n, M, 0, ko, d_lm, d_le, d_pdf));
            }
        } else if (cdf) {
            FZCHK(c->d_pl[0].ensure((size_t)n * M * 8));
            FZCHK(run_planes(c, mode, var, o->dim_prior, n, c->d_pl[0].as<double>(), nullptr, nullptr, nullptr, nullptr));
            if (c->prior.tab) FZCHK(prior_add(c, c->d_pl[0].as<double>(), n, M, nullptr, nullptr, nullptr, c->d_pl[0].as<double>()));
            FZCHK(run_cdf(c, n, (int)M, M, c->d_pl[0].as<double>(), nullptr, nullptr, 1, ko, d_pdf, d_lm, d_le));
        } else {
            // A chunk with some unobserved bands would run the masked kernels for every object.
            // When the models themselves are unmasked, the (usually large) share of objects with
            // every band observed keeps the mask-free kernels: the chunk is split in two launches.
            bool done = false;
            // real-catalogue inputs (pdf.py:76-87 with models_mask / per-model models_err): masked MODELS in any mode, or objects with
            // unobserved bands against per-model errors -- the one-pass kernel on the segmented model layout (fz_hist.h, SEG); +1: the
            // form does not apply (too many mask patterns, a KDE form it does not take ...) and the chunk goes on as before
            // ... with one dictionary kernel or many (per-model label errors: the layout is ordered by width class first).  Mask-free data
            // with many widths keep k_fused's class-sorted stack (80 against 112 ms per 2.6e10 pairs: the per-class convolutions cost both
            // kernels the same, and k_fused's weight-space loop has no group-by-group tiles at the class boundaries); FZ_HIST_SEG_MC=1
            // sends them here too (tests); FZ_HIST_SEG_FORCE=1 sends mask-free single-width data through the segmented form (tests; the
            // measure of what the form itself costs: +16 % / +25 % / +11 % on the same data in modes Ai / A / B).
            const char* smc = fz_dbg("FZ_HIST_SEG_MC");
            const bool many_widths = c->label_mode == 1 && !c->single_cls;
            const bool masks = c->models_real_masked || var == VAR_MASKED;
            const bool seg_mc = many_widths && c->mc_ok && !(smc && atoi(smc) == 0) && (masks || (smc && atoi(smc) == 1));
            if ((var == VAR_MASKED || var == VAR_FAST) && c->BT == c->B && !c->prior.tab &&
                (many_widths ? seg_mc : (c->models_real_masked || (var == VAR_MASKED && mode == 0) || fz_dbg("FZ_HIST_SEG_FORCE")))) {
                const int r0 = run_fitpredict(c, mode, VAR_SEG, o->dim_prior, n, ko, d_lm, d_le, d_pdf);
                if (r0 < 0) return r0;
                done = (r0 == 0);
            }
            // objects with unobserved bands, unmasked models, modes Ai / B: the one-pass kernel with per-object band counts takes the whole
            // chunk (masked and fully observed objects alike); where it does not apply (+1) the chunk is split as before
            if (!done && var == VAR_MASKED && !c->models_real_masked && mode != 0 && !c->prior.tab) {
                const int r0 = run_fitpredict(c, mode, VAR_OBJMASK, o->dim_prior, n, ko, d_lm, d_le, d_pdf);
                if (r0 < 0) return r0;
                done = (r0 == 0);
            }
            if (!done && var == VAR_MASKED && !c->models_real_masked && (c->BT == c->B || c->BT > 8) && n >= 4096 && !fz_dbg("FZ_NO_SPLIT")) {
                FZCHK(c->d_omap.ensure((size_t)n * 8 + 64));
                int* fast = c->d_omap.as<int>(); int* slow = fast + n; int* counts = slow + n;
                HIPCHK(hipMemsetAsync(counts, 0, 8, c->stream));
                hipLaunchKernelGGL(k_partition_masked, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                                   c->d_obits.as<uint32_t>(), n, (uint32_t)((1ull << c->B) - 1), fast, slow, counts);
                HIPCHK(hipGetLastError());
                int cnt[2] = {0, 0};
                FZCHK(copy_out(c, cnt, counts, sizeof cnt));
                if (cnt[0] >= 1024 && cnt[1] > 0) {
                    OmapGuard og{c};
                    c->omap = fast;
                    const int r1 = run_fitpredict(c, mode, pick_var(c, 0), o->dim_prior, cnt[0], ko, d_lm, d_le, d_pdf);
                    if (r1 < 0) return r1;
                    c->omap = slow;
                    const int r2 = (r1 == 0) ? run_fitpredict(c, mode, VAR_MASKED, o->dim_prior, cnt[1], ko, d_lm, d_le, d_pdf) : 2;
                    if (r2 < 0) return r2;
                    done = (r1 == 0 && r2 == 0);
                }
            }
            if (!done) FZCHK(run_fitpredict(c, mode, var, o->dim_prior, n, ko, d_lm, d_le, d_pdf));
        }
        if (pipe) {
            HIPCHK(hipEventRecord(c->ev_done[bsel], c->stream));          // chunk k is queued ...
            FZCHK(ship_prev());                                            // ... chunk k-1 leaves while it runs
            prev_i0 = i0; prev_n = n; prev_b = bsel;
            continue;
        }
        if (!pdf_dev) FZCHK(copy_out(c, pdfs + i0 * G, d_pdf, (size_t)n * G * 8));
        if (lmap && !lm_dev) FZCHK(copy_out(c, lmap + i0, d_lm, n * 8));
        if (levid && !le_dev) FZCHK(copy_out(c, levid + i0, d_le, n * 8));
    }
    if (pipe) {
        FZCHK(ship_prev());
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipStreamSynchronize(c->copy_stream));
        if (lmap && !lm_dev) FZCHK(copy_out(c, lmap, c->d_lmap.p, (size_t)N * 8));
        if (levid && !le_dev) FZCHK(copy_out(c, levid, c->d_levid.p, (size_t)N * 8));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------------------
// fz_predict_logwt : BruteForce._predict (bruteforce.py:303-372)
// ---------------------------------------------------------------------------
extern "C" int fz_predict_logwt(fz_ctx* c, const double* logwt, int64_t N, int32_t is_log, const fz_kde_opts* ko, double* pdfs,
                                double* lmap, double* levid) {
    if (!c || !logwt || !pdfs) return fail(-1, "fz_predict_logwt: NULL argument");
    FZCHK(check_kde_opts(ko));
    if (c->label_mode == 0) return fail(-1, "fz_predict_logwt: labels have not been uploaded");
    if (N <= 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    FZCHK(wait_for_producers(c, {logwt}));
    c->exact_evidence = ko->exact_evidence != 0;
    const int64_t M = c->label_M, G = c->G;
    const bool in_dev = is_device_ptr(logwt), pdf_dev = is_device_ptr(pdfs), lm_dev = is_device_ptr(lmap), le_dev = is_device_ptr(levid);
    const int linear = is_log ? 0 : 1;
    int64_t nc = in_dev ? N : std::max<int64_t>(1, c->ws_limit / (M * 8));
    nc = std::min<int64_t>(std::min<int64_t>(nc, N), 1 << 18);
    // c->M may be unset when only labels were uploaded (stand-alone KDE); the kernels only need label_M
    const int64_t savedM = c->M; c->M = M;
    int rc = 0;
    for (int64_t i0 = 0; i0 < N && rc == 0; i0 += nc) {
        const int64_t n = std::min(nc, N - i0);
        const double* d_in;
        if (in_dev) d_in = logwt + i0 * M;
        else { if ((rc = c->d_pl[0].ensure((size_t)n * M * 8))) break; if ((rc = copy_in(c, c->d_pl[0].p, logwt + i0 * M, (size_t)n * M * 8))) break; d_in = c->d_pl[0].as<double>(); }
        double* d_pdf; double* d_lm; double* d_le;
        if (pdf_dev) d_pdf = pdfs + i0 * G; else { if ((rc = c->d_pdfs.ensure((size_t)n * G * 8))) break; d_pdf = c->d_pdfs.as<double>(); }
        if (lmap && lm_dev) d_lm = lmap + i0; else { if ((rc = c->d_lmap.ensure(n * 8))) break; d_lm = c->d_lmap.as<double>(); }
        if (levid && le_dev) d_le = levid + i0; else { if ((rc = c->d_levid.ensure(n * 8))) break; d_le = c->d_levid.as<double>(); }
        if (!ko->use_wt_thresh) {
            if ((rc = run_cdf(c, n, (int)M, M, d_in, nullptr, nullptr, is_log ? 1 : 0, ko, d_pdf, d_lm, d_le))) break;
        } else {
            if ((rc = fz_launch_plane_predict(c, d_in, n, M, linear, ko, d_lm, d_le, d_pdf))) break;
        }
        if (!pdf_dev && (rc = copy_out(c, pdfs + i0 * G, d_pdf, (size_t)n * G * 8))) break;
        if (lmap && !lm_dev && (rc = copy_out(c, lmap + i0, d_lm, n * 8))) break;
        if (levid && !le_dev && is_log && (rc = copy_out(c, levid + i0, d_le, n * 8))) break;
    }
    c->M = savedM;
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

#include "fz_knn_host.inc"
#include "fz_summary_host.inc"
#include "fz_net_host.inc"

#ifdef FZ_KM_STATS
extern "C" int fz_debug_kmstats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fz::fz_kmstats), 128 * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[128] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(fz::fz_kmstats), z, 128 * 8) != hipSuccess) return -1; }
    return 0;
}
#endif
