/*
 * frankenz_hip.h -- C ABI of libfrankenz_hip.so: the MI355X (gfx950) engine for
 * frankenz's brute-force photometric likelihood -> weighted Gaussian-KDE PDF path.
 *
 * This is the drop-in boundary.  The reference (joshspeagle/frankenz v0.3.5) is
 * pure Python/NumPy and has no FFI, so each entry point names the reference
 * function (file:line) whose arithmetic it replaces; INTEGRATION.md shows the
 * ctypes stub a frankenz maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; the message is available
 *    from fz_last_error() (thread-local).
 *  - all floating-point data is float64, indices are int64, matrices are
 *    C-contiguous row-major exactly as the reference's NumPy arrays are.
 *  - every array pointer may be a HOST pointer or a DEVICE (hipMalloc'd) pointer;
 *    the library detects which (hipPointerGetAttributes) and stages host arrays
 *    through its own buffers.  Output pointers may be NULL = "not wanted".
 *  - masks are float64 0/1 (the reference's demos use np.ones_like(phot)).
 *  - STREAM ORDERING.  The library runs on a stream of its own.  By default an entry point
 *    that is handed device pointers first waits for ALL work queued on the device so far
 *    (hipDeviceSynchronize), so inputs produced by kernels on any caller stream are
 *    complete before they are read; and every entry point returns only when its own
 *    work has finished, so outputs may be consumed from any stream at once.  No
 *    caller-side synchronisation is needed on either side of a call.  A caller that
 *    overlaps the library with other device work (an RCCL all-gather of the previous
 *    block's rows in flight) names the stream its inputs are produced on instead --
 *    fz_set_producer_stream -- and the library waits for that stream alone.
 *  - one fz_ctx per device; a ctx is not thread-safe.
 */
#ifndef FRANKENZ_HIP_H
#define FRANKENZ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fz_ctx fz_ctx;

/* lprob_kwargs of frankenz.pdf.logprob / loglike (pdf.py:238-240, 326-328). */
typedef struct fz_like_opts {
    int32_t free_scale;       /* pdf.py:313  free_scale        (default 0) */
    int32_t ignore_model_err; /* pdf.py:76   ignore_model_err  (default 0) */
    int32_t dim_prior;        /* pdf.py:90   dim_prior         (default 1) */
    int32_t max_iter;         /* guard for the unbounded loop at pdf.py:199;
                                 <=0 means 10000.  Hitting it is an error.   */
    double  ltol;             /* pdf.py:199  ltol              (default 1e-4) */
    int32_t exact_evidence;   /* extension (no reference counterpart, default 0).  The default kernel of the fused
                                 path (k_hist, every form) forms and sums EVERY weight in fp64 -- the reference's
                                 logsumexp, bruteforce.py:619 -- and this flag changes nothing there.  It matters
                                 only to the older k_fused weight-space body, which serves what k_hist does not
                                 take (ln-prior tables, the direct grid KDE, many dictionary widths, wild values):
                                 0 = that body sums the weights BELOW wt_thresh of the best in fp32 (~1e-9
                                 relative on levid; lmap, PDFs and every stacked weight are fp64 either way),
                                 1 = its all-fp64 ln-space body.                                          */
    int32_t reserved_;        /* keep 0 */
} fz_like_opts;

/* kde_kwargs of gauss_kde / gauss_kde_dict (pdf.py:444-445, 529-531). */
typedef struct fz_kde_opts {
    double  wt_thresh;     /* relative-amplitude threshold (strict >), 1e-3     */
    int32_t use_wt_thresh; /* 1: wt_thresh rule (pdf.py:507-510 / 589-591).
                              0: the reference's CDF rule (wt_thresh=None,
                                 pdf.py:513-516 / 593-597) with cdf_thresh.
                              "no thresholding" is wt_thresh=-inf, use=1.       */
    int32_t normalize;     /* 1: pdf /= pdf.sum() (bruteforce.py:370, 629)      */
    double  cdf_thresh;    /* CDF rule: keep the ascending prefix with
                              cdf <= 1 - cdf_thresh (default 2e-4)              */
    int32_t exact_evidence;/* extension, default 0: as fz_like_opts.exact_evidence, for the calls that take
                              no likelihood options (predict from stored ln-weights): 1 = the whole
                              logsumexp in fp64; 0 = the weights below wt_thresh of the best in fp32  */
    int32_t reserved_;     /* keep 0 */
} fz_kde_opts;

/* accumulated device time per kernel family since fz_timing_reset (HIP events on
 * the context's own stream). */
typedef struct fz_timing {
    double  ms_planes;  int64_t n_planes;   /* materialising fit kernel       */
    double  ms_fused;   int64_t n_fused;    /* single-pass fit_predict kernel */
    double  ms_stats;   int64_t n_stats;    /* pass 1: max + logsumexp        */
    double  ms_kde;     int64_t n_kde;      /* pass 2: threshold + KDE stack  */
    double  ms_modec;   int64_t n_modec;    /* mode-C fixed-point iterations  */
    double  ms_knn;     int64_t n_knn;      /* brute-force top-k search       */
    double  ms_other;   int64_t n_other;    /* prep/clean/transposes          */
} fz_timing;

const char* fz_last_error(void);

/* Test / tuning switches.  The library does not read the environment: every switch that selects an alternative kernel form, a
 * launch geometry or a diagnostic (the "FZ_..." names listed in INTEGRATION.md section 5; none of them changes a result beyond
 * rounding) is set through this ONE call.  spec = "NAME=value;NAME=value;..." REPLACES the whole set (NULL or "": none set).
 * Process-wide; not to be called while another entry point runs.  The Python layer forwards the process's FZ_* environment
 * variables through it before each call (frankenz_amd/_lib.py), which is how the tests and tools/ set them. */
int  fz_debug_opts(const char* spec);
int  fz_device_count(void);

int  fz_ctx_create(int device, fz_ctx** out);
void fz_ctx_destroy(fz_ctx* ctx);
int  fz_sync(fz_ctx* ctx);
int  fz_timing_reset(fz_ctx* ctx);
int  fz_timing_get(fz_ctx* ctx, fz_timing* out);
/* which kernel form the last fused fit_predict / predict launch took ("k_hist<screen>", "k_hist<screen> (per-object band
 * counts)", "k_hist<exact>", "k_fused", "k_plane_rows", "k_plane_fused", "k_stats + k_kde", ...): the choice depends on the data (likelihood mode, masks, label errors, how broad the
 * likelihoods are), and a measurement should say what it measured.  No reference counterpart. */
const char* fz_last_form(fz_ctx* ctx);
/* byte budget for internal work space (candidate lists of the single-pass kernel, (N x M)
 * planes of mode C, host staging); default 45 % of the device memory, allocated on demand. */
int  fz_set_workspace_limit(fz_ctx* ctx, int64_t bytes);
/* Where device-resident INPUTS of the following calls come from (no reference counterpart: the reference has no device).
 * mode 0 (default): unknown -- every entry point handed device pointers drains the whole device first (hipDeviceSynchronize).
 * mode 1: they are produced by work queued on `stream` (a hipStream_t; NULL = the legacy default stream): the library records an
 *         event there and makes ITS stream wait for it -- no host wait, and work on other streams (a collective in flight) is
 *         not waited for.
 * mode 2: they are complete (the caller has synchronised): no wait at all.
 * Outputs are complete when a call returns in every mode. */
int  fz_set_producer_stream(fz_ctx* ctx, void* stream, int32_t mode);
/* page-locked host memory (hipHostMalloc) for results: device-to-host copies into it run at the link rate and overlap the next
 * chunk's kernel, which copies into pageable memory (np.zeros) do not.  The drop-in classes return their PDF arrays in it. */
int  fz_host_alloc(int64_t bytes, void** out);
int  fz_host_free(void* p);

/* BruteForce.__init__ (bruteforce.py:36-64): the model set (M,B) x3. */
int  fz_models_upload(fz_ctx* ctx, const double* models, const double* models_err,
                      const double* models_mask, int64_t M, int32_t B);

/* PDFDict tables (pdf.py:800-819): Ngrid, Ndict, sigma_width[D], ragged kernels
 * and their running sums flattened with offsets[D+1] (lengths are as built by
 * the reference, malformed entries included; they are rejected only if a label
 * maps onto one). */
int  fz_kdedict_upload(fz_ctx* ctx, int64_t G, int64_t D, const int64_t* widths,
                       const int64_t* offsets, const double* kern, const double* kcdf);

/* labels already mapped by PDFDict.fit (pdf.py:843-852) -> gauss_kde_dict path
 * (pdf.py:599-620).  Returns -3 (IndexError semantics) if a label's window lies
 * wholly off the grid, -4 (ValueError) if it maps onto a malformed entry. */
int  fz_labels_upload_dict(fz_ctx* ctx, const int64_t* y_idx, const int64_t* y_std_idx,
                           int64_t M);
/* raw labels for the direct gauss_kde path (pdf.py:489-502, 519-524). */
int  fz_labels_upload_grid(fz_ctx* ctx, const double* y, const double* y_std, int64_t M,
                           const double* grid, int64_t G, double dx, double sig_thresh);

/* pdf.loglike's in-place clean (pdf.py:309-311) over N objects: x, xe, xm (N,B)
 * are MODIFIED (host or device).  fz_fit / fz_fit_predict call it themselves. */
int  fz_clean(fz_ctx* ctx, double* x, double* xe, double* xm, int64_t N, int32_t B);

/* BruteForce._fit (bruteforce.py:127-205) with lprob_func = pdf.logprob
 * (pdf.py:326-411 -> _loglike pdf.py:27-100 / _loglike_s pdf.py:103-235).
 * Output planes are (N,M); any may be NULL.  x/xe/xm are cleaned in place. */
int  fz_fit(fz_ctx* ctx, double* x, double* xe, double* xm, int64_t N,
            const fz_like_opts* opts, double* lnlike, double* chi2, int64_t* ndim,
            double* scale, double* scale_err);

/* BruteForce._fit_predict with save_fits=False (bruteforce.py:505-631): never
 * materialises (N,M).  pdfs is (N,G); lmap/levid (N) may be NULL. */
int  fz_fit_predict(fz_ctx* ctx, double* x, double* xe, double* xm, int64_t N,
                    const fz_like_opts* opts, const fz_kde_opts* kde,
                    double* pdfs, double* lmap, double* levid);

/* mode C (pdf.py:196-223) bookkeeping since the last fz_timing_reset, out4 = {objects whose stop decision fell within rounding
 * of ltol under the reciprocal-based solve and were re-run with IEEE divisions, iterations of the slowest object, 1 = one block
 * per object, one iteration per record read / 2 = state planes in HBM / 3 = one block per object, several iterations per record
 * read (k_modec_rounds; objects it hands back to the IEEE kernel are counted in the first entry too), threads per block of 1 / 3}; and the number of passes of the loop at pdf.py:199
 * each of the first n objects of the LAST mode-C chunk took (int32, host) -- the reference does not return it, the tests compare
 * it with their restated loop. */
int  fz_modec_info(fz_ctx* ctx, int64_t* out4);
int  fz_modec_niter(fz_ctx* ctx, int64_t n, int32_t* out);

/* ---- additive ln-prior (extension) ----
 * The reference's plug-in point is lprob_func, called once per object and returning
 * (lnprior, lnlike, lnprob = lnlike + lnprior, ...) (bruteforce.py:193-199; the BPZ
 * posterior of demos/2 cell 69 is the one use).  A Python callable cannot run on the
 * device, so the prior is passed as data: a (P,M) table of ln-prior rows over the M
 * models plus the row each object reads (e.g. P magnitude bins of P(z,t|m), or P = N
 * for a dense per-object prior, or P = 1 for one prior shared by all objects).
 * lnprob[i][j] = lnlike[i][j] + table[rows[i]][j].  -inf entries (prior 0) and nan
 * propagate as they do through the reference's lnlike + lnprior. */
typedef struct fz_prior {
    const double*  table;  /* (P,M) float64 row-major, host or device          */
    int64_t        P;      /* 1 <= P < 2^31                                     */
    const int64_t* rows;   /* (N) int64 in [0,P), host or device; NULL means
                              row 0 for everybody if P == 1, row i if P == N    */
} fz_prior;
/* fz_fit with the three probability planes of bruteforce.py:197-199 (prior may be
 * NULL: lnprior = 0, lnprob = lnlike, pdf.py:404-405). */
int  fz_fit_prior(fz_ctx* ctx, double* x, double* xe, double* xm, int64_t N,
                  const fz_like_opts* opts, const fz_prior* prior, double* lnprior,
                  double* lnlike, double* lnprob, double* chi2, int64_t* ndim,
                  double* scale, double* scale_err);
/* fz_fit_predict weighting by lnprob = lnlike + lnprior (bruteforce.py:618-620). */
int  fz_fit_predict_prior(fz_ctx* ctx, double* x, double* xe, double* xm, int64_t N,
                          const fz_like_opts* opts, const fz_kde_opts* kde,
                          const fz_prior* prior, double* pdfs, double* lmap, double* levid);

/* BruteForce._predict (bruteforce.py:303-372): rows of logwt (N,M) -> PDFs.
 * is_log=0 treats the rows as linear weights y_wt and skips the softmax
 * (gauss_kde / gauss_kde_dict called directly, pdf.py:444, 529). */
int  fz_predict_logwt(fz_ctx* ctx, const double* logwt, int64_t N, int32_t is_log,
                      const fz_kde_opts* kde, double* pdfs, double* lmap, double* levid);

/* ---- Monte-Carlo k-nearest-neighbour variant (knn.py) ---- */
/* K float32 feature sets (K,M,F) as fed to KDTree (knn.py:177-186). */
int  fz_knn_upload_trees(fz_ctx* ctx, const float* feats, int32_t K, int64_t M, int32_t F);
/* exact k-NN of N float64 queries (N,F) in each of the K sets under the Minkowski
 * norm lp_norm (1, 2 or inf): the flattened (N,K*k) table of knn.py:834-837;
 * entries at or beyond distance_upper_bound are M (KDTree's "missing" index). */
int  fz_knn_query(fz_ctx* ctx, const double* q, int64_t N, int32_t k, double lp_norm,
                  double distance_upper_bound, int64_t* idx);
/* knn.py:840-872: first-appearance de-dup of each row, likelihood on the subset,
 * weights, KDE.  Outputs padded like knn.py:812-821: neighbors (N,W) with -99,
 * nnbr (N), planes (N,W) with -inf/+inf/0/1 padding; any may be NULL. */
int  fz_knn_fit_predict(fz_ctx* ctx, double* x, double* xe, double* xm, int64_t N,
                        const int64_t* idx, int64_t W, const fz_like_opts* opts,
                        const fz_kde_opts* kde, int64_t* neighbors, int64_t* nnbr,
                        double* lnlike, double* chi2, int64_t* ndim, double* scale,
                        double* scale_err, double* pdfs, double* lmap, double* levid);

/* the same with an additive ln-prior (see fz_prior): lnprior[i][s] = table[rows[i]][
 * neighbors[i][s]], lnprob = lnlike + lnprior, padded with -inf like knn.py:815-817;
 * PDFs are weighted by lnprob (knn.py:859-863).  prior may be NULL (lnprior = 0). */
int  fz_knn_fit_predict_prior(fz_ctx* ctx, double* x, double* xe, double* xm, int64_t N,
                              const int64_t* idx, int64_t W, const fz_like_opts* opts,
                              const fz_kde_opts* kde, const fz_prior* prior, int64_t* neighbors,
                              int64_t* nnbr, double* lnprior, double* lnlike, double* lnprob,
                              double* chi2, int64_t* ndim, double* scale, double* scale_err,
                              double* pdfs, double* lmap, double* levid);

/* NearestNeighbors._fit_predict (knn.py:826-874) in one call: fz_knn_query + fz_knn_fit_predict_prior over chunks of <= 2^17
 * objects with the (N, K*k) neighbour table kept on the device (it only leaves through `neighbors`, if wanted).  q (N,F): the
 * objects' query features (knn.py:830-832), drawn by the caller so that the random stream is the reference's. */
int  fz_knn_search_fit_predict_prior(fz_ctx* ctx, const double* q, double* x, double* xe, double* xm, int64_t N,
                                     int32_t k, double lp_norm, double distance_upper_bound,
                                     const fz_like_opts* opts, const fz_kde_opts* kde, const fz_prior* prior,
                                     int64_t* neighbors, int64_t* nnbr, double* lnprior, double* lnlike,
                                     double* lnprob, double* chi2, int64_t* ndim, double* scale,
                                     double* scale_err, double* pdfs, double* lmap, double* levid);

/* NearestNeighbors._predict (knn.py:488-558): PDFs from stored (N,W) ln-weights,
 * the stored neighbour table (N,W) and counts (N). */
int  fz_knn_predict_logwt(fz_ctx* ctx, const double* logwt, const int64_t* neighbors,
                          const int64_t* nnbr, int64_t N, int64_t W, const fz_kde_opts* kde,
                          double* pdfs, double* lmap, double* levid);

/* ---- consumers of the PDF stack ---- */
/* pdf.pdfs_summarize (pdf.py:899-1074).  pdfs (N,G) is renormalised IN PLACE when
 * renormalize != 0 (pdf.py:984-985).  urand (N): the rstate.rand() of each object
 * (pdf.py:1000), drawn by the caller so that the stream is the reference's.  loss
 * (G,G): 1 - kernel over (truth, guess) (pdf.py:1003-1024).  widths (N,4) or NULL:
 * the wconf_func windows around mean/median/mode/best; NULL means the default
 * (1 + point) * wconf_scale (pdf.py:1041-1043, 0.03).  stats (21,N) rows:
 * mean{value,std,conf,risk}, median{..}, mode{..}, best{..}, low95, low68, high68,
 * high95, Monte-Carlo draw -- the reference's return tuple flattened. */
int  fz_pdfs_summarize(fz_ctx* ctx, double* pdfs, int64_t N, int64_t G, const double* pgrid,
                       int32_t renormalize, const double* urand, const double* loss,
                       const double* widths, double wconf_scale, double* stats);
/* pdf.pdfs_resample (pdf.py:855-896): numpy.interp of every row of pdfs (N,G) from
 * old_grid onto new_grid (Gn points), `left` / `right` beyond the old grid's ends,
 * optional renormalisation to unit sum; out is (N,Gn). */
int  fz_pdfs_resample(fz_ctx* ctx, const double* pdfs, int64_t N, int64_t G, const double* old_grid,
                      int64_t Gn, const double* new_grid, double left, double right,
                      int32_t renormalize, double* out);
/* samplers.loglike_nz (samplers.py:23-86) for finite non-negative nz: overlap (N) =
 * pdfs @ nz + pair_step * (pdfs[:,i] - pdfs[:,j]) (pair_i < 0: no pair), lnlike =
 * sum(log(overlap)). */
int  fz_overlap_nz(fz_ctx* ctx, const double* pdfs, int64_t N, int64_t G, const double* nz,
                   int64_t pair_i, int64_t pair_j, double pair_step, double* overlap,
                   double* lnlike);

/* samplers.py:498-499, 519-520 -- the redshift-assignment step of the hierarchical / population
 * Gibbs samplers: one categorical draw per object from  pdfs[i] * nz / dot(pdfs[i], nz).  The
 * reference draws rstate.multinomial(1, ...) per object in a Python list comprehension and sums the
 * one-hot rows; here the draw is the inverse CDF of the caller's uniform u[i] in [0, 1)
 * (bin = number of grid points with running sum <= u[i] * total), so the caller keeps the random
 * stream.  bins (N, int64; -1 for a row without mass; may be NULL) and counts (G, int64: objects per
 * bin, i.e. the summed one-hot rows). */
int  fz_nz_assign(fz_ctx* ctx, const double* pdfs, int64_t N, int64_t G, const double* nz, const double* u,
                  int64_t* bins, int64_t* counts);

/* ---- inference through a trained network (networks.py:244-356, 782-936, 1200-1473) ----
 * The network is DATA here (node positions in data space and, after populate_network, the per-node model lists); training it
 * is out of scope.  Node ln-probabilities come from fz_fit with the (matched) nodes uploaded as noiseless, unmasked models
 * (networks.py:305-307, 874-876).  Every array may live in host or device memory.
 *
 * fz_net_select -- which nodes an object's (N, Nn) ln-probabilities select, in the reference's order (networks.py:885-896, 316-327):
 *   use_wt != 0: lnprob > ln(wt_thresh) + max(lnprob), strict, node index ascending (wt_thresh <= 0 or -inf: every node);
 *   use_wt == 0: ascending ln-prob, the prefix whose running probability exp(l - logsumexp) stays <= 1 - cdf_thresh.
 *   match (Nn, or NULL = identity): the node behind column c (networks.py:873 match_sel); csr_off (Nnodes + 1, or NULL): offsets of
 *   the per-node lists.  Out: nsel (N) int32, sel (N, Nn) int32 column indices (first nsel[i] valid), rawlen (N) int64 summed
 *   list length of the selected nodes (NULL if not wanted), lmap / levid (N): max and logsumexp over the selected entries
 *   (networks.py:330-333; NULL if not wanted).  Nn <= 4096. */
int  fz_net_select(fz_ctx* ctx, const double* lnprob, int64_t N, int32_t Nn, int32_t use_wt, double wt_thresh,
                   double cdf_thresh, const int32_t* match, const int64_t* csr_off, int64_t Nnodes, int32_t* nsel,
                   int32_t* sel, int64_t* rawlen, double* lmap, double* levid);
/* networks.py:912-918: idx (N, W) = the lists (csr_items, int64 model indices) of every object's selected nodes concatenated in
 * order, padded to W with the row's first entry -- the table fz_knn_fit_predict takes: it removes repeats in first-appearance
 * order, which is pandas.unique (networks.py:919), and fits the remaining models (networks.py:925-928). */
int  fz_net_table(fz_ctx* ctx, const int32_t* nsel, const int32_t* sel, int64_t N, int32_t Nn, const int32_t* match,
                  const int64_t* csr_off, const int64_t* csr_items, int64_t Nnodes, int64_t W, int64_t* idx);
/* networks.py:907-909 (nodes_only): out (N, W) 8-byte elements = plane (N, Nn) at the selected columns, `pad_bits` beyond nsel. */
int  fz_net_gather(fz_ctx* ctx, const void* plane, const int32_t* nsel, const int32_t* sel, int64_t N, int32_t Nn, int32_t W,
                   uint64_t pad_bits, void* out);
/* networks.py:1459-1466, 1473 (nodes_only): pdfs (N, G) = softmax over the selected ln-probabilities @ node_pdfs (Nnodes, G) rows of
 * the selected nodes, each row divided by its sum; lmap / levid (N) over the selected entries. */
int  fz_net_stack(fz_ctx* ctx, const double* lnprob, const int32_t* nsel, const int32_t* sel, int64_t N, int32_t Nn,
                  const int32_t* match, const double* node_pdfs, int64_t Nnodes, int64_t G, double* pdfs, double* lmap,
                  double* levid);

/* diagnostic: evaluate one of the library's device math helpers elementwise
 * (which: 0 v_rcp_f64 seed, 1 / 2 rcp with one / two Newton steps, 3 log_pos,
 * 4 exp_neg).  Used by tests to pin their accuracy against NumPy. */
int  fz_selftest_math(fz_ctx* ctx, int32_t which, const double* x, int64_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif /* FRANKENZ_HIP_H */
