// development only (tools/devbuild.sh): every band count except 5 answers "not compiled", so that the experimental
// library is small (it travels to the GPU box with every gpurun call).  Never part of a release build.
#include "../frankenz_amd/csrc/fz_ctx.h"
#define FZ_STUB_BT(N)                                                                                     \
    int fz_planes_bt##N(fz_ctx*, int, int, int, int64_t, double*, double*, int64_t*, double*, double*) { return fail(-1, "dev build: 5 bands only"); } \
    int fz_fitpredict_bt##N(fz_ctx*, int, int, int, int64_t, const fz_kde_opts*, double*, double*, double*) { return fail(-1, "dev build: 5 bands only"); } \
    int fz_modec_bt##N(fz_ctx*, int, int64_t, const fz_like_opts*, const int64_t*, const int64_t*, int) { return fail(-1, "dev build: 5 bands only"); } \
    int fz_knnsubset_bt##N(fz_ctx*, int, int, int, int64_t, const int64_t*, int, const fz_kde_opts*, const fz::KnnOut*, int*) { return fail(-1, "dev build: 5 bands only"); } \
    int fz_knnquery_bt##N(fz_ctx*, const double*, int64_t, int, double, int64_t*, int) { return fail(-1, "dev build: 5 bands only"); }
FZ_STUB_BT(4)
FZ_STUB_BT(6)
FZ_STUB_BT(7)
FZ_STUB_BT(8)
FZ_STUB_BT(12)
FZ_STUB_BT(16)
FZ_STUB_BT(24)
FZ_STUB_BT(32)
