// Read-pattern microbenchmark for the stored-plane predict path (k_plane_fused): how fast can 1e5 rows of 1e4 doubles be
// streamed by (a) one wave per row, (b) one block per row, (c) a flat grid-stride sweep -- with and without the nt hint,
// at two prefetch depths.  hipcc --offload-arch=gfx950 -O3 tools/ubench_read.hip -o /tmp/ubench_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ d2 ld(const d2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// MODE 0: wave per row; MODE 1: block per row; rows of M doubles (M even), U loads per trip, one trip ahead
template <int NW, int U, bool NT, int MODE>
__global__ __launch_bounds__(NW * 64) void k_rows(const double* __restrict__ plane, int64_t N, int M, double* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nstreams = MODE == 0 ? (int64_t)gridDim.x * NW : gridDim.x;
    const int64_t s0 = MODE == 0 ? (int64_t)blockIdx.x * NW + wave : blockIdx.x;
    const int T = MODE == 0 ? 64 : NW * 64, t = MODE == 0 ? lane : tid;
    const int STEP = T * 2 * U;
    double acc = 0.0;
    for (int64_t i = s0; i < N; i += nstreams) {
        const d2* r = reinterpret_cast<const d2*>(plane + i * M);
        d2 l[U], ln[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int j = (u * T + t); l[u] = (2 * j < M) ? ld<NT>(r + j) : d2{0, 0}; }
        for (int jb = 0; jb < M; jb += STEP) {
            const int jn = jb + STEP;
#pragma unroll
            for (int u = 0; u < U; ++u) { const int j = jn / 2 + (u * T + t); ln[u] = (2 * j < M) ? ld<NT>(r + j) : d2{0, 0}; }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc += l[u].x; acc = fmax(acc, l[u].y); l[u] = ln[u]; }
        }
    }
    if (acc == 1.2345) out[0] = acc;
}
template <int U, bool NT>
__global__ __launch_bounds__(512) void k_flat(const double* __restrict__ plane, int64_t n2, double* __restrict__ out) {
    const d2* p = reinterpret_cast<const d2*>(plane);
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 512 * U;
    for (int64_t b = (int64_t)blockIdx.x * 512 * U; b < n2; b += stride) {
        d2 l[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t j = b + u * 512 + threadIdx.x; l[u] = j < n2 ? ld<NT>(p + j) : d2{0, 0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += l[u].x; acc = fmax(acc, l[u].y); }
    }
    if (acc == 1.2345) out[0] = acc;
}

template <class F> void timeit(const char* name, double bytes, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int k = 0; k < 5; ++k) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / ms * 1e-9); fflush(stdout);
}
int main() {
    const int64_t N = 100000; const int M = 10000;
    double *plane, *out; CK(hipMalloc(&plane, N * M * 8)); CK(hipMalloc(&out, 8));
    CK(hipMemset(plane, 0, N * M * 8));
    const double bytes = (double)N * M * 8;
    const int CU = 256;
#define ROWS(NW, U, NT, MODE, BPC) timeit("rows NW=" #NW " U=" #U " nt=" #NT " mode=" #MODE " bpc=" #BPC, bytes, [&] { \
        hipLaunchKernelGGL((k_rows<NW, U, NT, MODE>), dim3(CU * BPC), dim3(NW * 64), 0, 0, plane, N, M, out); })
    ROWS(8, 4, true, 0, 2);
    ROWS(8, 4, false, 0, 2);
    ROWS(8, 8, true, 0, 2);
    ROWS(8, 4, true, 0, 4);
    ROWS(8, 2, true, 0, 4);
    ROWS(8, 4, true, 0, 1);
    ROWS(4, 4, true, 0, 2);
    ROWS(8, 4, true, 1, 2);
    ROWS(8, 4, false, 1, 2);
    ROWS(8, 2, true, 1, 4);
    ROWS(8, 1, true, 1, 4);
    ROWS(4, 4, true, 1, 4);
    ROWS(4, 2, true, 1, 8);
    timeit("flat U=4 nt", bytes, [&] { hipLaunchKernelGGL((k_flat<4, true>), dim3(CU * 4), dim3(512), 0, 0, plane, N * M / 2, out); });
    timeit("flat U=4", bytes, [&] { hipLaunchKernelGGL((k_flat<4, false>), dim3(CU * 4), dim3(512), 0, 0, plane, N * M / 2, out); });
    timeit("flat U=8 nt", bytes, [&] { hipLaunchKernelGGL((k_flat<8, true>), dim3(CU * 2), dim3(512), 0, 0, plane, N * M / 2, out); });
    timeit("flat U=2 nt x8", bytes, [&] { hipLaunchKernelGGL((k_flat<2, true>), dim3(CU * 8), dim3(512), 0, 0, plane, N * M / 2, out); });
    return 0;
}
