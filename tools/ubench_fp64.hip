// Micro-benchmark: fp64 VALU issue behaviour on gfx950 (used to size the fused kernel's
// wave / ILP configuration).  hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP, int MODE>   // MODE 0: fma all VGPR; 1: fma with one SGPR operand; 2: v_rcp_f64; 3: mul+add mix VGPR; 4: ds_read_b64 random + fma
__global__ __launch_bounds__(256) void k(double* out, const double* in, int iters, double sc) {
    __shared__ double tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double a[ILP], b[ILP], c[ILP];
    const double u = in[0];            // uniform -> SGPR
    const double x = in[threadIdx.x];
#pragma unroll
    for (int q = 0; q < ILP; ++q) { a[q] = x + q; b[q] = in[threadIdx.x + 1 + q] * 0.999; c[q] = in[threadIdx.x + 9 + q] * 1e-3; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int q = 0; q < ILP; ++q) {
                if (MODE == 0) a[q] = fma(a[q], x, a[q]);
                else if (MODE == 1) a[q] = fma(a[q], u, a[q]);
                else if (MODE == 2) a[q] = __builtin_amdgcn_rcp(a[q]);
                else if (MODE == 3) a[q] = (a[q] * x) + x;
                else if (MODE == 5) a[q] = fma(b[q], c[q], a[q]);          // three distinct VGPR operands
                else if (MODE == 6) a[q] = a[q] * b[q];
                else if (MODE == 7) a[q] = a[q] + c[q];
                else if (MODE == 8) a[q] = fma(a[q], b[q], c[q]);
                else { const int idx = (__double2loint(a[q]) ^ (threadIdx.x * 37)) & 2047; a[q] = fma(a[q], tab[idx], x); }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < ILP; ++q) s += a[q];
    if (s == sc) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ILP, int MODE>
void run(const char* name, double* out, double* in, int wpsimd) {
    const int cus = 256, iters = 40000;
    // 256 threads = 4 waves = 1 wave per SIMD per block; wpsimd blocks per CU
    dim3 grid(cus * wpsimd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ILP, MODE>), grid, block, 0, 0, out, in, 10, -1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ILP, MODE>), grid, block, 0, 0, out, in, iters, -1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops_per_wave = (double)iters * 8 * ILP * (MODE == 3 ? 2 : 1);
    const double total = ops_per_wave * cus * wpsimd * 4;          // wave-instructions
    const double per_simd_per_s = total / (ms * 1e-3) / (cus * 4);
    printf("%-28s ILP=%d waves/SIMD=%d : %.1f ms  %.3f G wave-instr/s/SIMD -> %.2f cycles/instr @2.4GHz\n", name, ILP, wpsimd, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    double *out, *in;
    hipMalloc(&out, 256 * 2048 * 8 * 8); hipMalloc(&in, 4096);
    std::vector<double> h(512, 1.0000001);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4, 8}) {
        run<1, 0>("fma vgpr", out, in, w); run<2, 0>("fma vgpr", out, in, w); run<4, 0>("fma vgpr", out, in, w);
    }
    for (int w : {1, 4}) { run<1, 1>("fma sgpr-operand", out, in, w); run<4, 1>("fma sgpr-operand", out, in, w); }
    for (int w : {1, 4}) { run<2, 5>("fma d=b*c+d (3 regs)", out, in, w); run<4, 5>("fma d=b*c+d (3 regs)", out, in, w); run<8, 5>("fma d=b*c+d (3 regs)", out, in, w); }
    for (int w : {1, 4}) { run<4, 8>("fma d=d*b+c (3 regs)", out, in, w); run<8, 8>("fma d=d*b+c (3 regs)", out, in, w); }
    for (int w : {1, 4}) { run<4, 6>("mul a*=b", out, in, w); run<4, 7>("add a+=c", out, in, w); }
    for (int w : {1, 4}) { run<1, 2>("v_rcp_f64", out, in, w); run<4, 2>("v_rcp_f64", out, in, w); }
    for (int w : {1, 4}) { run<1, 3>("mul+add", out, in, w); run<4, 3>("mul+add", out, in, w); }
    for (int w : {1, 4}) { run<1, 4>("lds-lookup + fma", out, in, w); run<4, 4>("lds-lookup + fma", out, in, w); }
    return 0;
}
