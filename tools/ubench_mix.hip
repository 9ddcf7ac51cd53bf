// Micro-benchmark: what the fp64 matrix pipe and the fp32 vector pipe give on gfx950, alone and
// side by side in one wave (sizes the MFMA form of the chi2 contraction, fz_mfma.h).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mix.hip -o tools/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: v_fma_f32   1: v_pk_fma_f32   2: v_exp_f32   3: v_sqrt_f32   4: v_cvt_f32_f64 (+ cvt back)
//      5: v_fma_f64   6: v_rsq_f32      7: v_max_f32   8: v_cvt_f32_f64 only (fresh f64 each time)
template <int ILP, int MODE>
__global__ __launch_bounds__(256) void k_valu(float* out, const float* in, int iters, float sc) {
    float a[ILP]; f2 p[ILP]; double d[ILP];
    const float x = in[threadIdx.x], y = in[threadIdx.x + 1];
#pragma unroll
    for (int q = 0; q < ILP; ++q) { a[q] = x + q; p[q] = f2{x + q, y + q}; d[q] = (double)x + q; }
    const f2 px = {x, y};
    const double dx = x, dy = y;
    const float sx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int q = 0; q < ILP; ++q) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(x), "v"(y));
                else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(px));
                else if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[q]));
                else if (MODE == 3) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[q]));
                else if (MODE == 4) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[q]) : "v"(d[q])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[q]) : "v"(a[q])); }
                else if (MODE == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[q]) : "v"(dx), "v"(dy));
                else if (MODE == 6) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[q]));
                else if (MODE == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[q]) : "v"(y));
                else if (MODE == 8) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[q]) : "v"(d[q])); }
                else if (MODE == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[q]) : "v"(px));
                else if (MODE == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[q]) : "v"(px));
                else if (MODE == 11) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[q]), "v"(y) : "vcc");
                else if (MODE == 12) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[q]) : "v"(dx));
                else if (MODE == 13) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[q]) : "v"(dx));
                else if (MODE == 14) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[q]) : "v"(y));
                else if (MODE == 15) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[q]) : "v"(y));
                else if (MODE == 16) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[q]) : "v"(y));
                else if (MODE == 17) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(x), "v"(y));
                else if (MODE == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(y));
                else if (MODE == 19) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[q]) : "v"(y));
                else if (MODE == 20) asm volatile("v_max_f32_e64 %0, %0, %1" : "+v"(a[q]) : "v"(y));
                else if (MODE == 21) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[q]) : "s"(sx), "v"(y));
                else if (MODE == 22) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[q]) : "v"(dx));
                else if (MODE == 23) asm volatile("v_mov_b32 %0, %1" : "=v"(a[q]) : "v"(y));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int q = 0; q < ILP; ++q) s += a[q] + p[q].x + p[q].y + (float)d[q];
    if (s == sc) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// NACC independent fp64 MFMA accumulators, NV fp32 VALU instructions of kind VK per MFMA
//   VK 0: v_fma_f32   1: v_exp_f32   2: v_fma_f64    3: v_pk_fma_f32
template <int NACC, int NV, int VK>
__global__ __launch_bounds__(256) void k_mfma(double* out, const double* in, int iters, double sc) {
    d4 acc[NACC];
    const double av = in[threadIdx.x], bv = in[threadIdx.x + 7];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = d4{0, 0, 0, 0};
    float v[8]; double w[8]; f2 p[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { v[q] = (float)av + q; w[q] = av + q; p[q] = f2{(float)av, (float)bv}; }
    const float fx = (float)av, fy = (float)bv;
    const f2 px = {fx, fy};
    // inline asm pins the instruction stream: one MFMA, then NV vector instructions, in program order
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(av), "v"(bv));
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                if (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(fx), "v"(fy));
                else if (VK == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u & 7]));
                else if (VK == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(w[u & 7]) : "v"(av), "v"(bv));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u & 7]) : "v"(px), "v"(px));
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q] + w[q] + p[q].x + p[q].y;
    if (s == sc) out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double timeit(void (*launch)(int, int), int wps, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(wps, 10); hipDeviceSynchronize();
    hipEventRecord(e0); launch(wps, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
static float* g_out; static float* g_in; static double* g_dout; static double* g_din;

template <int ILP, int MODE>
void launch_valu(int wps, int iters) { hipLaunchKernelGGL((k_valu<ILP, MODE>), dim3(256 * wps), dim3(256), 0, 0, g_out, g_in, iters, -1.0f); }
template <int NACC, int NV, int VK>
void launch_mfma(int wps, int iters) { hipLaunchKernelGGL((k_mfma<NACC, NV, VK>), dim3(256 * wps), dim3(256), 0, 0, g_dout, g_din, iters, -1.0); }

template <int ILP, int MODE>
void run_valu(const char* name, int wps, int per = 1) {
    const int iters = 20000;
    const double ms = timeit(launch_valu<ILP, MODE>, wps, iters);
    const double instr = (double)iters * 8 * ILP * per * wps;     // wave-instructions per SIMD
    printf("%-22s ILP=%d waves/SIMD=%d : %7.2f ms  %.2f cycles/instr/SIMD @2.4GHz\n", name, ILP, wps, ms, ms * 1e-3 * 2.4e9 / instr);
}
template <int NACC, int NV, int VK>
void run_mfma(const char* name, int wps) {
    const int iters = 20000;
    const double ms = timeit(launch_mfma<NACC, NV, VK>, wps, iters);
    const double n = (double)iters * NACC * wps;                  // MFMAs per SIMD
    printf("mfma_f64_16x16x4 acc=%d + %2d x %-12s waves/SIMD=%d : %7.2f ms  %.1f cycles per MFMA(+VALU group)/SIMD @2.4GHz = %.1f TFLOP/s fp64 MFMA\n",
           NACC, NV, name, wps, ms, ms * 1e-3 * 2.4e9 / n, n * 1024 * 2048.0 / (ms * 1e-3) / 1e12);
}

int main() {
    hipMalloc(&g_out, 256 * 2048 * 8 * 4); hipMalloc(&g_in, 4096); hipMalloc(&g_dout, 256 * 2048 * 8 * 8); hipMalloc(&g_din, 8192);
    std::vector<float> h(1024, 1.0000001f); std::vector<double> hd(1024, 1.0000001);
    hipMemcpy(g_in, h.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(g_din, hd.data(), 8192, hipMemcpyHostToDevice);
    for (int w : {2, 4, 8}) {
        run_valu<4, 14>("v_mul_f32", w); run_valu<4, 15>("v_add_f32", w); run_valu<4, 16>("v_min_f32", w); run_valu<4, 17>("v_med3_f32", w);
        run_valu<4, 18>("v_cndmask_b32", w); run_valu<4, 19>("v_mul_f32_e64", w); run_valu<4, 20>("v_max_f32_e64", w); run_valu<4, 21>("v_fma_f32 sgpr", w);
        run_valu<4, 22>("v_min_f64", w); run_valu<4, 23>("v_mov_b32", w);
    }
    for (int w : {1, 2, 4, 8}) {
        run_valu<4, 0>("v_fma_f32", w); run_valu<4, 1>("v_pk_fma_f32", w); run_valu<4, 5>("v_fma_f64", w);
        run_valu<4, 2>("v_exp_f32", w); run_valu<4, 3>("v_sqrt_f32", w); run_valu<4, 6>("v_rsq_f32", w);
        run_valu<4, 4>("cvt f64->f32->f64", w, 2); run_valu<4, 8>("v_cvt_f32_f64", w); run_valu<4, 7>("v_max_f32", w);
        run_valu<4, 9>("v_pk_mul_f32", w); run_valu<4, 10>("v_pk_add_f32", w); run_valu<4, 11>("v_cmp_gt_f32", w); run_valu<4, 12>("v_mul_f64", w); run_valu<4, 13>("v_add_f64", w);
    }
    for (int w : {1, 2, 4}) {
        run_mfma<1, 0, 0>("-", w); run_mfma<2, 0, 0>("-", w); run_mfma<4, 0, 0>("-", w);
        run_mfma<4, 4, 0>("v_fma_f32", w); run_mfma<4, 8, 0>("v_fma_f32", w); run_mfma<4, 12, 0>("v_fma_f32", w); run_mfma<4, 16, 0>("v_fma_f32", w); run_mfma<4, 24, 0>("v_fma_f32", w);
        run_mfma<4, 2, 1>("v_exp_f32", w); run_mfma<4, 4, 1>("v_exp_f32", w); run_mfma<4, 8, 1>("v_exp_f32", w);
        run_mfma<4, 4, 2>("v_fma_f64", w); run_mfma<4, 8, 2>("v_fma_f64", w); run_mfma<4, 12, 2>("v_fma_f64", w);
        run_mfma<4, 8, 3>("v_pk_fma_f32", w); run_mfma<4, 16, 3>("v_pk_fma_f32", w);
    }
    return 0;
}
