// Microbenchmark (GPU box): issue rate of the f64 instructions the multi-set kernels are built on -
// v_fma_f64 / v_mul_f64 (VALU) and v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 (matrix cores), every CU busy.
//   hipcc -O3 --offload-arch=gfx950 tools/f64_rate.hip -o /tmp/f64_rate && /tmp/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void rate(double *out, int iters, double seed) {
    double a[8], b = seed + threadIdx.x * 1e-9, c = 1.0 - 1e-9;
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1);
    double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double acc1[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], c, b);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = a[i] * c;
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b, acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b, acc1[i], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc1[i];
    if (s == 12345.678) out[0] = s;
}

template <int MODE>
static void run(const char *name, int ops_per_iter, double flops_per_op, int waves_per_simd) {
    double *out;
    hipMalloc(&out, 8);
    const int iters = 20000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate<MODE><<<blocks, 256>>>(out, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate<MODE><<<blocks, 256>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waveops = (double)iters * ops_per_iter * waves_per_simd;           // per SIMD
    printf("%-28s %d wave(s)/SIMD: %.3f ms, %.1f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz), %.1f TFLOP/s\n", name, waves_per_simd, ms,
           ms * 1e6 / waveops, ms * 1e6 / waveops * 2.4, (double)iters * ops_per_iter * flops_per_op * blocks * 4 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 3}) {
        run<0>("v_fma_f64", 8, 128, w);
        run<1>("v_mul_f64", 8, 64, w);
        run<2>("v_mfma_f64_16x16x4_f64", 4, 2048, w);
        run<3>("v_mfma_f64_4x4x4_4b_f64", 4, 512, w);
    }
    return 0;
}
