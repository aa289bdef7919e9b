// Streaming ceilings on this GPU: pure read (sum), read+write (scale), for calibration of
// the roofline fraction of jt_collect_level / jt_distribute_level.  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void read_sum(const float4* __restrict__ in, double* __restrict__ out, size_t n_per_block) {
    const float4* p = in + (size_t)blockIdx.x * n_per_block + threadIdx.x;
    double acc = 0;
    for (size_t i = 0; i < n_per_block; i += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (double)v[u].x + (double)v[u].y + (double)v[u].z + (double)v[u].w;
    }
    if (acc == 12345.678) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void scale(const float4* __restrict__ in, float4* __restrict__ outp, size_t n_per_block) {
    const float4* p = in + (size_t)blockIdx.x * n_per_block + threadIdx.x;
    float4* q = outp + (size_t)blockIdx.x * n_per_block + threadIdx.x;
    for (size_t i = 0; i < n_per_block; i += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u].x *= 1.5f; v[u].y *= 1.5f; v[u].z *= 1.5f; v[u].w *= 1.5f; q[i + u * 256] = v[u]; }
    }
}
// 64 pieces of 4 KiB per block at a 16 KiB stride (4 neighbouring blocks interleave): the access
// shape of a leaf clique in jt_collect_level
template <int U>
__global__ __launch_bounds__(256) void read_strided(const float4* __restrict__ in, double* __restrict__ out, int iters, size_t stride4, size_t group4) {
    const float4* p = in + (size_t)(blockIdx.x / 4) * group4 + (size_t)(blockIdx.x % 4) * 256 + threadIdx.x;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < iters; i += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(size_t)(i + u) * stride4];
#pragma unroll
        for (int u = 0; u < U; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
    }
    if (a0 + a1 + a2 + a3 == 12345.678) out[0] = a0;
}
// the jt_pass loop skeleton: iteration offsets from an LDS table, 4 static load slots,
// fp64 accumulation; VARIANT 0 = as in the kernel, 1 = offsets computed (no LDS table)
template <int VARIANT>
__global__ __launch_bounds__(256) void jt_like(const float* __restrict__ in, double* __restrict__ out, int total, const int* __restrict__ gtab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* tabw = reinterpret_cast<int*>(smem);
    const int tid = threadIdx.x;
    const unsigned xF = (blockIdx.x / 4) * 262144u + (blockIdx.x % 4) * 1024u + tid * 4u;
    float4 q0 = *reinterpret_cast<const float4*>(in + xF + 0 * 4096u);
    float4 q1 = *reinterpret_cast<const float4*>(in + xF + 1 * 4096u);
    float4 q2 = *reinterpret_cast<const float4*>(in + xF + 2 * 4096u);
    float4 q3 = *reinterpret_cast<const float4*>(in + xF + 3 * 4096u);
    for (int i = tid; i < total * 8; i += 256) tabw[i] = gtab[i];
    __syncthreads();
    const int* tab = tabw;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    auto step = [&](float4& slot, int i) {
        const float4 v = slot;
        const int inext = (i + 4 < total) ? i + 4 : total - 1;
        unsigned xo;
        if (VARIANT == 0) xo = (unsigned)tab[inext * 8];
        else xo = (unsigned)inext * 4096u;
        slot = *reinterpret_cast<const float4*>(in + (xF + xo));
        if (VARIANT == 0) {
            const int4 r0 = *reinterpret_cast<const int4*>(tab + i * 8);
            if (r0.y == 12345) a0 += 1.0;
        }
        a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    };
    for (int i0 = 0; i0 < total; i0 += 4) { step(q0, i0); step(q1, i0 + 1); step(q2, i0 + 2); step(q3, i0 + 3); }
    if (a0 + a1 + a2 + a3 == 12345.678) out[0] = a0;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    float4 *a, *b; double* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 8));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n4 = bytes / 16;
    for (int blocks : {2048, 4096, 16384, 65536}) {
        size_t npb = n4 / blocks;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(read_sum<4>, dim3(blocks), dim3(256), 0, 0, a, o, npb);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("read_sum<4>  blocks %6d: %.1f GB/s\n", blocks, 10.0 * bytes / ms / 1e6);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(read_sum<1>, dim3(blocks), dim3(256), 0, 0, a, o, npb);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("read_sum<1>  blocks %6d: %.1f GB/s\n", blocks, 10.0 * bytes / ms / 1e6);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(scale<4>, dim3(blocks), dim3(256), 0, 0, a, b, npb);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("scale<4>     blocks %6d: %.1f GB/s (read+write)\n", blocks, 20.0 * bytes / ms / 1e6);
        }
    }
    {
        const int blocks = 2048 * 2;           // 1 GiB = 1024 groups of 1 MiB, 4 blocks each
        for (int rep = 0; rep < 2; ++rep) {
            float ms;
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(read_strided<4>, dim3(blocks), dim3(256), 0, 0, a, o, 64, (size_t)1024, (size_t)65536);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("read_strided<4> 4096 blocks x 64 pieces of 4 KiB, stride 16 KiB: %.1f GB/s\n", 10.0 * bytes / ms / 1e6);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(read_strided<1>, dim3(blocks), dim3(256), 0, 0, a, o, 64, (size_t)1024, (size_t)65536);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("read_strided<1> same: %.1f GB/s\n", 10.0 * bytes / ms / 1e6);
            // 16 pieces per block at stride 64 KiB (16 blocks interleave)
        }
    }
    {
        std::vector<int> tab(64 * 8, 0);
        for (int i = 0; i < 64; ++i) tab[i * 8] = i * 4096;
        int* dtab; CK(hipMalloc(&dtab, tab.size() * 4)); CK(hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        for (int blocks : {2048, 4096}) for (int rep = 0; rep < 2; ++rep) {
            float ms;
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(jt_like<0>, dim3(blocks), dim3(256), 6144, 0, (const float*)a, o, 64, dtab);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("jt_like<0> (LDS table) blocks %d: %.1f GB/s\n", blocks, 10.0 * blocks * 262144.0 / ms / 1e6);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(jt_like<1>, dim3(blocks), dim3(256), 6144, 0, (const float*)a, o, 64, dtab);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("jt_like<1> (computed offsets) blocks %d: %.1f GB/s\n", blocks, 10.0 * blocks * 262144.0 / ms / 1e6);
        }
    }
    return 0;
}
