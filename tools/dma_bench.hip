// Streaming rate of the jt_pass element loop in isolation: LDS-DMA ring of 4 x 1 KiB per wave,
// hand-counted vmcnt, ds_read_b128, fp64 sums - no messages, no epilogue.  Compare with
// tools/membench.hip (register loads).   hipcc -O3 --offload-arch=gfx950 tools/dma_bench.hip -o dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void dma16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// each workgroup streams `iters` rows of 4 KiB: consecutive rows (stride 1), or every `stride`-th row of its
// 4 MiB table (the loop bits of a clique are not always its lowest free bits)
template <int STORE>
__global__ __launch_bounds__(256) void dma_stream(const float *__restrict__ in, float *__restrict__ outp, double *__restrict__ out, int iters, int stride, size_t wshift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // table = 1024 rows; workgroups per table = 1024 / iters; workgroup j of a table starts at row j (strided) or j * iters
    const int per_table = 1024 / iters;
    const size_t table = blockIdx.x / per_table, j = blockIdx.x % per_table;
    const size_t row0 = table * 1024 + (stride > 1 ? j : j * iters);
    const size_t rs = (size_t)(stride > 1 ? per_table : 1) * 1024;          // floats between this workgroup's rows
    const float *base = in + row0 * 1024 + tid * 4;
    float *obase = outp + wshift + row0 * 1024 + tid * 4;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + (uint32_t)wave * 4096;
    const char *ring = smem + wave * 4096 + lane * 16;
    for (int u = 0; u < 4; ++u) dma16(base + u * rs, __builtin_amdgcn_readfirstlane(ring_lds + u * 1024));
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i0 = 0; i0 < iters; i0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (STORE) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            const float4 v = *reinterpret_cast<const float4 *>(ring + u * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int inext = (i0 + u + 4 < iters) ? i0 + u + 4 : iters - 1;
            dma16(base + (size_t)inext * rs, __builtin_amdgcn_readfirstlane(ring_lds + u * 1024));
            a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
            if (STORE) {
                typedef float ext_t __attribute__((ext_vector_type(4)));
                ext_t ov = {v.x * 2, v.y * 2, v.z * 2, v.w * 2};
                __builtin_nontemporal_store(ov, reinterpret_cast<ext_t *>(obase + (size_t)(i0 + u) * rs));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a0 + a1 + a2 + a3 == 12345.678) out[0] = a0;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    float *a, *b; double *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes + (16 << 20))); CK(hipMalloc(&o, 8));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t wshift : {(size_t)0, (size_t)64, (size_t)1024, (size_t)16384, (size_t)(1 << 18), (size_t)(3 << 19)}) {     // write stream shifted by this many floats
        float ms;
        const int iters = 64, blocks = (int)(bytes / ((size_t)iters * 4096));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(dma_stream<1>, dim3(blocks), dim3(256), 37120, 0, (const float *)a, b, o, iters, 1, wshift);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("read+write, write stream shifted by %8zu bytes: %.0f GB/s\n", wshift * 4, 10.0 * 2 * bytes / ms / 1e6);
        }
    }
    for (int lds : {37120}) for (int iters : {64}) for (int stride : {1, 2}) {
        const int blocks = (int)(bytes / ((size_t)iters * 4096));
        if (lds > 65536) continue;
        for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
            float ms;
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) {
                if (mode == 0) hipLaunchKernelGGL(dma_stream<0>, dim3(blocks), dim3(256), lds, 0, (const float *)a, b, o, iters, stride, (size_t)0);
                else hipLaunchKernelGGL(dma_stream<1>, dim3(blocks), dim3(256), lds, 0, (const float *)a, b, o, iters, stride, (size_t)0);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%s  LDS %5d B  %2d iterations/workgroup  %6d workgroups  rows %s: %.0f GB/s\n", mode ? "read+write" : "read      ", lds, iters, blocks, stride > 1 ? "strided   " : "contiguous",
                            10.0 * (mode ? 2 : 1) * bytes / ms / 1e6);
        }
    }
    return 0;
}
