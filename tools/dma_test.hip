// Check of the LDS-DMA recipe used by jt_pass: global_load_lds_dwordx4 into a wave-private ring,
// counted s_waitcnt vmcnt, ds_read_b128 back.  hipcc --offload-arch=gfx950 tools/dma_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void jt_dma16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + 256;   // non-zero offset on purpose
    const float* src = in + (size_t)blockIdx.x * iters * 1024 + tid * 4;
    float acc = 0;
    for (int u = 0; u < 4; ++u) jt_dma16(src + u * 1024, __builtin_amdgcn_readfirstlane(base + wave * 4096 + u * 1024));
    for (int i = 0; i < iters; ++i) {
        wait_vm<3>();
        const float4 v = *reinterpret_cast<const float4 *>(smem + 256 + wave * 4096 + (i & 3) * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int nx = i + 4 < iters ? i + 4 : iters - 1;
        jt_dma16(src + nx * 1024, __builtin_amdgcn_readfirstlane(base + wave * 4096 + (i & 3) * 1024));
        acc += v.x + 2 * v.y + 3 * v.z + 4 * v.w + i;
    }
    out[blockIdx.x * 256 + tid] = acc;
}
int main() {
    const int blocks = 512, iters = 64;
    std::vector<float> h((size_t)blocks * iters * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 97) * 0.25f;
    float *d, *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, blocks * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 256 + 16384, 0, d, o, iters);
    std::vector<float> r(blocks * 256);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < blocks; ++b) for (int t = 0; t < 256; ++t) {
        float acc = 0;
        for (int i = 0; i < iters; ++i) { const float* v = &h[(size_t)b * iters * 1024 + i * 1024 + t * 4]; acc += v[0] + 2 * v[1] + 3 * v[2] + 4 * v[3] + i; }
        if (acc != r[b * 256 + t]) ++bad;
    }
    printf("lds-dma ring check: %d mismatches of %d (%s)\n", bad, blocks * 256, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
