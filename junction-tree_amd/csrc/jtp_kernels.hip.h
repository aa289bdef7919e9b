// CDNA4 (gfx950) kernels of the junction-tree message-passing hot path.
//
// One kernel body, `jt_pass`, covers the reference's per-clique work in both traversal
// directions (junctiontree/computation.py):
//   collect    (:47-96)   up[S_p]     = sum_{C \ S_p} psi * prod_k up_k
//   distribute (:140-224) down_k[S_k] = sum_{C \ S_k} psi * down_p * prod_{j != k} up_j
//                         belief[C]   = psi * down_p * prod_k up_k
// i.e. the einsum call sites K1..K5 and K7 of SURVEY.md section 2.1 fused into ONE pass
// over the clique table per phase.  The divide-out of `remove_message` (:99-136) does not
// exist here: the all-but-one products are formed directly.
//
// Bound: HBM.  Per clique element the kernel reads sizeof(T) bytes (collect) or reads and
// writes sizeof(T) (distribute); everything else (messages) lives in LDS.  See DESIGN.md.
//
// Work decomposition (jtp_internal.h): a workgroup of 256 threads owns one chunk (fixed F
// bits) of one clique.  Thread t loads VEC consecutive elements (16 bytes) at
//   x = xF + xA(a) + xR(r) + t*VEC
// so a wave reads 1 KiB contiguous per instruction.  Incoming messages are staged once per
// workgroup into LDS as the sub-box this chunk can touch (partial copies summed on the
// way); per element the message entry is one ds_read at  thread_offset + uniform_offset.
// Outgoing sums are kept in VEC registers per message over the R loop, then reduced
// in-thread (e bits), by wave shuffles (lane bits) and through the LDS sub-box (wave bits
// and A loop), and written once per workgroup as one partial copy.  No atomics: results are
// bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>

#include "jtp_internal.h"

template <typename T> struct JtVec;
template <> struct JtVec<float> { using type = float4; };
template <> struct JtVec<double> { using type = double2; };

__device__ __forceinline__ double jt_shfl_xor(double v, int laneMask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, laneMask, 64);
    hi = __shfl_xor(hi, laneMask, 64);
    return __hiloint2double(hi, lo);
}

template <typename T, int NIN, int NOUT, int MODE>
__device__ __forceinline__ void jt_pass(const JtTask *__restrict__ tasks, const uint2 *__restrict__ blk,
                                        const T *__restrict__ psi_arena, T *__restrict__ bel_arena,
                                        double *__restrict__ msg_arena) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int EB = (VEC == 4) ? 2 : 1;
    constexpr int NMSG = NIN + NOUT;
    using VT = typename JtVec<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const uint2 bt = blk[blockIdx.x];
    const JtTask &tk = tasks[bt.x];
    const uint32_t chunk = bt.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- chunk decode: element base, message bases, partial-copy numbers ----------------
    uint32_t xF = 0;
    int gbase[NMSG > 0 ? NMSG : 1];
    int pnum[NOUT > 0 ? NOUT : 1];
#pragma unroll
    for (int k = 0; k < NMSG; ++k) gbase[k] = 0;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) pnum[k] = 0;
    for (int j = 0; j < tk.nF; ++j) {
        if ((chunk >> j) & 1u) {
            xF += tk.f_x[j];
#pragma unroll
            for (int k = 0; k < NIN; ++k) gbase[k] += tk.msg[k].f_w[j];
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                gbase[NIN + k] += tk.msg[JT_MAX_IN + k].f_w[j];
                pnum[k] += tk.msg[JT_MAX_IN + k].f_p[j];
            }
        }
    }

    // ---- stage incoming sub-boxes (summing partial copies), zero outgoing sub-boxes -----
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const JtMsg &m = tk.msg[k];
        double *sub = reinterpret_cast<double *>(smem + m.lds_off);
        const double *src = msg_arena + m.off + gbase[k];
        const int n = 1 << m.nfree;
        for (int s = tid; s < n; s += JT_THREADS) {
            int idx = 0;
            for (int b = 0; b < m.nfree; ++b) idx += ((s >> b) & 1) << m.free_pos[b];
            double sum = 0.0;
            for (int p = 0; p < m.npart; ++p) sum += src[(int64_t)p * m.pstride + idx];
            sub[s] = sum;
        }
    }
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const JtMsg &m = tk.msg[JT_MAX_IN + k];
        double *sub = reinterpret_cast<double *>(smem + m.lds_off);
        const int n = 1 << m.nfree;
        for (int s = tid; s < n; s += JT_THREADS) sub[s] = 0.0;
    }
    __syncthreads();

    // ---- per-thread constants ---------------------------------------------------------------
    int thr[NMSG > 0 ? NMSG : 1];
    const double *in_sub[NIN > 0 ? NIN : 1];
    double *out_sub[NOUT > 0 ? NOUT : 1];
#pragma unroll
    for (int k = 0; k < NMSG; ++k) {
        const JtMsg &m = tk.msg[k < NIN ? k : JT_MAX_IN + (k - NIN)];
        int t = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) t += ((lane >> b) & 1) * m.t_w[b];
#pragma unroll
        for (int b = 0; b < 2; ++b) t += ((wave >> b) & 1) * m.t_w[6 + b];
        thr[k] = t;
        if (k < NIN) in_sub[k < NIN ? k : 0] = reinterpret_cast<const double *>(smem + m.lds_off);
        else out_sub[k >= NIN ? k - NIN : 0] = reinterpret_cast<double *>(smem + m.lds_off);
    }

    const bool virt = tk.psi_off < 0;
    const bool wbel = (MODE == 1) && tk.bel_off >= 0;
    const T *psi = psi_arena + (virt ? 0 : tk.psi_off);
    T *bel = bel_arena + (wbel ? tk.bel_off : 0);
    const uint32_t real_limit = tk.real_bits >= 32 ? 0xffffffffu : (1u << tk.real_bits);
    const int nA = 1 << tk.nA, nR = 1 << tk.nR;
    constexpr int NPAR = NIN - NOUT * (MODE == 1);   // distribute: leading inputs that are not children

    uint32_t xa = 0;
    int oa[NMSG > 0 ? NMSG : 1];
#pragma unroll
    for (int k = 0; k < NMSG; ++k) oa[k] = 0;

    for (int a = 0; a < nA; ++a) {
        double acc[NOUT > 0 ? NOUT : 1][VEC];
#pragma unroll
        for (int j = 0; j < NOUT; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[j][e] = 0.0;
        uint32_t xr = 0;
        int orr[NIN > 0 ? NIN : 1];
#pragma unroll
        for (int k = 0; k < NIN; ++k) orr[k] = 0;

        for (int r = 0; r < nR; ++r) {
            const uint32_t x = xF + xa + xr + (uint32_t)tid * VEC;
            double p[VEC];
            if (!virt) {
                const VT v = *reinterpret_cast<const VT *>(psi + x);
                p[0] = (double)v.x;
                p[1] = (double)v.y;
                if constexpr (VEC == 4) {
                    p[2] = (double)v.z;
                    p[3] = (double)v.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) p[e] = (x + e) < real_limit ? 1.0 : 0.0;
            }
            double in[NIN > 0 ? NIN : 1][VEC];
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
                const JtMsg &m = tk.msg[k];
                const int base = oa[k] + orr[k] + thr[k];
                if (m.e_dep) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const int eo = ((e & 1) ? m.e_w[0] : 0) + ((e & 2) ? m.e_w[1] : 0);
                        in[k][e] = in_sub[k][base + eo];
                    }
                } else {
                    const double t = in_sub[k][base];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) in[k][e] = t;
                }
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    double q = p[e];
#pragma unroll
                    for (int k = 0; k < NIN; ++k) q *= in[k][e];
                    if constexpr (NOUT > 0) acc[0][e] += q;
                }
            } else {
                double b[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    double pre = p[e];
#pragma unroll
                    for (int k = 0; k < NPAR; ++k) pre *= in[k][e];
                    // all-but-one products over the children: prefix * suffix
                    double suf[NOUT + 1];
                    suf[NOUT] = 1.0;
#pragma unroll
                    for (int j = NOUT - 1; j >= 0; --j) suf[j] = suf[j + 1] * in[NPAR + j][e];
                    double pref = pre;
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        acc[j][e] += pref * suf[j + 1];
                        pref *= in[NPAR + j][e];
                    }
                    b[e] = pref;
                }
                if (wbel) {
                    VT o;
                    o.x = (T)b[0];
                    o.y = (T)b[1];
                    if constexpr (VEC == 4) {
                        o.z = (T)b[2];
                        o.w = (T)b[3];
                    }
                    *reinterpret_cast<VT *>(bel + x) = o;
                }
            }
            if (r + 1 < nR) {
                const int t = __builtin_ctz((unsigned)(r + 1));
                xr += (uint32_t)tk.dR[t][0];
#pragma unroll
                for (int k = 0; k < NIN; ++k) orr[k] += tk.dR[t][1 + k];
            }
        }

        // ---- epilogue: fold this thread's sums into the outgoing sub-boxes ------------------
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            const JtMsg &m = tk.msg[JT_MAX_IN + j];
            if constexpr (VEC == 4) {
                if (m.red_e & 1) {
                    acc[j][0] += acc[j][1];
                    acc[j][2] += acc[j][3];
                }
                if (m.red_e & 2) {
                    acc[j][0] += acc[j][2];
                    acc[j][1] += acc[j][3];
                }
            } else {
                if (m.red_e & 1) acc[j][0] += acc[j][1];
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                if ((m.red_lane >> b) & 1) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if ((e & m.red_e) == 0) acc[j][e] += jt_shfl_xor(acc[j][e], 1 << b);
                }
            }
            const bool rep = (lane & m.red_lane) == 0;
            const int slot = oa[NIN + j] + thr[NIN + j];
            const int nph = 1 << __builtin_popcount((unsigned)m.red_wave);
            // waves that share slots (wave bits not in the message) take turns, in wave order
            int myph = 0;
            if (m.red_wave == 1) myph = wave & 1;
            else if (m.red_wave == 2) myph = wave >> 1;
            else if (m.red_wave == 3) myph = wave;
            for (int ph = 0; ph < nph; ++ph) {
                if (rep && myph == ph) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        if ((e & m.red_e) == 0) {
                            const int eo = ((e & 1) ? m.e_w[0] : 0) + ((e & 2) ? m.e_w[1] : 0);
                            out_sub[j][slot + eo] += acc[j][e];
                        }
                    }
                }
                if (nph > 1) __syncthreads();
            }
        }
        if (a + 1 < nA) {
            const int t = __builtin_ctz((unsigned)(a + 1));
            xa += (uint32_t)tk.dA[t][0];
#pragma unroll
            for (int k = 0; k < NIN; ++k) oa[k] += tk.dA[t][1 + k];
#pragma unroll
            for (int j = 0; j < NOUT; ++j) oa[NIN + j] += tk.dA[t][1 + JT_MAX_IN + j];
        }
    }

    // ---- flush outgoing sub-boxes as this chunk's partial copy ----------------------------------
    if constexpr (NOUT > 0) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            const JtMsg &m = tk.msg[JT_MAX_IN + j];
            double *dst = msg_arena + m.off + (int64_t)pnum[j] * m.pstride + gbase[NIN + j];
            const int n = 1 << m.nfree;
            for (int s = tid; s < n; s += JT_THREADS) {
                int idx = 0;
                for (int b = 0; b < m.nfree; ++b) idx += ((s >> b) & 1) << m.free_pos[b];
                dst[idx] = out_sub[j][s];
            }
        }
    }
}

// Named entry points (these names appear in rocprofv3 traces).
template <typename T, int NCH>
__global__ __launch_bounds__(JT_THREADS) void jt_collect(const JtTask *__restrict__ tasks, const uint2 *__restrict__ blk,
                                                         const T *__restrict__ psi, T *__restrict__ bel,
                                                         double *__restrict__ msg) {
    jt_pass<T, NCH, 1, 0>(tasks, blk, psi, bel, msg);
}

template <typename T, int HASP, int NCH>
__global__ __launch_bounds__(JT_THREADS) void jt_distribute(const JtTask *__restrict__ tasks, const uint2 *__restrict__ blk,
                                                            const T *__restrict__ psi, T *__restrict__ bel,
                                                            double *__restrict__ msg) {
    jt_pass<T, HASP + NCH, NCH, 1>(tasks, blk, psi, bel, msg);
}

// ------------------------------------------------------------------------------------------
// Layout conversion and synthetic fill (off the hot path).

__device__ __forceinline__ uint64_t jt_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// device index -> host index; returns false for padding entries
__device__ __forceinline__ bool jt_dev_to_host(const JtPackDesc &d, uint32_t x, int64_t &hidx) {
    bool valid = true;
    int64_t h = 0;
    int used = 0;
    for (int i = 0; i < d.nvars; ++i) {
        const int nb = d.nb[i];
        const int digit = (int)((x >> d.pos[i]) & ((1u << nb) - 1u));
        valid = valid && (digit < d.card[i]);
        h += (int64_t)digit * d.hstride[i];
        used += nb;
    }
    if (used < 32 && (x >> used) != 0) valid = false;
    hidx = h;
    return valid;
}

// MODE 0: arena[x] = stage[host index] (pack);  MODE 1: synthetic fill
template <typename T, typename S, int MODE>
__global__ __launch_bounds__(256) void jt_pack(JtPackDesc d, const S *__restrict__ stage, T *__restrict__ arena,
                                               uint64_t key, double scale) {
    const int64_t n = (int64_t)1 << d.nbits;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
        int64_t h;
        const bool valid = jt_dev_to_host(d, (uint32_t)x, h);
        double v = 0.0;
        if (valid) {
            if constexpr (MODE == 0) v = (double)stage[h];
            else {
                const uint64_t bits = jt_splitmix64(key + (uint64_t)h);
                v = (0.5 + (double)(bits >> 11) * (1.0 / 9007199254740992.0)) * scale;
            }
        }
        arena[d.dev_off + x] = (T)v;
    }
}

// host index -> device index
__device__ __forceinline__ uint32_t jt_host_to_dev(const JtPackDesc &d, int64_t h) {
    uint32_t x = 0;
    for (int i = d.nvars - 1; i >= 0; --i) {
        const int c = d.card[i];
        const int digit = (int)(h % c);
        h /= c;
        x |= (uint32_t)digit << d.pos[i];
    }
    return x;
}

template <typename T, typename S>
__global__ __launch_bounds__(256) void jt_unpack(JtPackDesc d, const T *__restrict__ arena, S *__restrict__ stage) {
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < d.host_elems;
         h += (int64_t)gridDim.x * blockDim.x)
        stage[h] = (S)arena[d.dev_off + jt_host_to_dev(d, h)];
}

// message(s) -> host order: out[h] = (sum_p up[p]) * (dn ? sum_p dn[p] : 1)
template <typename S>
__global__ __launch_bounds__(256) void jt_msg_unpack(JtPackDesc d, const double *__restrict__ up, int up_npart,
                                                     const double *__restrict__ dn, int dn_npart, int64_t pstride,
                                                     S *__restrict__ stage) {
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < d.host_elems;
         h += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t x = jt_host_to_dev(d, h);
        double u = 0.0;
        for (int p = 0; p < up_npart; ++p) u += up[(int64_t)p * pstride + x];
        if (dn) {
            double w = 0.0;
            for (int p = 0; p < dn_npart; ++p) w += dn[(int64_t)p * pstride + x];
            u *= w;
        }
        stage[h] = (S)u;
    }
}
