// stripelab.hip -- laboratory for the NO-split-K batch-1 4-bit g128 dequant-matvec on a load-time repacked
// "stripe16" layout (development tool, not shipped).  Every workgroup owns 16 whole output columns; its weight
// slice is contiguous in memory; one wave instruction = one 1-KiB block = 16 packed rows x 16 columns:
//     lane l -> column l % 16, packed rows 4 (l / 16) .. +3  (one dwordx4 = 4 rows of ONE column = 32 k),
// nibbles inside a word re-ordered so that the shift/and_or unpack yields NATURAL k pairs (2i, 2i+1).
// x is staged once per workgroup in LDS (+ sum x per 32 k); the group's {scale, -(64+z)*scale} per column sits in
// LDS too.  Reduction: 2 cross-lane steps + one LDS pass over the waves; y stored directly: no atomics, no workspace.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -o tools/stripelab tools/stripelab.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "../gptq-for-llama_amd/csrc/gptq_device.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return a;
}
__global__ void fill_u32(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = hash32((uint32_t)i * 2654435761u + seed);
}
// table entry: half2 {scale, 64 + zero + 1}
__global__ void fill_tab(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t h = hash32((uint32_t)i + seed);
        const half_t s = (half_t)(0.001f + 0.01f * (h >> 8) * (1.0f / 16777216.0f));
        const half_t z = (half_t)(65.0f + (float)(h & 15u));
        half2_t e = {s, z};
        p[i] = as_u32(e);
    }
}
__global__ void fill_x(half_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < 12; j++) s += (hash32((uint32_t)(i * 12 + j) + seed) >> 8) * (1.0f / 16777216.0f);
        p[i] = (half_t)(s - 6.0f);
    }
}

template <int CTRL>
GPTQ_DEV float dpp_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the four 16-lane rows of a wave (every lane ends with the total of its column): gfx950 permlane swaps
GPTQ_DEV float fold_rows(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s16 = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    const uint32_t u2 = __builtin_bit_cast(uint32_t, s16);
    auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
}

struct SP {
    const half_t *__restrict__ x;
    const uint32_t *__restrict__ R;     // [stripe][rb][set][64][4]
    const uint32_t *__restrict__ tab;   // [stripe][set][g][16] half2{s, 64+z+1}
    half_t *y;
    float *part;
    u64_t *dbg;
    int K, N, nrb, G;
};

template <int NS>
__global__ void ref_kernel(const SP p, double *y) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= p.N) return;
    const int stripe = n / 16, col = n % 16;
    double acc[NS];
    for (int s = 0; s < NS; s++) {
        acc[s] = 0;
        for (int rb = 0; rb < p.nrb; rb++) {
            const half2_t e = as_half2(p.tab[(((size_t)stripe * NS + s) * p.G + rb) * 16 + col]);
            for (int rq = 0; rq < 4; rq++)
                for (int j = 0; j < 4; j++) {
                    const int row = rb * 16 + rq * 4 + j;
                    const uint32_t w = p.R[((((size_t)stripe * p.nrb + rb) * NS + s) * 64 + rq * 16 + col) * 4 + j];
                    for (int i = 0; i < 4; i++) {
                        const int q0 = (w >> (4 * i)) & 15, q1 = (w >> (4 * (i + 4))) & 15;
                        acc[s] += (double)(float)p.x[row * 8 + 2 * i] * ((double)q0 + 64.0 - (double)(float)e[1]) * (double)(float)e[0];
                        acc[s] += (double)(float)p.x[row * 8 + 2 * i + 1] * ((double)q1 + 64.0 - (double)(float)e[1]) * (double)(float)e[0];
                    }
                }
        }
    }
    if (NS == 2) y[n] = acc[0] / (1.0 + exp(-acc[0])) * acc[1];
    else y[n] = acc[0];
}

// MODE 0 full, 2 loads + staging (no math, no output), 3 weight loads only, 4 full + per-wave timeline, 5 loads + staging + y store
// MATH 0: 3 shifts + 4 and_or + 4 dot2 per word, all offsets 64 (the product kernel's unpack)
//      1: 1 shift + 4 and_or + 4 dot2, offsets 1024 / 64 (nibbles at mantissa bits [3:0] and [7:4])
//      2: 1 shift + 4 and_or + 2 v_mfma_f32_4x4x4_16b_f16 (dot products on the matrix pipe; 3 of 4 result rows unused)
// DU = units (row blocks) requested per wave BEFORE x is staged; the rest is requested progressively, one unit ahead of
// the math, so that a wave is never parked in a full memory queue while data it could work on has arrived.
template <int NU, int NS, int NW, int DU, int MODE, int MATH>
__global__ void __launch_bounds__(NW * 64) k_stripe16(const SP p) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    u64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (MODE == 4) { st[0] = stamp_realtime(); st[1] = stamp_cycles(0); }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = NW * 64;
    constexpr int XP = (NU + 3) / 4;               // 16-byte pieces of x per thread
    constexpr int TP = (NS * NU + 15) / 16;        // 16-byte pieces of the table per thread
    const int K = p.K, nrb = p.nrb, G = p.G;
    half_t *xl = (half_t *)smem;
    float2 *xs4 = (float2 *)(smem + (size_t)K * 2);     // per 32 k: {sum x * OFF_k, sum x}
    float2 *tabf = (float2 *)(smem + (size_t)K * 2 + (size_t)(K / 32) * 8);
    float *red = (float *)(tabf + (size_t)NS * G * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stripe = blockIdx.x;
    const half2_t ones = {(half_t)1.f, (half_t)1.f};
    const uint32_t MSK = sreg_const(0x00F000F0u), MAG = vreg_const(0x54005400u);
    const uint32_t MSK0 = sreg_const(0x000F000Fu), MAG0 = vreg_const(0x64006400u);
    const half2_t c1024 = {(half_t)1024.f, (half_t)1024.f}, c64 = {(half_t)64.f, (half_t)64.f};

    u32x4 xv[XP], tv[TP];
    if constexpr (MODE != 3) {
#pragma unroll
        for (int i = 0; i < XP; i++) {
            const int idx = min(tid + i * T, K / 8 - 1);   // clamped: no branch in the load phase (a branch costs a vmcnt(0))
            xv[i] = *(const u32x4 *)(p.x + (size_t)idx * 8);
        }
        const uint32_t *tsrc = p.tab + (size_t)stripe * NS * G * 16;
        if constexpr (MATH != 3) {
#pragma unroll
            for (int i = 0; i < TP; i++) {
                const int idx = min(tid + i * T, NS * G * 4 - 1);
                tv[i] = *(const u32x4 *)(tsrc + (size_t)idx * 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    u32x4 w[NU][NS];
    uint32_t twg[NU][NS];
    const uint32_t *wbase = p.R + ((size_t)stripe * nrb * NS * 64 + lane) * 4;
    const uint32_t *tbaseg = p.tab + (size_t)stripe * NS * G * 16 + (lane & 15);
    auto issue = [&](int u) {
        const int rb = min(wave + NW * u, nrb - 1);   // ragged tail: re-read the last block (dropped below)
#pragma unroll
        for (int s = 0; s < NS; s++) {
            w[u][s] = __builtin_nontemporal_load((const u32x4 *)(wbase + ((size_t)rb * NS + s) * 256));
            if constexpr (MATH == 3) twg[u][s] = tbaseg[((size_t)s * G + rb) * 16];
        }
    };
#pragma unroll
    for (int u = 0; u < (DU < NU ? DU : NU); u++) issue(u);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 4) st[2] = stamp_cycles(0);
    if constexpr (MODE != 3) {
#pragma unroll
        for (int i = 0; i < XP; i++) {
            const int idx = tid + i * T;
            float s8 = 0.f, o8 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) s8 = __builtin_amdgcn_fdot2(as_half2(xv[i][q]), ones, s8, false);
            if constexpr (MATH == 0) {
                o8 = 64.f * s8;
            } else {
                o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][0]), c1024, o8, false);
                o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][1]), c64, o8, false);
                o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][2]), c1024, o8, false);
                o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][3]), c64, o8, false);
            }
            s8 += dpp_quad<0xB1>(s8);   // quad_perm [1,0,3,2]
            s8 += dpp_quad<0x4E>(s8);   // quad_perm [2,3,0,1]
            o8 += dpp_quad<0xB1>(o8);
            o8 += dpp_quad<0x4E>(o8);
            if (idx < K / 8) {
                *(u32x4 *)(xl + (size_t)idx * 8) = xv[i];
                if ((idx & 3) == 0) xs4[idx >> 2] = float2{o8, s8};
            }
        }
#pragma unroll
        for (int i = 0; i < (MATH == 3 ? 0 : TP); i++) {
            const int idx = tid + i * T;
            if (idx < NS * G * 4) {
                float4_t a, b;
                const half2_t e0 = as_half2(tv[i][0]), e1 = as_half2(tv[i][1]), e2 = as_half2(tv[i][2]), e3 = as_half2(tv[i][3]);
                a[0] = (float)e0[0]; a[1] = -((float)e0[1] - 64.f) * (float)e0[0]; a[2] = (float)e1[0]; a[3] = -((float)e1[1] - 64.f) * (float)e1[0];
                b[0] = (float)e2[0]; b[1] = -((float)e2[1] - 64.f) * (float)e2[0]; b[2] = (float)e3[0]; b[3] = -((float)e3[1] - 64.f) * (float)e3[0];
                *(float4_t *)(tabf + (size_t)idx * 4) = a;
                *(float4_t *)(tabf + (size_t)idx * 4 + 2) = b;
            }
        }
        __syncthreads();
    }
    if constexpr (MODE == 4) st[3] = stamp_cycles(0);
    const int rq = lane >> 4, col = lane & 15;
    float y[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) y[s] = 0.f;
    uint32_t xo = 0;
#pragma unroll
    for (int u = 0; u < NU; u++) {
        if (u + DU < NU) {
            issue(u + DU);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 2 || MODE == 3 || MODE == 5) {
#pragma unroll
            for (int s = 0; s < NS; s++) xo ^= w[u][s][0] ^ w[u][s][1] ^ w[u][s][2] ^ w[u][s][3];
        } else {
            const bool valid = wave + NW * u < nrb;
            const int rb = min(wave + NW * u, nrb - 1);
            const u32x4 *xp = (const u32x4 *)(xl + (size_t)(rb * 16 + rq * 4) * 8);
            u32x4 X[4];
#pragma unroll
            for (int j = 0; j < 4; j++) X[j] = xp[j];
            const float2 xs = xs4[rb * 4 + rq];
            if constexpr (MODE == 4) {
                if (u == 0) st[4] = stamp_cycles(w[0][0][0]);
                if (u == NU - 1) st[5] = stamp_cycles(w[NU - 1][NS - 1][0]);
            }
#pragma unroll
            for (int s = 0; s < NS; s++) {
                float acc = 0.f;
                float4_t accv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t ww = w[u][s][j];
                    if constexpr (MATH == 0) {
                        const half2_t t0 = as_half2(((ww << 4) & MSK) | MAG), t1 = as_half2((ww & MSK) | MAG);
                        const half2_t t2 = as_half2(((ww >> 4) & MSK) | MAG), t3 = as_half2(((ww >> 8) & MSK) | MAG);
                        acc = __builtin_amdgcn_fdot2(t0, as_half2(X[j][0]), acc, false);
                        acc = __builtin_amdgcn_fdot2(t1, as_half2(X[j][1]), acc, false);
                        acc = __builtin_amdgcn_fdot2(t2, as_half2(X[j][2]), acc, false);
                        acc = __builtin_amdgcn_fdot2(t3, as_half2(X[j][3]), acc, false);
                    } else {
                        const uint32_t hi = ww >> 8;
                        const uint32_t t0 = (ww & MSK0) | MAG0, t1 = (ww & MSK) | MAG, t2 = (hi & MSK0) | MAG0, t3 = (hi & MSK) | MAG;
                        if constexpr (MATH == 1) {
                            acc = __builtin_amdgcn_fdot2(as_half2(t0), as_half2(X[j][0]), acc, false);
                            acc = __builtin_amdgcn_fdot2(as_half2(t1), as_half2(X[j][1]), acc, false);
                            acc = __builtin_amdgcn_fdot2(as_half2(t2), as_half2(X[j][2]), acc, false);
                            acc = __builtin_amdgcn_fdot2(as_half2(t3), as_half2(X[j][3]), acc, false);
                        } else {
                            const h4_t B1 = __builtin_bit_cast(h4_t, u32x2{t0, t1}), B2 = __builtin_bit_cast(h4_t, u32x2{t2, t3});
                            const h4_t A1 = __builtin_bit_cast(h4_t, u32x2{X[j][0], X[j][1]}), A2 = __builtin_bit_cast(h4_t, u32x2{X[j][2], X[j][3]});
                            accv = __builtin_amdgcn_mfma_f32_4x4x4f16(A1, B1, accv, 0, 0, 0);
                            accv = __builtin_amdgcn_mfma_f32_4x4x4f16(A2, B2, accv, 0, 0, 0);
                        }
                    }
                }
                if constexpr (MATH >= 2) acc = accv[0];
                float2 e;
                if constexpr (MATH == 3) {
                    const half2_t eh = as_half2(twg[u][s]);
                    e.x = (float)eh[0];
                    e.y = -((float)eh[1] - 64.f) * e.x;
                } else {
                    e = tabf[((size_t)s * G + rb) * 16 + col];
                }
                if (!valid) e = float2{0.f, 0.f};
                y[s] = fmaf(e.x, acc - xs.x, y[s]);
                y[s] = fmaf(e.y, xs.y, y[s]);
            }
        }
    }
    if constexpr (MODE == 2 || MODE == 3) {
        if (xo == 0x9e3779b9u) p.part[blockIdx.x] = 1.f;
        return;
    }
    if constexpr (MODE == 5) y[0] = (float)(xo & 1u);
#pragma unroll
    for (int s = 0; s < NS; s++) y[s] = fold_rows(y[s]);
    if constexpr (MODE == 4) st[6] = stamp_cycles(__builtin_bit_cast(uint32_t, y[0]));
    if (lane < 16) {
#pragma unroll
        for (int s = 0; s < NS; s++) red[(wave * NS + s) * 16 + lane] = y[s];
    }
    __syncthreads();
    if (tid < 16) {
        float a[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            a[s] = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; wv++) a[s] += red[(wv * NS + s) * 16 + tid];
        }
        float v = a[0];
        if constexpr (NS == 2) v = a[0] * (1.0f / (1.0f + __expf(-a[0]))) * a[1];
        half_t *dst = p.y + stripe * 16 + tid;
        if constexpr (MODE == 6) { if (v == 123.456f) *dst = (half_t)v; }
        else if constexpr (MODE == 7) __hip_atomic_store((unsigned short *)dst, __builtin_bit_cast(unsigned short, (half_t)v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if constexpr (MODE == 8) __builtin_nontemporal_store((half_t)v, dst);
        else *dst = (half_t)v;
    }
    if constexpr (MODE == 4) {
        st[7] = stamp_cycles(0);
        const u64_t te = stamp_realtime();
        if (lane == 0) {
            u64_t *d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 10;
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = st[i];
            d[8] = te;
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[9] = xcc;
        }
    }
}

static size_t lds_bytes(const SP &p, int NS, int NW) {
    return (size_t)p.K * 2 + (size_t)(p.K / 32) * 8 + (size_t)NS * p.G * 16 * 8 + (size_t)NW * NS * 16 * 4;
}
template <int NU, int NS, int NW, int DU, int MODE, int MATH>
static void launch(const SP &p, hipStream_t s) {
    hipLaunchKernelGGL((k_stripe16<NU, NS, NW, DU, MODE, MATH>), dim3(p.N / 16), dim3(NW * 64), lds_bytes(p, NS, NW), s, p);
}

// LDS-DMA loads only: every wave streams its row blocks into a PRIVATE LDS region (global_load_lds_dwordx4, nt), DEPTH blocks
// in flight, reads one dword of each landed block back (xor) -- the floor of an LDS-DMA version of the kernel above.
template <int NU, int NS, int NW, int DEPTH>
__global__ void __launch_bounds__(NW * 64) k_glds_only(const SP p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stripe = blockIdx.x, nrb = p.nrb;
    unsigned char *mine = smem + wave * (DEPTH * NS * 1024);
    const uint32_t *wbase = p.R + ((size_t)stripe * nrb * NS * 64 + lane) * 4;
    auto issue = [&](int u) {
        const int rb = min(wave + NW * u, nrb - 1);
#pragma unroll
        for (int s = 0; s < NS; s++)
            __builtin_amdgcn_global_load_lds((gptr_t)(wbase + ((size_t)rb * NS + s) * 256), (lptr_t)(mine + ((u % DEPTH) * NS + s) * 1024), 16, 0, 2);
    };
#pragma unroll
    for (int u = 0; u < (DEPTH < NU ? DEPTH : NU); u++) issue(u);
    uint32_t xo = 0;
#pragma unroll
    for (int u = 0; u < NU; u++) {
        // blocks u+1 .. min(u+DEPTH, NU)-1 may stay in flight
        constexpr int dummy = 0; (void)dummy;
        const int left = (u + DEPTH < NU ? DEPTH : NU - u) - 1;
        if (left >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NS) : "memory");
        else if (left == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NS) : "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * NS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; s++) xo ^= *(const uint32_t *)(mine + ((u % DEPTH) * NS + s) * 1024 + lane * 16);
        if (u + DEPTH < NU) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(u + DEPTH);
        }
    }
    if (xo == 0x9e3779b9u) p.part[blockIdx.x] = 1.f;
}
template <int NU, int NS, int NW, int DEPTH>
static void launch_glds(const SP &p, hipStream_t s) {
    hipLaunchKernelGGL((k_glds_only<NU, NS, NW, DEPTH>), dim3(p.N / 16), dim3(NW * 64), NW * DEPTH * NS * 1024, s, p);
}

// Barrier-free staging: wave w only ever touches the x of ITS row blocks (w, w + NW, ...), so it loads those 256-byte slices itself
// (one dwordx4 per lane covers four row blocks), keeps them and their lane-block sums in a PRIVATE LDS region and reads the table
// entries of its row blocks straight from global -- no workgroup barrier before the math, no shared table.  MATH 2 arithmetic.
template <int NU, int NS, int NW, int DU>
__global__ void __launch_bounds__(NW * 64) k_stripe16_priv(const SP p) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XI = (NU + 3) / 4;                 // x instructions per lane (four row blocks each)
    constexpr int WB = XI * 4 * (256 + 32);          // private bytes per wave: per row block 256 B of x + 4 float2 sums
    const int nrb = p.nrb, G = p.G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stripe = blockIdx.x;
    unsigned char *mine = smem + wave * WB;
    float *red = (float *)(smem + NW * WB);
    const half2_t ones = {(half_t)1.f, (half_t)1.f};
    const uint32_t MSK = sreg_const(0x00F000F0u), MAG = vreg_const(0x54005400u);
    const uint32_t MSK0 = sreg_const(0x000F000Fu), MAG0 = vreg_const(0x64006400u);
    const half2_t c1024 = {(half_t)1024.f, (half_t)1024.f}, c64 = {(half_t)64.f, (half_t)64.f};
    const int rq = lane >> 4, col = lane & 15;

    u32x4 xv[XI];
#pragma unroll
    for (int i = 0; i < XI; i++) {
        const int u = 4 * i + (lane >> 4);
        const int rb = min(wave + NW * u, nrb - 1);
        xv[i] = *(const u32x4 *)(p.x + (size_t)rb * 128 + (lane & 15) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 w[NU][NS];
    uint32_t tw[NU][NS];
    const uint32_t *wbase = p.R + ((size_t)stripe * nrb * NS * 64 + lane) * 4;
    const uint32_t *tbase = p.tab + (size_t)stripe * NS * G * 16 + col;
    auto issue = [&](int u) {
        const int rb = min(wave + NW * u, nrb - 1);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            w[u][s] = __builtin_nontemporal_load((const u32x4 *)(wbase + ((size_t)rb * NS + s) * 256));
            tw[u][s] = tbase[((size_t)s * G + rb) * 16];
        }
    };
#pragma unroll
    for (int u = 0; u < (DU < NU ? DU : NU); u++) issue(u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < XI; i++) {
        float s8 = 0.f, o8 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) s8 = __builtin_amdgcn_fdot2(as_half2(xv[i][q]), ones, s8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][0]), c1024, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][1]), c64, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][2]), c1024, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][3]), c64, o8, false);
        s8 += dpp_quad<0xB1>(s8);
        s8 += dpp_quad<0x4E>(s8);
        o8 += dpp_quad<0xB1>(o8);
        o8 += dpp_quad<0x4E>(o8);
        const int u = 4 * i + (lane >> 4), pc = lane & 15;
        *(u32x4 *)(mine + u * 288 + pc * 16) = xv[i];
        if ((pc & 3) == 0) *(float2 *)(mine + u * 288 + 256 + (pc >> 2) * 8) = float2{o8, s8};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float y[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) y[s] = 0.f;
#pragma unroll
    for (int u = 0; u < NU; u++) {
        if (u + DU < NU) {
            issue(u + DU);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool valid = wave + NW * u < nrb;
        const u32x4 *xp = (const u32x4 *)(mine + u * 288 + rq * 64);
        u32x4 X[4];
#pragma unroll
        for (int j = 0; j < 4; j++) X[j] = xp[j];
        const float2 xs = *(const float2 *)(mine + u * 288 + 256 + rq * 8);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            float4_t accv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t ww = w[u][s][j];
                const uint32_t hi = ww >> 8;
                const uint32_t t0 = (ww & MSK0) | MAG0, t1 = (ww & MSK) | MAG, t2 = (hi & MSK0) | MAG0, t3 = (hi & MSK) | MAG;
                const h4_t B1 = __builtin_bit_cast(h4_t, u32x2{t0, t1}), B2 = __builtin_bit_cast(h4_t, u32x2{t2, t3});
                const h4_t A1 = __builtin_bit_cast(h4_t, u32x2{X[j][0], X[j][1]}), A2 = __builtin_bit_cast(h4_t, u32x2{X[j][2], X[j][3]});
                accv = __builtin_amdgcn_mfma_f32_4x4x4f16(A1, B1, accv, 0, 0, 0);
                accv = __builtin_amdgcn_mfma_f32_4x4x4f16(A2, B2, accv, 0, 0, 0);
            }
            const half2_t e = as_half2(tw[u][s]);
            const float sc = valid ? (float)e[0] : 0.f;
            const float zs = -((float)e[1] - 64.f) * sc;
            y[s] = fmaf(sc, accv[0] - xs.x, y[s]);
            y[s] = fmaf(zs, xs.y, y[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < NS; s++) y[s] = fold_rows(y[s]);
    if (lane < 16) {
#pragma unroll
        for (int s = 0; s < NS; s++) red[(wave * NS + s) * 16 + lane] = y[s];
    }
    __syncthreads();
    if (tid < 16) {
        float a[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            a[s] = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; wv++) a[s] += red[(wv * NS + s) * 16 + tid];
        }
        float v = a[0];
        if constexpr (NS == 2) v = a[0] * (1.0f / (1.0f + __expf(-a[0]))) * a[1];
        p.y[stripe * 16 + tid] = (half_t)v;
    }
}
template <int NU, int NS, int NW, int DU>
__global__ void __launch_bounds__(NW * 64) k_stripe16_pers(const SP p) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XI = (NU + 3) / 4;                 // x instructions per lane (four row blocks each)
    constexpr int WB = XI * 4 * (256 + 32);          // private bytes per wave: per row block 256 B of x + 4 float2 sums
    const int nrb = p.nrb, G = p.G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *mine = smem + wave * WB;
    float *red = (float *)(smem + NW * WB);
    const half2_t ones = {(half_t)1.f, (half_t)1.f};
    const uint32_t MSK = sreg_const(0x00F000F0u), MAG = vreg_const(0x54005400u);
    const uint32_t MSK0 = sreg_const(0x000F000Fu), MAG0 = vreg_const(0x64006400u);
    const half2_t c1024 = {(half_t)1024.f, (half_t)1024.f}, c64 = {(half_t)64.f, (half_t)64.f};
    const int rq = lane >> 4, col = lane & 15;

    u32x4 xv[XI];
#pragma unroll
    for (int i = 0; i < XI; i++) {
        const int u = 4 * i + (lane >> 4);
        const int rb = min(wave + NW * u, nrb - 1);
        xv[i] = *(const u32x4 *)(p.x + (size_t)rb * 128 + (lane & 15) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 w[NU][NS];
    uint32_t tw[NU][NS];
    const uint32_t *wbase = nullptr, *tbase = nullptr;
    auto set_stripe = [&](int stripe) {
        wbase = p.R + ((size_t)stripe * nrb * NS * 64 + lane) * 4;
        tbase = p.tab + (size_t)stripe * NS * G * 16 + col;
    };
    const int nstripes = p.N / 16;
    set_stripe(blockIdx.x);
    auto issue = [&](int u) {
        const int rb = min(wave + NW * u, nrb - 1);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            w[u][s] = __builtin_nontemporal_load((const u32x4 *)(wbase + ((size_t)rb * NS + s) * 256));
            tw[u][s] = tbase[((size_t)s * G + rb) * 16];
        }
    };
#pragma unroll
    for (int u = 0; u < (DU < NU ? DU : NU); u++) issue(u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < XI; i++) {
        float s8 = 0.f, o8 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) s8 = __builtin_amdgcn_fdot2(as_half2(xv[i][q]), ones, s8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][0]), c1024, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][1]), c64, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][2]), c1024, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][3]), c64, o8, false);
        s8 += dpp_quad<0xB1>(s8);
        s8 += dpp_quad<0x4E>(s8);
        o8 += dpp_quad<0xB1>(o8);
        o8 += dpp_quad<0x4E>(o8);
        const int u = 4 * i + (lane >> 4), pc = lane & 15;
        *(u32x4 *)(mine + u * 288 + pc * 16) = xv[i];
        if ((pc & 3) == 0) *(float2 *)(mine + u * 288 + 256 + (pc >> 2) * 8) = float2{o8, s8};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int stripe = blockIdx.x; stripe < nstripes; stripe += gridDim.x) {
    float y[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) y[s] = 0.f;
#pragma unroll
        for (int u = 0; u < NU; u++) {
            if (u + DU < NU) {
                issue(u + DU);
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool valid = wave + NW * u < nrb;
            const u32x4 *xp = (const u32x4 *)(mine + u * 288 + rq * 64);
            u32x4 X[4];
#pragma unroll
            for (int j = 0; j < 4; j++) X[j] = xp[j];
            const float2 xs = *(const float2 *)(mine + u * 288 + 256 + rq * 8);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                float4_t accv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t ww = w[u][s][j];
                    const uint32_t hi = ww >> 8;
                    const uint32_t t0 = (ww & MSK0) | MAG0, t1 = (ww & MSK) | MAG, t2 = (hi & MSK0) | MAG0, t3 = (hi & MSK) | MAG;
                    const h4_t B1 = __builtin_bit_cast(h4_t, u32x2{t0, t1}), B2 = __builtin_bit_cast(h4_t, u32x2{t2, t3});
                    const h4_t A1 = __builtin_bit_cast(h4_t, u32x2{X[j][0], X[j][1]}), A2 = __builtin_bit_cast(h4_t, u32x2{X[j][2], X[j][3]});
                    accv = __builtin_amdgcn_mfma_f32_4x4x4f16(A1, B1, accv, 0, 0, 0);
                    accv = __builtin_amdgcn_mfma_f32_4x4x4f16(A2, B2, accv, 0, 0, 0);
                }
                const half2_t e = as_half2(tw[u][s]);
                const float sc = valid ? (float)e[0] : 0.f;
                const float zs = -((float)e[1] - 64.f) * sc;
                y[s] = fmaf(sc, accv[0] - xs.x, y[s]);
                y[s] = fmaf(zs, xs.y, y[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; s++) y[s] = fold_rows(y[s]);
        if (lane < 16) {
#pragma unroll
            for (int s = 0; s < NS; s++) red[(wave * NS + s) * 16 + lane] = y[s];
        }
        __syncthreads();
        if (tid < 16) {
            float a[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                a[s] = 0.f;
#pragma unroll
                for (int wv = 0; wv < NW; wv++) a[s] += red[(wv * NS + s) * 16 + tid];
            }
            float v = a[0];
            if constexpr (NS == 2) v = a[0] * (1.0f / (1.0f + __expf(-a[0]))) * a[1];
            p.y[stripe * 16 + tid] = (half_t)v;
        }
        const int nxt = stripe + gridDim.x;
        if (nxt < nstripes) {
            set_stripe(nxt);
#pragma unroll
            for (int u = 0; u < (DU < NU ? DU : NU); u++) issue(u);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
}
template <int NU, int NS, int NW, int DU, int PERCU>
static void launch_pers(const SP &p, hipStream_t s) {
    constexpr int XI = (NU + 3) / 4;
    const int grid = std::min(p.N / 16, 256 * PERCU);
    hipLaunchKernelGGL((k_stripe16_pers<NU, NS, NW, DU>), dim3(grid), dim3(NW * 64), NW * XI * 4 * 288 + NW * NS * 64, s, p);
}
template <int NU, int NS, int NW, int DU>
static void launch_priv(const SP &p, hipStream_t s) {
    constexpr int XI = (NU + 3) / 4;
    hipLaunchKernelGGL((k_stripe16_priv<NU, NS, NW, DU>), dim3(p.N / 16), dim3(NW * 64), NW * XI * 4 * 288 + NW * NS * 64, s, p);
}
typedef void (*launch_fn)(const SP &, hipStream_t);
struct WSet { uint32_t *R; uint32_t *tab; };

static float time_graph(launch_fn fn, SP base, const std::vector<WSet> &sets, hipStream_t s, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    static const bool freshx = getenv("LAB_FRESHX") != nullptr;   // x rewritten by a producer launch before every kernel (as in a real decode chain)
    static const int freshx_mode = freshx ? atoi(getenv("LAB_FRESHX")) : 0;
    static half_t *dummy_x = nullptr;
    if (freshx && !dummy_x) CK(hipMalloc(&dummy_x, 65536 * 2));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    int it = 0;
    // round 4 fix: >= 128 kernel nodes per graph, cycling through the weight sets.  Round 3 captured sets.size() nodes, so with LAB_NBUF=1..3
    // the figure was the cadence of launching a 1-3 node graph (~10.4 us), not a read path (tools/warmlab.hip A).
    const size_t nodes = ((std::max<size_t>(128, sets.size()) + sets.size() - 1) / sets.size()) * sets.size();
    for (size_t node = 0; node < nodes; node++) {
        const WSet &w = sets[node % sets.size()];
        SP p = base; p.R = w.R; p.tab = w.tab;
        // LAB_FRESHX=1: x itself is rewritten; LAB_FRESHX=2 (control): the same producer launch writes a buffer nobody reads
        if (freshx) hipLaunchKernelGGL(fill_x, dim3(64), dim3(256), 0, s, freshx_mode == 2 ? dummy_x : (half_t *)base.x, (size_t)base.K, 77u + (it++ % 2));
        fn(p, s);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (reps * nodes);
}
static float best_of(launch_fn fn, SP base, const std::vector<WSet> &sets, hipStream_t s) {
    float t = 1e9f;
    for (int r = 0; r < 3; r++) t = fminf(t, time_graph(fn, base, sets, s, 5));
    return t;
}

struct Ctx { hipStream_t s; double bytes; double *yref; std::vector<double> *href; std::vector<half_t> *hy; };

template <int NU, int NS, int NW, int DU, int MATH>
static void run_math(SP base, const std::vector<WSet> &sets, Ctx &c) {
    hipStream_t s = c.s;
    const size_t lds = lds_bytes(base, NS, NW);
    if (lds > 48 * 1024) {
        CK(hipFuncSetAttribute((const void *)k_stripe16<NU, NS, NW, DU, 0, MATH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void *)k_stripe16<NU, NS, NW, DU, 4, MATH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int N = base.N;
    SP p = base; p.R = sets[1 % sets.size()].R; p.tab = sets[1 % sets.size()].tab;
    CK(hipMemsetAsync(base.y, 0, N * 2, s));
    hipLaunchKernelGGL((ref_kernel<NS>), dim3((N + 255) / 256), dim3(256), 0, s, p, c.yref);
    launch<NU, NS, NW, DU, 0, MATH>(p, s);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(c.href->data(), c.yref, N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.hy->data(), base.y, N * 2, hipMemcpyDeviceToHost));
    double mx = 0, err = 0;
    for (int n = 0; n < N; n++) { mx = fmax(mx, fabs((*c.href)[n])); err = fmax(err, fabs((*c.href)[n] - (double)(float)(*c.hy)[n])); }
    const float t0 = best_of(launch<NU, NS, NW, DU, 0, MATH>, base, sets, s);
    const float t6 = best_of(launch<NU, NS, NW, DU, 6, MATH>, base, sets, s);
    const float t7 = best_of(launch<NU, NS, NW, DU, 7, MATH>, base, sets, s);
    const float t8 = best_of(launch<NU, NS, NW, DU, 8, MATH>, base, sets, s);
    printf("    math %d: full %6.2f us %5.0f GB/s err %.1e | no store %6.2f | sc1 store %6.2f | nt store %6.2f\n", MATH, t0, c.bytes / t0 / 1e3, err / mx, t6, t7, t8);
    if (getenv("LAB_TIMELINE") && MATH == 2) {
        const int nwg = N / 16, nw = nwg * NW;
        u64_t *dbg; CK(hipMalloc(&dbg, (size_t)nw * 10 * 8));
        for (int i = 0; i < 6; i++) { SP q = base; q.R = sets[i % sets.size()].R; q.tab = sets[i % sets.size()].tab; q.dbg = dbg; launch<NU, NS, NW, DU, 4, MATH>(q, s); }
        CK(hipStreamSynchronize(s));
        std::vector<u64_t> h((size_t)nw * 10);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        u64_t r0 = ~0ull, r1 = 0;
        for (int i = 0; i < nw; i++) { r0 = std::min(r0, h[i * 10 + 0]); r1 = std::max(r1, h[i * 10 + 8]); }
        printf("      timeline (%d waves): kernel span %.2f us\n", nw, (r1 - r0) / 100.0);
        const char *nm[] = {"start(real,us)", "loads issued", "x staged", "first w", "last w", "math done", "y stored", "end(real,us)"};
        for (int k = 0; k < 8; k++) {
            std::vector<double> v(nw);
            for (int i = 0; i < nw; i++) {
                const u64_t *d = &h[i * 10];
                if (k == 0) v[i] = (d[0] - r0) / 100.0;
                else if (k == 7) v[i] = (d[8] - r0) / 100.0;
                else v[i] = (double)(d[k + 1] - d[1]);
            }
            std::sort(v.begin(), v.end());
            printf("        %-16s min %8.2f  p10 %8.2f  p50 %8.2f  p90 %8.2f  max %8.2f %s\n", nm[k], v[0], v[nw / 10], v[nw / 2], v[nw * 9 / 10], v[nw - 1], (k == 0 || k == 7) ? "us" : "cycles");
        }
        CK(hipFree(dbg));
    }
}

template <int NU, int NS, int NW, int DU>
static void run_config(const char *name, SP base, const std::vector<WSet> &sets, Ctx &c) {
    const size_t lds = lds_bytes(base, NS, NW);
    if (lds > 48 * 1024) {
        CK(hipFuncSetAttribute((const void *)k_stripe16<NU, NS, NW, DU, 2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void *)k_stripe16<NU, NS, NW, DU, 3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void *)k_stripe16<NU, NS, NW, DU, 5, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const float t2 = best_of(launch<NU, NS, NW, DU, 2, 0>, base, sets, c.s);
    const float t3 = best_of(launch<NU, NS, NW, DU, 3, 0>, base, sets, c.s);
    const float t5 = best_of(launch<NU, NS, NW, DU, 5, 0>, base, sets, c.s);
    printf("  %-22s NU%-2d NS%d NW%-2d DU%d wgs %5d lds %5zu | loadsonly %6.2f | +stage %6.2f | +stage+store %6.2f\n", name, NU, NS, NW, DU, base.N / 16, lds, t3, t2, t5);
    {
        constexpr int D4 = NU < 4 ? NU : 4;
        CK(hipFuncSetAttribute((const void *)k_glds_only<NU, NS, NW, D4>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * D4 * NS * 1024));
        CK(hipFuncSetAttribute((const void *)k_glds_only<NU, NS, NW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 2 * NS * 1024));
        const float g4 = best_of(launch_glds<NU, NS, NW, D4>, base, sets, c.s);
        const float g2 = best_of(launch_glds<NU, NS, NW, 2>, base, sets, c.s);
        printf("    LDS-DMA loads only: depth %d %6.2f us | depth 2 %6.2f us\n", D4, g4, g2);
    }
    run_math<NU, NS, NW, DU, 0>(base, sets, c);
    run_math<NU, NS, NW, DU, 2>(base, sets, c);
    run_math<NU, NS, NW, DU, 3>(base, sets, c);
    {
        const int N = base.N;
        SP p = base; p.R = sets[1 % sets.size()].R; p.tab = sets[1 % sets.size()].tab;
        CK(hipMemsetAsync(base.y, 0, N * 2, c.s));
        hipLaunchKernelGGL((ref_kernel<NS>), dim3((N + 255) / 256), dim3(256), 0, c.s, p, c.yref);
        launch_priv<NU, NS, NW, DU>(p, c.s);
        CK(hipStreamSynchronize(c.s));
        CK(hipMemcpy(c.href->data(), c.yref, N * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c.hy->data(), base.y, N * 2, hipMemcpyDeviceToHost));
        double mx = 0, err = 0;
        for (int n = 0; n < N; n++) { mx = fmax(mx, fabs((*c.href)[n])); err = fmax(err, fabs((*c.href)[n] - (double)(float)(*c.hy)[n])); }
        const float tp = best_of(launch_priv<NU, NS, NW, DU>, base, sets, c.s);
        printf("    private staging (no barrier), math 2: full %6.2f us %5.0f GB/s err %.1e\n", tp, c.bytes / tp / 1e3, err / mx);
        CK(hipMemsetAsync(base.y, 0, N * 2, c.s));
        launch_pers<NU, NS, NW, DU, 1>(p, c.s);
        CK(hipStreamSynchronize(c.s));
        CK(hipMemcpy(c.hy->data(), base.y, N * 2, hipMemcpyDeviceToHost));
        double err2 = 0;
        for (int n = 0; n < N; n++) err2 = fmax(err2, fabs((*c.href)[n] - (double)(float)(*c.hy)[n]));
        const float t1 = best_of(launch_pers<NU, NS, NW, DU, 1>, base, sets, c.s);
        const float t2 = best_of(launch_pers<NU, NS, NW, DU, 2>, base, sets, c.s);
        printf("    persistent stripes (x staged once): 256 wgs %6.2f us | 512 wgs %6.2f us  err %.1e\n", t1, t2, err2 / mx);
    }
}

int main(int argc, char **argv) {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *part; CK(hipMalloc(&part, 64 << 20));
    struct Shape { int K, N, NS; const char *name; };
    const Shape shapes[] = {{4096, 4096, 1, "o 4096x4096"}, {4096, 12288, 1, "qkv 4096x12288"}, {4096, 11008, 2, "gate/up 2x4096x11008"},
                            {11008, 4096, 1, "down 11008x4096"}};
    for (const Shape &sh : shapes) {
        const int K = sh.K, N = sh.N, NS = sh.NS, G = K / 128, nrb = K / 128;
        const size_t r_n = (size_t)(K / 8) * N * NS, t_n = (size_t)NS * G * N;
        // algorithmic bytes of the ORIGINAL format: qweight + qzeros + scales per set, x, y
        const double bytes = NS * ((double)(K / 8) * N * 4 + (double)G * (N / 8) * 4 + (double)G * N * 2) + 2.0 * K + 2.0 * N;
        int nbuf = (int)((400ull << 20) / (r_n * 4 + t_n * 4)) + 1;
        if (getenv("LAB_NBUF")) nbuf = atoi(getenv("LAB_NBUF"));
        std::vector<WSet> sets(nbuf);
        for (int i = 0; i < nbuf; i++) {
            CK(hipMalloc(&sets[i].R, r_n * 4)); CK(hipMalloc(&sets[i].tab, t_n * 4));
            hipLaunchKernelGGL(fill_u32, dim3(2048), dim3(256), 0, s, sets[i].R, r_n, 1000u + i);
            hipLaunchKernelGGL(fill_tab, dim3(256), dim3(256), 0, s, sets[i].tab, t_n, 2000u + i);
        }
        half_t *x, *y; double *yref;
        CK(hipMalloc(&x, K * 2)); CK(hipMalloc(&y, N * 2)); CK(hipMalloc(&yref, N * 8));
        hipLaunchKernelGGL(fill_x, dim3(64), dim3(256), 0, s, x, (size_t)K, 77u);
        CK(hipStreamSynchronize(s));
        printf("== %s: %.2f MB algorithmic (%.2f MB stripe16 traffic) x %d sets\n", sh.name, bytes / 1e6, (r_n * 4 + t_n * 4 + 2.0 * K + 2.0 * N) / 1e6, nbuf);
        SP base{}; base.x = x; base.y = y; base.part = part; base.K = K; base.N = N; base.nrb = nrb; base.G = G;
        std::vector<double> href(N); std::vector<half_t> hy(N);
        Ctx c{s, bytes, yref, &href, &hy};
        if (K == 4096 && NS == 1) {
            run_config<4, 1, 8, 2>(sh.name, base, sets, c);
        } else if (K == 4096 && NS == 2) {
            run_config<4, 2, 8, 1>(sh.name, base, sets, c);
            run_config<4, 2, 8, 2>(sh.name, base, sets, c);
        } else {
            run_config<11, 1, 8, 4>(sh.name, base, sets, c);
            run_config<11, 1, 8, 3>(sh.name, base, sets, c);
        }
        for (auto &w : sets) { CK(hipFree(w.R)); CK(hipFree(w.tab)); }
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(yref));
    }
    return 0;
}
