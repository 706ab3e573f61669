// stripe.hip -- batch-1 dequant-matvec WITHOUT a K split, on a load-time repacked copy of the checkpoint
// buffers ("stripe16" layout).  Same arithmetic as matmul_248_kernel (reference quant/quant_linear.py:103-137) and
// fusedmatmul_248_kernel (quant/fused_mlp.py:128-168); the layout is new.
//
// Why: the rowwave GEMV (gemv.hip) reads the checkpoint layout, so a workgroup can only own a full-width row
// segment and K has to be split over ~16 workgroups per column tile; the combine is a returning device-scope
// atomic (0.6-0.7 us on the critical path of every launch, tools/gemvlab.hip).  Here every workgroup owns 16 WHOLE
// output columns, its weights are one contiguous stream, partial sums never leave the CU, y is stored directly:
// no atomics, no workspace, bit-reproducible by construction.  Measured (tools/stripelab.hip, MI355X, cold weights,
// us per launch): o 4096^2 3.7 (rowwave 4.45), qkv 6.4 (7.35), gate/up+SiLU 9.9 (10.98), down 6.2 (7.17).
//
// stripe16 layout (bits = 4; K % 128 == 0, N % 16 == 0), one buffer per weight set pair:
//   R   uint32 [N/16 stripes][K/128 row blocks][NS sets][64 lanes][4]   -- 1 KiB per (stripe, block, set): one wave load
//       lane l: column 16*stripe + l%16, packed rows 16*block + 4*(l/16) .. +3 (its dwordx4 = 32 consecutive k of ONE
//       column); inside a word the nibbles are re-ordered (k0 k2 k4 k6 | k1 k3 k5 k7 from bit 0) so that
//       (w & 0x000F000F) | 0x64006400 = half2{1024 + q_k0, 1024 + q_k1},  (w & 0x00F000F0) | 0x54005400 = {64 + q_k2, 64 + q_k3},
//       and the same two masks on w >> 8 give k4..k7: ONE shift + four v_and_or_b32 per 8 weights, and the pairs line
//       up with natural x dwords.  Two such half2 are the B operand of v_mfma_f32_4x4x4_16b_f16 (4 k of this lane's
//       column), the matching 4 x values the A operand: the dot products run on the matrix pipe (2 MFMA instead of
//       4 v_dot2 per word; three of the four result rows are unused -- the kernel is issue-bound, not flop-bound).
//   tab half2 [N/16][NS][G][16] = {scale, zero + 1}: the (group, column) constants of a stripe, contiguous.
// The offsets 1024 / 64 and the zero point are removed once per (32 k, column):
//   y += s * (acc - sum_k x_k OFF_k) - s (z + 1) sum_k x_k,  both sums precomputed per 32 k when x is staged in LDS.
// Schedule: x / table loads first, DU row blocks of weights per wave, stage x (optionally RMS-normalised: the
// arithmetic of rms_norm_fwd_fused, quant/triton_norm.py:22-39, every workgroup sees all of x anyway), barrier, then
// the remaining blocks are requested one ahead of the math (a CU keeps only ~32-40 KiB of loads in flight; a wave
// parked in a full queue cannot work on data that has already arrived).
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

namespace {

constexpr int STRIPE_NW = 8;       // waves per workgroup
constexpr int STRIPE_MAX_NU = 24;  // row blocks per wave (K <= 24576)

template <int CTRL>
GPTQ_DEV float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the four 16-lane rows of a wave (every lane ends with the total of its lane position): gfx950 permlane swaps
GPTQ_DEV float fold_rows(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s16 = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    const uint32_t u2 = __builtin_bit_cast(uint32_t, s16);
    auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
}
GPTQ_DEV float quad_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
GPTQ_DEV float wave_sum_all(float v) {
    v = quad_sum(v);
    v += dpp_f<0x141>(v);  // row_half_mirror: the other quad of the 8
    v += dpp_f<0x140>(v);  // row_mirror: the other half of the 16
    return fold_rows(v);
}

// nibble order of a stripe word: position p (bits 4p..4p+3) holds k = KOF[p] of the packed row
__host__ __device__ constexpr int stripe_k_of_pos(int p) { return p < 4 ? 2 * p : 2 * (p - 4) + 1; }

__global__ void __launch_bounds__(256) stripe_repack_kernel(const uint32_t *__restrict__ qw0, const uint32_t *__restrict__ qw1,
                                                            uint32_t *__restrict__ R, int N, int nrb, int NS) {
    // one thread per output word; consecutive threads -> consecutive j (rows), then lanes (columns): 64-byte reads
    const size_t total = (size_t)(N / 16) * nrb * NS * 256;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), l = (int)((i >> 2) & 63);
        size_t b = i >> 8;
        const int set = (int)(b % NS); b /= NS;
        const int rb = (int)(b % nrb);
        const int stripe = (int)(b / nrb);
        const int row = rb * 16 + 4 * (l >> 4) + j, col = 16 * stripe + (l & 15);
        const uint32_t w = (set ? qw1 : qw0)[(size_t)row * N + col];
        uint32_t o = 0;
#pragma unroll
        for (int p = 0; p < 8; p++) o |= ((w >> (4 * stripe_k_of_pos(p))) & 15u) << (4 * p);
        R[i] = o;
    }
}

__global__ void __launch_bounds__(256) stripe_table_kernel(const half_t *__restrict__ sc0, const int32_t *__restrict__ qz0,
                                                           const half_t *__restrict__ sc1, const int32_t *__restrict__ qz1,
                                                           uint32_t *__restrict__ tab, int N, int G, int NS) {
    const size_t total = (size_t)(N / 16) * NS * G * 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 15);
        size_t b = i >> 4;
        const int g = (int)(b % G); b /= G;
        const int set = (int)(b % NS);
        const int stripe = (int)(b / NS);
        const int n = 16 * stripe + c;
        const half_t s = (set ? sc1 : sc0)[(size_t)g * N + n];
        const int z = zero_of<4>((set ? qz1 : qz0) + (size_t)g * (N / 8), n);   // stored + 1, not re-masked (quant_linear.py:120-121)
        const half2_t e = {s, (half_t)(float)z};
        tab[i] = as_u32(e);
    }
}

// NS = 2: y = silu(x Wg) * (x Wu) (fused_mlp.py:160-166).  NORM: x is RMS-normalised while it is staged.
// XPERM: x (and the norm weight) are gathered through a permutation (act-order layer whose rows were sorted by group
// at load time, gptq_act_order_repack).  gq_shift = log2(groupsize / 32), or -1 for a single group.
template <int NU, int NS, int DU, bool NORM, bool XPERM>
__global__ void __launch_bounds__(STRIPE_NW * 64) stripe_gemv_kernel(const half_t *__restrict__ x, const uint32_t *__restrict__ R,
                                                                     const uint32_t *__restrict__ tab, half_t *__restrict__ y, int K, int nrb, int G,
                                                                     int gq_shift, const half_t *__restrict__ bias, const half_t *__restrict__ nw,
                                                                     float eps, const int32_t *__restrict__ xperm, float *__restrict__ y32) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    constexpr int NW = STRIPE_NW, T = NW * 64;
    constexpr int XP = (NU + 3) / 4;   // 16-byte pieces of x per thread: K / 8 <= NU * NW * 16 = XP * T (rounded up)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t *xl = (half_t *)smem;                                                    // [K] (normalised) x
    float2 *xs4 = (float2 *)(smem + (size_t)K * 2);                                 // [K / 32] {sum x OFF_k, sum x}
    float2 *tabf = (float2 *)(smem + (size_t)K * 2 + (size_t)(K / 32) * 8);         // [NS][G][16] {s, -(z + 1) s}
    float *red = (float *)(tabf + (size_t)NS * G * 16);                             // [NW][NS][16], also the norm partials; XPERM: + raw x [K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stripe = blockIdx.x;
    const half2_t ones = {(half_t)1.f, (half_t)1.f}, c1024 = {(half_t)1024.f, (half_t)1024.f}, c64 = {(half_t)64.f, (half_t)64.f};
    const uint32_t MSK1 = sreg_const(0x00F000F0u), MAG1 = vreg_const(0x54005400u);
    const uint32_t MSK0 = sreg_const(0x000F000Fu), MAG0 = vreg_const(0x64006400u);
    const int npieces = K / 8, ntab = NS * G * 4;   // 16-byte pieces of x / of this stripe's table

    // ---- 1. x (+ norm weight) and the first table piece: requested BEFORE the weights (vector loads return in order) ----
    u32x4 xv[XP], nv[NORM ? XP : 1], pv[XPERM ? XP : 1][2], tv0;
#pragma unroll
    for (int i = 0; i < XP; i++) {
        const int idx = min(tid + i * T, npieces - 1);   // clamped, not branched: a branch in the load phase costs a vmcnt(0)
        xv[i] = *(const u32x4 *)(x + (size_t)idx * 8);    // natural order; XPERM gathers from LDS below (no dependent global loads)
        if constexpr (NORM) nv[i] = *(const u32x4 *)(nw + (size_t)idx * 8);
        if constexpr (XPERM) {
            pv[i][0] = *(const u32x4 *)(xperm + (size_t)idx * 8);
            pv[i][1] = *(const u32x4 *)(xperm + (size_t)idx * 8 + 4);
        }
    }
    const uint32_t *tsrc = tab + (size_t)stripe * NS * G * 16;
    tv0 = *(const u32x4 *)(tsrc + (size_t)min(tid, ntab - 1) * 4);
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. the first DU row blocks of this wave ----
    u32x4 w[NU][NS];
    const uint32_t *wbase = R + ((size_t)stripe * nrb * NS * 64 + lane) * 4;
    auto issue = [&](int u) {
        const int rb = min(wave + NW * u, nrb - 1);   // ragged tail: re-read the last block (its table entry is zeroed below)
#pragma unroll
        for (int s = 0; s < NS; s++) w[u][s] = __builtin_nontemporal_load((const u32x4 *)(wbase + ((size_t)rb * NS + s) * 256));
    };
#pragma unroll
    for (int u = 0; u < (DU < NU ? DU : NU); u++) issue(u);
    __builtin_amdgcn_sched_barrier(0);

    // ---- 3. stage x, the per-32-k sums and the table in LDS ----
    float rstd = 1.f;
    if constexpr (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < XP; i++) {
            if (tid + i * T < npieces) {
#pragma unroll
                for (int q = 0; q < 4; q++) ss = __builtin_amdgcn_fdot2(as_half2(xv[i][q]), as_half2(xv[i][q]), ss, false);
            }
        }
        ss = wave_sum_all(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; wv++) tot += red[wv];
        rstd = 1.0f / sqrtf(tot / (float)K + eps);
        __syncthreads();   // red is reused for the output partials
    }
#pragma unroll
    for (int i = 0; i < XP; i++) {
        u32x4 xn = xv[i];
        if constexpr (NORM) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const half2_t a = as_half2(xv[i][q]), b = as_half2(nv[i][q]);
                const half2_t r = {(half_t)((float)a[0] * rstd * (float)b[0]), (half_t)((float)a[1] * rstd * (float)b[1])};
                xn[q] = as_u32(r);
            }
        }
        xv[i] = xn;
    }
    if constexpr (XPERM) {
        // (normalised) x in natural order -> LDS -> gathered through the permutation: random access stays on chip
        half_t *xraw = (half_t *)(red + NW * NS * 16);
#pragma unroll
        for (int i = 0; i < XP; i++) {
            const int idx = tid + i * T;
            if (idx < npieces) *(u32x4 *)(xraw + (size_t)idx * 8) = xv[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < XP; i++) {
            half_t e[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                e[q] = xraw[pv[i][0][q]];
                e[4 + q] = xraw[pv[i][1][q]];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) xv[i][q] = as_u32(half2_t{e[2 * q], e[2 * q + 1]});
        }
    }
#pragma unroll
    for (int i = 0; i < XP; i++) {
        const int idx = tid + i * T;
        const u32x4 xn = xv[i];
        float s8 = 0.f, o8 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) s8 = __builtin_amdgcn_fdot2(as_half2(xn[q]), ones, s8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xn[0]), c1024, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xn[1]), c64, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xn[2]), c1024, o8, false);
        o8 = __builtin_amdgcn_fdot2(as_half2(xn[3]), c64, o8, false);
        s8 = quad_sum(s8);   // 4 adjacent pieces = 32 k
        o8 = quad_sum(o8);
        if (idx < npieces) {
            *(u32x4 *)(xl + (size_t)idx * 8) = xn;
            if ((idx & 3) == 0) xs4[idx >> 2] = float2{o8, s8};
        }
    }
    auto stage_tab = [&](int idx, const u32x4 tv) {
        if (idx < ntab) {
            float4_t a, b;
            const half2_t e0 = as_half2(tv[0]), e1 = as_half2(tv[1]), e2 = as_half2(tv[2]), e3 = as_half2(tv[3]);
            a[0] = (float)e0[0]; a[1] = -(float)e0[1] * (float)e0[0]; a[2] = (float)e1[0]; a[3] = -(float)e1[1] * (float)e1[0];
            b[0] = (float)e2[0]; b[1] = -(float)e2[1] * (float)e2[0]; b[2] = (float)e3[0]; b[3] = -(float)e3[1] * (float)e3[0];
            *(float4_t *)(tabf + (size_t)idx * 4) = a;
            *(float4_t *)(tabf + (size_t)idx * 4 + 2) = b;
        }
    };
    stage_tab(tid, tv0);
    if (ntab > T) {   // more than 512 table pieces (group size < 128 on a long K): the rest, blocking (rare)
        for (int idx = tid + T; idx < ntab; idx += T) stage_tab(idx, *(const u32x4 *)(tsrc + (size_t)idx * 4));
    }
    __syncthreads();

    // ---- 4. row blocks in arrival order; the next request goes out before the math of the current one ----
    const int rq = lane >> 4, col = lane & 15;
    float yv[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) yv[s] = 0.f;
#pragma unroll
    for (int u = 0; u < NU; u++) {
        if (u + DU < NU) {
            issue(u + DU);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool valid = wave + NW * u < nrb;
        const int rb = min(wave + NW * u, nrb - 1);
        const int qd = rb * 4 + rq;                              // 32-k block of this lane
        const int g = gq_shift >= 0 ? (qd >> gq_shift) : 0;
        const u32x4 *xp = (const u32x4 *)(xl + (size_t)qd * 32);
        u32x4 X[4];
#pragma unroll
        for (int j = 0; j < 4; j++) X[j] = xp[j];
        const float2 xs = xs4[qd];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t ww = w[u][s][j], hi = ww >> 8;
                const uint32_t t0 = (ww & MSK0) | MAG0, t1 = (ww & MSK1) | MAG1, t2 = (hi & MSK0) | MAG0, t3 = (hi & MSK1) | MAG1;
                const h4_t B1 = __builtin_bit_cast(h4_t, u32x2{t0, t1}), B2 = __builtin_bit_cast(h4_t, u32x2{t2, t3});
                const h4_t A1 = __builtin_bit_cast(h4_t, u32x2{X[j][0], X[j][1]}), A2 = __builtin_bit_cast(h4_t, u32x2{X[j][2], X[j][3]});
                acc = __builtin_amdgcn_mfma_f32_4x4x4f16(A1, B1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x4f16(A2, B2, acc, 0, 0, 0);
            }
            float2 e = tabf[((size_t)s * G + g) * 16 + col];
            if (!valid) e = float2{0.f, 0.f};
            yv[s] = fmaf(e.x, acc[0] - xs.x, yv[s]);
            yv[s] = fmaf(e.y, xs.y, yv[s]);
        }
    }

    // ---- 5. 4 row lanes -> 1 (permlane swaps), 8 waves -> 1 (LDS), epilogue, store ----
#pragma unroll
    for (int s = 0; s < NS; s++) yv[s] = fold_rows(yv[s]);
    if (lane < 16) {
#pragma unroll
        for (int s = 0; s < NS; s++) red[(wave * NS + s) * 16 + lane] = yv[s];
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
        if constexpr (NS == 2) v = a[0] * (1.0f / (1.0f + __expf(-a[0]))) * a[1];   // silu on the fp32 accumulator (fused_mlp.py:160-164)
        const int n = stripe * 16 + tid;
        if (y32) {           // K-shard of a row-sharded layer: the partial sums leave in fp32 (rounded once, after the all-reduce);
            y32[n] = a[0];   // gate and up separately ([2][N]): SiLU needs the complete sums
            if constexpr (NS == 2) y32[(size_t)gridDim.x * 16 + n] = a[1];
        } else {
            half_t h = (half_t)v;
            if (bias) h = (half_t)((float)h + (float)bias[n]);
            y[n] = h;
        }
    }
}

size_t stripe_lds_bytes(int K, int G, int NS, bool xperm) {
    return (size_t)K * 2 + (size_t)(K / 32) * 8 + (size_t)NS * G * 16 * 8 + (size_t)STRIPE_NW * NS * 16 * 4 + (xperm ? (size_t)K * 2 : 0);
}

template <int NU, int NS, bool NORM, bool XPERM>
int stripe_launch_one(const StripeParams &p, hipStream_t s) {
    constexpr int DU = NS == 2 ? 1 : (NU > 4 ? 3 : 2);
    auto kern = stripe_gemv_kernel<NU, NS, DU, NORM, XPERM>;
    const size_t lds = stripe_lds_bytes(p.K, p.G, NS, XPERM);
    if (lds > 160 * 1024 - 64) return GPTQ_E_VARIANT;
    if (lds > 48 * 1024) {
        static size_t configured = 0;   // per instantiation
        if (lds > configured) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            configured = lds;
        }
    }
    hipLaunchKernelGGL(kern, dim3(p.N / 16), dim3(STRIPE_NW * 64), lds, s, p.x, p.R, p.tab, p.y, p.K, p.K / 128, p.G, p.gq_shift, p.bias,
                       p.norm_w, p.norm_eps, p.xperm, p.y32);
    return (int)hipGetLastError();
}

template <int NS, bool NORM, bool XPERM>
int stripe_launch_nu(int nu, const StripeParams &p, hipStream_t s) {
    switch (nu) {
#define GPTQ_STRIPE_CASE(n) case n: return stripe_launch_one<n, NS, NORM, XPERM>(p, s);
        GPTQ_STRIPE_CASE(1) GPTQ_STRIPE_CASE(2) GPTQ_STRIPE_CASE(3) GPTQ_STRIPE_CASE(4) GPTQ_STRIPE_CASE(5) GPTQ_STRIPE_CASE(6)
        GPTQ_STRIPE_CASE(7) GPTQ_STRIPE_CASE(8) GPTQ_STRIPE_CASE(9) GPTQ_STRIPE_CASE(10) GPTQ_STRIPE_CASE(11) GPTQ_STRIPE_CASE(12)
        GPTQ_STRIPE_CASE(13) GPTQ_STRIPE_CASE(14) GPTQ_STRIPE_CASE(15) GPTQ_STRIPE_CASE(16) GPTQ_STRIPE_CASE(17) GPTQ_STRIPE_CASE(18)
        GPTQ_STRIPE_CASE(19) GPTQ_STRIPE_CASE(20) GPTQ_STRIPE_CASE(21) GPTQ_STRIPE_CASE(22) GPTQ_STRIPE_CASE(23) GPTQ_STRIPE_CASE(24)
#undef GPTQ_STRIPE_CASE
    }
    return GPTQ_E_VARIANT;
}

}  // namespace

// groupsize is the effective one (K for the reference's -1).  Returns log2(groupsize / 32), -1 for one group, -2 if ineligible.
int stripe_gq_shift(int K, int N, int bits, int groupsize) {
    if (bits != 4 || K <= 0 || N <= 0 || K % 128 != 0 || N % 16 != 0) return -2;
    if ((K / 128 + STRIPE_NW - 1) / STRIPE_NW > STRIPE_MAX_NU) return -2;
    if (groupsize >= K) return -1;
    if (groupsize < 32 || groupsize % 32 != 0 || K % groupsize != 0) return -2;
    const int q = groupsize / 32;
    for (int sft = 0; sft < 16; sft++)
        if ((1 << sft) == q) return sft;
    return -2;
}

size_t stripe_tab_offset(int K, int N, int nsets) { return (size_t)(K / 8) * N * nsets * 4; }

size_t stripe_total_bytes(int K, int N, int bits, int groupsize, int nsets) {
    if (stripe_gq_shift(K, N, bits, groupsize) == -2 || nsets < 1 || nsets > 2) return 0;
    const int G = groupsize >= K ? 1 : K / groupsize;
    return stripe_tab_offset(K, N, nsets) + (size_t)nsets * G * N * 4;
}

int stripe_repack_launch(const uint32_t *qw0, const half_t *sc0, const int32_t *qz0, const uint32_t *qw1, const half_t *sc1, const int32_t *qz1,
                         void *out, int K, int N, int groupsize, hipStream_t s) {
    const int NS = qw1 ? 2 : 1;
    const int G = groupsize >= K ? 1 : K / groupsize;
    uint32_t *R = (uint32_t *)out;
    uint32_t *tab = (uint32_t *)((char *)out + stripe_tab_offset(K, N, NS));
    hipLaunchKernelGGL(stripe_repack_kernel, dim3(2048), dim3(256), 0, s, qw0, qw1, R, N, K / 128, NS);
    hipLaunchKernelGGL(stripe_table_kernel, dim3(512), dim3(256), 0, s, sc0, qz0, sc1, qz1, tab, N, G, NS);
    return (int)hipGetLastError();
}

int stripe_gemv_dispatch(const StripeParams &p, hipStream_t s) {
    const int nu = (p.K / 128 + STRIPE_NW - 1) / STRIPE_NW;
    const bool norm = p.norm_w != nullptr, perm = p.xperm != nullptr;
    if (p.NS == 2) {
        if (norm) return perm ? stripe_launch_nu<2, true, true>(nu, p, s) : stripe_launch_nu<2, true, false>(nu, p, s);
        return perm ? stripe_launch_nu<2, false, true>(nu, p, s) : stripe_launch_nu<2, false, false>(nu, p, s);
    }
    if (norm) return perm ? stripe_launch_nu<1, true, true>(nu, p, s) : stripe_launch_nu<1, true, false>(nu, p, s);
    return perm ? stripe_launch_nu<1, false, true>(nu, p, s) : stripe_launch_nu<1, false, false>(nu, p, s);
}

}  // namespace gptq
