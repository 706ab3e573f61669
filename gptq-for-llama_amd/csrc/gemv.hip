// gemv.hip -- batch-1 dequant-matvec for gfx950: the decode hot path behind QuantLinear.forward
// (reference quant/quant_linear.py:72-137, 263-269, 373-377) and the fused gate/up + SiLU*mul of
// QuantLlamaMLP (reference quant/fused_mlp.py:84-168).
//
// "rowwave" design (HBM-bound; measurements in DESIGN.md and tools/gemvlab.hip):
//  * a wave reads whole 1-KiB row segments of qweight: 64 lanes x one global_load_dwordx4 =
//    256 consecutive columns of ONE packed row (fully coalesced, >= 256-byte DRAM bursts);
//    every lane owns 4 columns, so there is no k-reduction inside the wave at all;
//  * because all lanes of a wave work on the same k rows, x is wave-uniform: it is fetched with
//    s_load through the scalar cache into SGPRs and fed to v_dot2_f32_f16 as a scalar operand --
//    no LDS staging, no barrier in front of the math;
//  * all U rows (x 1 or 2 weight sets) of a wave's chunk are requested before any arithmetic
//    (8-16 dwordx4 in flight per lane, 32-64 KiB per workgroup) and consumed in arrival order;
//  * a word (KPW k of one column) is expanded with the fp16 magic-exponent trick to half2 pairs
//    {OFF+q_i, OFF+q_{i+NP}} (shift + v_and_or_b32) and multiplied with v_dot2_f32_f16 in fp32;
//    the offset and the zero point are removed once per (group, column):
//        y += s * (sum_k x_k (OFF + q_k) - (OFF + z) * sum_k x_k);
//  * the 4 waves of a workgroup take 4 consecutive U-row blocks (one LDS reduce), K is split over
//    S workgroups per 256-column tile and combined with ONE returning 64-bit fixed-point atomic
//    per output (gptq_device.h): bit-reproducible, no second pass.
//  * act-order / odd group sizes / 3-bit go through gemv_generic_kernel, which keeps a per-tile
//    {scale, zero} table in LDS indexed by g_idx[k].
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

// ---------------------------------------------------------------------------------------
// fast path: M == 1, trivial g_idx, power-of-two group of >= U packed rows (or one group),
// bits in {2,4,8}.  Kernel arguments are plain scalars so that the first 16 dwords are preloaded
// into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=16): the weight loads are
// issued a few hundred cycles after the wave starts.
// ---------------------------------------------------------------------------------------
// NORM: x is RMS-normalised on the fly -- xn = fp16(x * rsqrt(mean(x^2) + eps) * nw), the arithmetic of
// rms_norm_fwd_fused (reference quant/triton_norm.py:22-39) -- so that [RMSNorm -> QuantLinear] of a
// decoder layer is ONE launch.  Every wave computes sum(x^2) redundantly from L2 while its weight
// loads are in flight (no barrier) and normalises only the U*KPW values it needs (through LDS).
// XPERM: x is gathered as x[xperm[k]] (an act-order layer whose rows were sorted by group at load
// time, so that it is a trivial-g_idx layer of the permuted input).  Lane l fetches perm[k] and then
// x[perm[k]] for the wave's own U*KPW values BEFORE the weights are requested (one unloaded L2 round
// trip at kernel start; behind the weights the gather would only land after all of them) and the
// values are broadcast through LDS like the NORM path.
template <int BITS, int U, bool FUSED2, bool DBG, bool NORM = false, bool XPERM = false>
__global__ void __launch_bounds__(256) gemv_rowwave_kernel(
    const uint32_t *__restrict__ qw0, const half_t *__restrict__ x, const half_t *__restrict__ sc0,
    const int32_t *__restrict__ qz0, int N, int rows, int S, int gshift, const uint32_t *__restrict__ qw1,
    const half_t *__restrict__ sc1, const int32_t *__restrict__ qz1, half_t *__restrict__ y, u64_t *__restrict__ ws,
    const half_t *__restrict__ bias, u64_t *__restrict__ dbg, const half_t *__restrict__ nw, float eps,
    const int32_t *__restrict__ xperm) {
    using UP = Unpack<BITS>;
    constexpr int KPW = UP::KPW, NP = UP::NP;
    constexpr int XW = KPW / 2;  // dwords of x per packed row
    constexpr int NS = FUSED2 ? 2 : 1;
    typedef uint32_t xrow_t __attribute__((ext_vector_type(XW)));
    __shared__ float red[NS][4][256];
    __shared__ __attribute__((aligned(16))) half_t xn[(NORM || XPERM) ? 4 : 1][(NORM || XPERM) ? U * KPW : 8];
    float rstd = 0.f;
    bool have_rstd = false;
    // NORM: this lane's share of x for sum(x^2) is requested BEFORE the weights (vector loads
    // return in order: behind 8-16 weight rows it would arrive last) and reduced while they stream.
    half8_t xsq[NORM ? 8 : 1];
    float ss_tail = 0.f;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u64_t st[8];
    if constexpr (DBG) {
        st[0] = stamp_realtime();
        st[1] = stamp_cycles(0);
    }
    const uint32_t tile = blockIdx.x, slice = blockIdx.y;  // grid = (256-column tiles, K slices)
    const uint32_t n0 = tile * 256 + lane * 4;
    const uint32_t nc = n0 < (uint32_t)N ? n0 : 0;  // ragged N: idle lanes read column 0, results are dropped
    const uint32_t nchunk = ((uint32_t)rows + 4 * U - 1) / (4 * U);
    const uint32_t *qw[2] = {qw0, qw1};
    const half_t *sc[2] = {sc0, sc1};
    const int32_t *qz[2] = {qz0, qz1};
    const half2_t ones = {(half_t)1.0f, (half_t)1.0f};
    const uint32_t MSK = sreg_const(UP::MSK_C), MAG = vreg_const(UP::MAG_C);

    float yv[NS][4];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) yv[s][j] = 0.f;
    if constexpr (NORM) {
        const int nv = rows * KPW / 8;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = lane + 64 * j;
            xsq[j] = i < nv ? *(const half8_t *)(x + (size_t)i * 8) : (half8_t)(half_t)0;
        }
        for (int i = 512 + lane; i < nv; i += 64) {  // K > 4096: the rest, blocking (rare)
            const half8_t v = *(const half8_t *)(x + (size_t)i * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) ss_tail += (float)v[e] * (float)v[e];
        }
    }

    for (uint32_t c = slice; c < nchunk; c += (uint32_t)S) {
        const uint32_t row = c * (4 * U) + wave * U;  // first packed row of this wave's block (uniform)
        if (row >= (uint32_t)rows) continue;          // rows % U == 0: a block is all in or all out
        u32x4 w[NS][U];
        half4_t s4[NS];
        uint32_t zw[NS];
        const uint32_t g = gshift >= 0 ? (row >> gshift) : 0u;
        // NORM: the raw x / norm-weight values of this wave's own rows, requested ahead of the weights
        constexpr int XE = (NORM || XPERM) ? (U * KPW + 63) / 64 : 1;
        half_t xe[XE], nwe[XE];
        if constexpr (XPERM) {
            int32_t pk[XE];
#pragma unroll
            for (int i = 0; i < XE; i++) {
                const int e = lane + 64 * i;
                pk[i] = xperm[(size_t)row * KPW + (e < U * KPW ? e : 0)];
            }
#pragma unroll
            for (int i = 0; i < XE; i++) {                   // dependent gather, still ahead of the weights
                xe[i] = x[pk[i]];
                if constexpr (NORM) nwe[i] = nw[pk[i]];       // NORM + XPERM: normalise the gathered values (sum(x^2) is order-free)
            }
            __builtin_amdgcn_sched_barrier(0);               // (the scheduler would sink it behind them)
        }
        if constexpr (NORM && !XPERM) {
#pragma unroll
            for (int i = 0; i < XE; i++) {
                const int e = lane + 64 * i;
                const size_t k = (size_t)row * KPW + (e < U * KPW ? e : 0);
                xe[i] = x[k];
                nwe[i] = nw[k];
            }
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
#pragma unroll
            for (int u = 0; u < U; u++)
                w[s][u] = __builtin_nontemporal_load((const u32x4 *)(qw[s] + (size_t)(row + u) * (uint32_t)N + nc));
            s4[s] = *(const half4_t *)(sc[s] + (size_t)g * (uint32_t)N + nc);
            zw[s] = (uint32_t)qz[s][(size_t)g * ((uint32_t)N / KPW) + nc / KPW];
        }
        xrow_t xr[U];
        if constexpr (XPERM) {
            // filled from LDS below
        } else if constexpr (!NORM) {
            const xrow_t *xq = (const xrow_t *)x + row;  // wave-uniform: scalar loads
#pragma unroll
            for (int u = 0; u < U; u++) xr[u] = xq[u];
        }
        __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is in flight before any math
        if constexpr (DBG) st[2] = stamp_cycles(0);
        if constexpr (XPERM && !NORM) {
#pragma unroll
            for (int i = 0; i < XE; i++) {
                const int e = lane + 64 * i;
                if (e < U * KPW) xn[wave][e] = xe[i];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < U; u++) xr[u] = *(const xrow_t *)&xn[wave][u * KPW];
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (NORM) {
            const int K = rows * KPW;
            if (!have_rstd) {  // once per wave: sum(x^2) over all of K (the loads were issued first)
                // opaque use INSIDE the loop, after the weight loads were issued: keeps the compiler from
                // hoisting this reduction (and its vmcnt wait) in front of them
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    u32x4 pin = __builtin_bit_cast(u32x4, xsq[j]);
                    asm volatile("" : "+v"(pin));
                    xsq[j] = __builtin_bit_cast(half8_t, pin);
                }
                float ss = ss_tail;
#pragma unroll
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int e = 0; e < 8; e++) ss += (float)xsq[j][e] * (float)xsq[j][e];
                ss = wave_sum_xor(ss, 1);
                rstd = 1.0f / sqrtf(ss / (float)K + eps);
                have_rstd = true;
            }
#pragma unroll
            for (int i = 0; i < XE; i++) {
                const int e = lane + 64 * i;
                if (e < U * KPW) xn[wave][e] = (half_t)((float)xe[i] * rstd * (float)nwe[i]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < U; u++) xr[u] = *(const xrow_t *)&xn[wave][u * KPW];  // same address in every lane: broadcast
            __builtin_amdgcn_wave_barrier();
        }

        float acc[NS][4];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[s][j] = 0.f;
        float xs = 0.f;
#pragma unroll
        for (int u = 0; u < U; u++) {
            // x pairs in the order the unpack produces fields: pair q = (x[q], x[q + NP])
            half2_t X[NP];
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const uint32_t a = xr[u][q / 2], b = xr[u][(q + NP) / 2];
                X[q] = as_half2((q & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16)));
                xs = __builtin_amdgcn_fdot2(X[q], ones, xs, false);
            }
            if constexpr (DBG) {
                if (u == 0) {
                    st[3] = stamp_cycles(xr[0][0]);
                    st[4] = stamp_cycles(w[0][0][0]);
                }
                if (u == U - 1) st[5] = stamp_cycles(w[NS - 1][U - 1][0]);
            }
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    half2_t t[NP];
                    UP::pairs_rc(w[s][u][j], t, MSK, MAG);
#pragma unroll
                    for (int q = 0; q < NP; q++) acc[s][j] = __builtin_amdgcn_fdot2(t[q], X[q], acc[s][j], false);
                }
        }
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float zf = (float)(((zw[s] >> (BITS * ((nc + j) % KPW))) & ((1u << BITS) - 1u)) + 1u) + UP::OFF;
                yv[s][j] += (float)s4[s][j] * (acc[s][j] - zf * xs);
            }
    }
    if constexpr (DBG) st[6] = stamp_cycles(__builtin_bit_cast(uint32_t, yv[0][0]));

    // ---- 4 waves -> one partial per column (LDS), K slices -> one atomic round trip ------------
#pragma unroll
    for (int s = 0; s < NS; s++) *(float4_t *)&red[s][wave][4 * lane] = float4_t{yv[s][0], yv[s][1], yv[s][2], yv[s][3]};
    __syncthreads();
    const int t = threadIdx.x;
    const uint32_t n = tile * 256 + t;
    float t0 = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t], t1 = 0.f;
    if constexpr (FUSED2) t1 = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
    if (n < (uint32_t)N) {
        bool mine = true;
        if (S > 1) {
            if constexpr (FUSED2) mine = splitk_add2(ws + 2 * (size_t)n, t0, t1, S, t0, t1);
            else mine = splitk_add1(ws + n, t0, S, t0);
        }
        if (mine) {
            float v = t0;
            if constexpr (FUSED2) v = t0 * (1.0f / (1.0f + __expf(-t0))) * t1;  // silu on the fp32 accumulator
            half_t h = (half_t)v;
            if (bias) h = (half_t)((float)h + (float)bias[n]);
            y[n] = h;
        }
    }
    if constexpr (DBG) {
        st[7] = stamp_cycles(__builtin_bit_cast(uint32_t, t0));
        const u64_t te = stamp_realtime();
        if (dbg && lane == 0) {
            u64_t *d = dbg + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 10;
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = st[i];
            d[8] = te;
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[9] = xcc;
        }
    }
}


// ---------------------------------------------------------------------------------------
// 3-bit rowwave (EXTENSION: the reference rejects bits == 3, quant_linear.py:308-309).
// A 32-k block of a column is a dense little-endian 96-bit stream over 3 consecutive packed rows;
// a wave takes UB blocks (3*UB row loads of 1 KiB), x of a block is 16 SGPR dwords, and pair i of
// the stream (fields 2i, 2i+1 = 6 bits at bit 6i) lines up with x dword i:
//     t = bfe(stream, 6i, 6);  half2 = ((t << 13 | t) & 0x00070007) | 0x64006400 = {1024+q0, 1024+q1}
// The 1024 offset is removed per BLOCK (a - 1024 * sum x) before the group accumulator so that a
// single group over all of K (3-bit "no-group" checkpoints) keeps fp32 accuracy.
// ---------------------------------------------------------------------------------------
template <int UB, bool FUSED2>
__global__ void __launch_bounds__(256) gemv_rowwave3_kernel(const uint32_t *__restrict__ qw0, const half_t *__restrict__ x,
                                                            const half_t *__restrict__ sc0, const int32_t *__restrict__ qz0, int N,
                                                            int nblocks, int S, int gshift, const uint32_t *__restrict__ qw1,
                                                            const half_t *__restrict__ sc1, const int32_t *__restrict__ qz1,
                                                            half_t *__restrict__ y, u64_t *__restrict__ ws, const half_t *__restrict__ bias) {
    typedef uint32_t x16_t __attribute__((ext_vector_type(16)));
    constexpr int NS = FUSED2 ? 2 : 1;
    __shared__ float red[NS][4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x, slice = blockIdx.y;
    const uint32_t n0 = tile * 256 + lane * 4;
    const uint32_t nc = n0 < (uint32_t)N ? n0 : 0;
    const uint32_t nchunk = ((uint32_t)nblocks + 4 * UB - 1) / (4 * UB);
    const half2_t ones = {(half_t)1.0f, (half_t)1.0f};
    const uint32_t MSK = sreg_const(0x00070007u), MAG = vreg_const(0x64006400u);
    const int ldz = N / 32 * 3;
    const uint32_t *qw[2] = {qw0, qw1};
    const half_t *sc[2] = {sc0, sc1};
    const int32_t *qz[2] = {qz0, qz1};

    float yv[NS][4];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) yv[s][j] = 0.f;
    for (uint32_t c = slice; c < nchunk; c += (uint32_t)S) {
        const uint32_t blk0 = c * (4 * UB) + wave * UB;   // first 32-k block of this wave (uniform)
        if (blk0 >= (uint32_t)nblocks) continue;           // nblocks % UB == 0
        const uint32_t g = gshift >= 0 ? (blk0 >> gshift) : 0u;
        u32x4 w[NS][UB][3];
        half4_t s4[NS];
        uint32_t z[NS][3];
#pragma unroll
        for (int s = 0; s < NS; s++) {
#pragma unroll
            for (int b = 0; b < UB; b++)
#pragma unroll
                for (int r = 0; r < 3; r++)
                    w[s][b][r] = __builtin_nontemporal_load((const u32x4 *)(qw[s] + (size_t)((blk0 + b) * 3 + r) * (uint32_t)N + nc));
            s4[s] = *(const half4_t *)(sc[s] + (size_t)g * (uint32_t)N + nc);
            const uint32_t *zrow = (const uint32_t *)qz[s] + (size_t)g * ldz + 3 * (nc >> 5);
            z[s][0] = zrow[0]; z[s][1] = zrow[1]; z[s][2] = zrow[2];
        }
        const x16_t *xq = (const x16_t *)x + blk0;
        x16_t xr[UB];
#pragma unroll
        for (int b = 0; b < UB; b++) xr[b] = xq[b];
        __builtin_amdgcn_sched_barrier(0);

        float gacc[NS][4];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) gacc[s][j] = 0.f;
        float xs_g = 0.f;
#pragma unroll
        for (int b = 0; b < UB; b++) {
            float xs = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) xs = __builtin_amdgcn_fdot2(as_half2(xr[b][i]), ones, xs, false);
            xs_g += xs;
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t a0 = w[s][b][0][j], a1 = w[s][b][1][j], a2 = w[s][b][2][j];
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int bit = 6 * i;
                        uint32_t t;
                        if (bit + 6 <= 32) t = __builtin_amdgcn_ubfe(a0, bit, 6);
                        else if (bit < 32) t = __builtin_amdgcn_alignbit(a1, a0, bit) & 0x3Fu;
                        else if (bit + 6 <= 64) t = __builtin_amdgcn_ubfe(a1, bit - 32, 6);
                        else if (bit < 64) t = __builtin_amdgcn_alignbit(a2, a1, bit - 32) & 0x3Fu;
                        else t = __builtin_amdgcn_ubfe(a2, bit - 64, 6);
                        const uint32_t h = (((t << 13) | t) & MSK) | MAG;
                        a = __builtin_amdgcn_fdot2(as_half2(h), as_half2(xr[b][i]), a, false);
                    }
                    gacc[s][j] += a - 1024.0f * xs;
                }
        }
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int bit = 3 * ((nc + j) & 31), wi = bit >> 5, o = bit & 31;
                const uint64_t lo = wi == 0 ? z[s][0] : (wi == 1 ? z[s][1] : z[s][2]);
                const uint64_t hi = wi == 0 ? z[s][1] : (wi == 1 ? z[s][2] : 0u);
                const float zf = (float)((uint32_t)(((lo | (hi << 32)) >> o) & 7u) + 1u);
                yv[s][j] += (float)s4[s][j] * (gacc[s][j] - zf * xs_g);
            }
    }
#pragma unroll
    for (int s = 0; s < NS; s++) *(float4_t *)&red[s][wave][4 * lane] = float4_t{yv[s][0], yv[s][1], yv[s][2], yv[s][3]};
    __syncthreads();
    const int t = threadIdx.x;
    const uint32_t n = tile * 256 + t;
    float t0 = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t], t1 = 0.f;
    if constexpr (FUSED2) t1 = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
    if (n < (uint32_t)N) {
        bool mine = true;
        if (S > 1) {
            if constexpr (FUSED2) mine = splitk_add2(ws + 2 * (size_t)n, t0, t1, S, t0, t1);
            else mine = splitk_add1(ws + n, t0, S, t0);
        }
        if (mine) {
            float v = t0;
            if constexpr (FUSED2) v = t0 * (1.0f / (1.0f + __expf(-t0))) * t1;  // silu on the fp32 accumulator
            half_t h = (half_t)v;
            if (bias) h = (half_t)((float)h + (float)bias[n]);
            y[n] = h;
        }
    }
}

// ---------------------------------------------------------------------------------------
// generic path: any g_idx (act-order), any group size, bits in {2,3,4,8}.  A {scale, zero}
// table for the tile's columns lives in LDS ([G][TILE] of half2{s, z}) and is indexed by
// g_idx[k]; weights are dequantised to fp16 exactly like the reference ((q - z) exact in
// fp16, times the fp16 scale, one fp16 rounding) and accumulated in fp32.
// ---------------------------------------------------------------------------------------
template <int BITS, int NL, int WAVES, int MR, bool FUSED2>
__global__ void __launch_bounds__(WAVES * 64) gemv_generic_kernel(const GemvParams p) {
    constexpr int CH = BITS;  // packed rows per 32-k block (32*BITS/32)
    constexpr int KLW = 64 / NL, KL = WAVES * KLW, T = WAVES * 64;
    constexpr int NS = FUSED2 ? 2 : 1;
    constexpr int TILE = 4 * NL;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = lane % NL, kl = wave * KLW + lane / NL;
    const int n0 = tile * TILE + 4 * cg;
    const bool active = n0 < p.N;
    const int N = p.N, K = p.K, G = p.G;
    const int ldz = N / 32 * BITS;

    // LDS carve: x [MR][K] half | gidx [NS][K] u16 | table [NS][G][TILE] half2
    half_t *lx = (half_t *)smem;
    uint16_t *lg = (uint16_t *)(lx + (size_t)MR * K);
    half2_t *tab = (half2_t *)(smem + (((size_t)MR * K * 2 + (size_t)NS * K * 2 + 15) & ~(size_t)15));

    for (int idx = tid; idx < MR * K; idx += T) {
        const int m = idx / K, k = idx % K;
        lx[idx] = (m < p.M) ? p.x[(size_t)m * p.ldx + k] : (half_t)0;
    }
    for (int idx = tid; idx < NS * K; idx += T) {
        const int s = idx / K, k = idx % K;
        int g = p.gi[s] ? p.gi[s][k] : k / p.groupsize;
        g = (g < 0 || g >= G) ? 0 : g;
        lg[idx] = (uint16_t)g;
    }
    for (int idx = tid; idx < NS * G * TILE; idx += T) {
        const int s = idx / (G * TILE), g = (idx / TILE) % G, j = idx % TILE;
        const int n = tile * TILE + j;
        half2_t e = {(half_t)0, (half_t)0};
        if (n < N) {
            e[0] = p.sc[s][(size_t)g * N + n];
            e[1] = (half_t)(float)zero_of<BITS>(p.qz[s] + (size_t)g * ldz, n);
        }
        tab[idx] = e;
    }
    __syncthreads();

    float y[NS][MR][4];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int m = 0; m < MR; m++)
#pragma unroll
            for (int j = 0; j < 4; j++) y[s][m][j] = 0.f;

    const int nblk = K / 32;
    for (int c = kl; c < nblk; c += KL) {
        if (!active) continue;
        u32x4 w[NS][CH];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int i = 0; i < CH; i++)
                w[s][i] = __builtin_nontemporal_load((const u32x4 *)(p.qw[s] + ((size_t)c * CH + i) * N + n0));
#pragma unroll
        for (int jj = 0; jj < 32; jj++) {
            const int k = c * 32 + jj;
            float xv[MR];
#pragma unroll
            for (int m = 0; m < MR; m++) xv[m] = (float)lx[(size_t)m * K + k];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int g = lg[(size_t)s * K + k];
                const half2_t *e = tab + ((size_t)s * G + g) * TILE + 4 * cg;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t col[CH];
#pragma unroll
                    for (int i = 0; i < CH; i++) col[i] = w[s][i][j];
                    const int q = field_of_block<BITS>(col, jj);
                    const half2_t sz = e[j];
                    // reference numerics: fp16(q - z) * fp16 scale -> fp16 (quant_linear.py:128)
                    const half_t wq = (half_t)((half_t)(float)q - sz[1]) * sz[0];
#pragma unroll
                    for (int m = 0; m < MR; m++) y[s][m][j] += xv[m] * (float)wq;
                }
            }
        }
    }

#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int m = 0; m < MR; m++)
#pragma unroll
            for (int j = 0; j < 4; j++) y[s][m][j] = wave_sum_xor(y[s][m][j], NL);

    __syncthreads();
    float *red = (float *)smem;
    if (lane < NL) {
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int m = 0; m < MR; m++) {
                float4_t v = {y[s][m][0], y[s][m][1], y[s][m][2], y[s][m][3]};
                *(float4_t *)(red + (((size_t)wave * NS + s) * MR + m) * TILE + 4 * lane) = v;
            }
    }
    __syncthreads();
    constexpr int NOUT = MR * TILE;
    for (int e = tid; e < NOUT; e += T) {
        const int m = e / TILE, n = tile * TILE + e % TILE;
        if (m < p.M && n < N) {
            float a[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                a[s] = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; w++) a[s] += red[((size_t)w * NS + s) * NOUT + e];
            }
            float v = a[0];
            if constexpr (FUSED2) v = a[0] * (1.0f / (1.0f + __expf(-a[0]))) * a[1];
            if constexpr (!FUSED2) {
                if (p.y32) {   // a row shard's partial product: rounded once, by whoever sums the shards (tensor parallel act-order layers)
                    p.y32[(size_t)m * p.ldy + n] = v;
                    continue;
                }
            }
            half_t h = (half_t)v;
            if (p.bias) h = (half_t)((float)h + (float)p.bias[n]);
            p.y[(size_t)m * p.ldy + n] = h;
        }
    }
}

// ---------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------
template <int BITS, int U, bool FUSED2>
static int launch_rowwave_norm(const GemvParams &p, hipStream_t stream) {
    constexpr int KPW = 32 / BITS;
    const int rows = p.K / KPW;
    dim3 grid((p.N + 255) / 256, p.split_k), block(256);
    hipLaunchKernelGGL((gemv_rowwave_kernel<BITS, U, FUSED2, false, true>), grid, block, 0, stream, p.qw[0], p.x, p.sc[0], p.qz[0], p.N,
                       rows, p.split_k, p.upg_shift, p.qw[1], p.sc[1], p.qz[1], p.y, p.ws, p.bias, (u64_t *)nullptr, p.norm_w, p.norm_eps, (const int32_t *)nullptr);
    return (int)hipGetLastError();
}

template <int BITS, int U, bool FUSED2>
static int launch_rowwave_xperm(const GemvParams &p, hipStream_t stream) {
    constexpr int KPW = 32 / BITS;
    const int rows = p.K / KPW;
    dim3 grid((p.N + 255) / 256, p.split_k), block(256);
    hipLaunchKernelGGL((gemv_rowwave_kernel<BITS, U, FUSED2, false, false, true>), grid, block, 0, stream, p.qw[0], p.x, p.sc[0], p.qz[0],
                       p.N, rows, p.split_k, p.upg_shift, p.qw[1], p.sc[1], p.qz[1], p.y, p.ws, p.bias, (u64_t *)nullptr,
                       (const half_t *)nullptr, 0.f, p.xperm);
    return (int)hipGetLastError();
}

template <int BITS, int U, bool FUSED2>
static int launch_rowwave_norm_xperm(const GemvParams &p, hipStream_t stream) {
    constexpr int KPW = 32 / BITS;
    const int rows = p.K / KPW;
    dim3 grid((p.N + 255) / 256, p.split_k), block(256);
    hipLaunchKernelGGL((gemv_rowwave_kernel<BITS, U, FUSED2, false, true, true>), grid, block, 0, stream, p.qw[0], p.x, p.sc[0], p.qz[0],
                       p.N, rows, p.split_k, p.upg_shift, p.qw[1], p.sc[1], p.qz[1], p.y, p.ws, p.bias, (u64_t *)nullptr, p.norm_w,
                       p.norm_eps, p.xperm);
    return (int)hipGetLastError();
}

template <int BITS, int U, bool FUSED2>
static int launch_rowwave(const GemvParams &p, hipStream_t stream) {
    if (p.xperm) {
        if constexpr (BITS == 4) {
            if (p.norm_w) {
                if constexpr (U == 8) return launch_rowwave_norm_xperm<BITS, U, FUSED2>(p, stream);
                else return GPTQ_E_VARIANT;
            }
            return launch_rowwave_xperm<BITS, U, FUSED2>(p, stream);
        } else {
            return GPTQ_E_VARIANT;
        }
    }
    if (p.norm_w) {
        if constexpr (BITS == 4 && U == 8) return launch_rowwave_norm<BITS, U, FUSED2>(p, stream);
        else return GPTQ_E_VARIANT;
    }
    constexpr int KPW = 32 / BITS;
    const int rows = p.K / KPW;
    const int ntile = (p.N + 255) / 256;
    dim3 grid(ntile, p.split_k), block(256);
    const int gshift = p.upg_shift;  // log2(packed rows per group), or -1 for a single group
    if (p.dbg) {
        if constexpr (BITS == 4 && U == 8) {
            hipLaunchKernelGGL((gemv_rowwave_kernel<BITS, U, FUSED2, true>), grid, block, 0, stream, p.qw[0], p.x, p.sc[0], p.qz[0], p.N,
                               rows, p.split_k, gshift, p.qw[1], p.sc[1], p.qz[1], p.y, p.ws, p.bias, p.dbg, (const half_t *)nullptr, 0.f, (const int32_t *)nullptr);
            return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL((gemv_rowwave_kernel<BITS, U, FUSED2, false>), grid, block, 0, stream, p.qw[0], p.x, p.sc[0], p.qz[0], p.N, rows,
                       p.split_k, gshift, p.qw[1], p.sc[1], p.qz[1], p.y, p.ws, p.bias, (u64_t *)nullptr, (const half_t *)nullptr, 0.f, (const int32_t *)nullptr);
    return (int)hipGetLastError();
}

template <int BITS, int NL, int WAVES, int MR, bool FUSED2>
static int launch_generic(const GemvParams &p, hipStream_t stream) {
    constexpr int NS = FUSED2 ? 2 : 1;
    constexpr int TILE = 4 * NL;
    size_t a = (((size_t)MR * p.K * 2 + (size_t)NS * p.K * 2 + 15) & ~(size_t)15) + (size_t)NS * p.G * TILE * 4;
    const size_t red_bytes = (size_t)WAVES * NS * MR * TILE * 4;
    const size_t lds = ((a > red_bytes ? a : red_bytes) + 15) & ~(size_t)15;
    if (lds > 160 * 1024 - 64) return GPTQ_E_SHAPE;
    auto kern = gemv_generic_kernel<BITS, NL, WAVES, MR, FUSED2>;
    static LdsOptIn opt_in;   // per instantiation; per device inside
    if (int rc = opt_in.ensure((const void *)kern, lds)) return rc;
    dim3 grid(p.ntiles), block(WAVES * 64);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

template <int BITS, bool FUSED2>
static int launch_rowwave_u(int u, const GemvParams &p, hipStream_t s) {
    switch (u) {
        case 8: return launch_rowwave<BITS, 8, FUSED2>(p, s);
        case 4: return launch_rowwave<BITS, 4, FUSED2>(p, s);
        case 2: return launch_rowwave<BITS, 2, FUSED2>(p, s);
    }
    return GPTQ_E_VARIANT;
}

// ---------------------------------------------------------------------------------------
// rowwave for 2 <= M <= 4 (small decode batches): the same weight stream, each unpacked word multiplied with MR
// rows of x.  The unpack (7 of the 11 VALU per word at M = 1) is shared, so MR = 2 costs 15 and MR = 4 costs 23
// VALU per word: still under the stream at M = 2, about level with it at M = 4 -- and one launch instead of M
// launches (M = 2) or the MFMA stream kernel's three-round-trip combine (M = 3, 4).  4-bit, trivial g_idx.
// x rows are MR scalar streams (x + m * ldx); rows past M re-read row M-1 and are not written.
// Combine words: ws[m * N + n].
// ---------------------------------------------------------------------------------------
template <int U, bool FUSED2, int MR>
__global__ void __launch_bounds__(256) gemv_rowwave_mr_kernel(const uint32_t *__restrict__ qw0, const half_t *__restrict__ x, int ldx,
                                                              const half_t *__restrict__ sc0, const int32_t *__restrict__ qz0, int N, int rows,
                                                              int S, int gshift, const uint32_t *__restrict__ qw1,
                                                              const half_t *__restrict__ sc1, const int32_t *__restrict__ qz1,
                                                              half_t *__restrict__ y, int ldy, u64_t *__restrict__ ws,
                                                              const half_t *__restrict__ bias, int M) {
    constexpr int BITS = 4;
    using UP = Unpack<BITS>;
    constexpr int KPW = UP::KPW, NP = UP::NP, XW = KPW / 2;
    constexpr int NS = FUSED2 ? 2 : 1;
    typedef uint32_t xrow_t __attribute__((ext_vector_type(XW)));
    __shared__ float red[MR][NS][4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x, slice = blockIdx.y;
    const uint32_t n0 = tile * 256 + lane * 4;
    const uint32_t nc = n0 < (uint32_t)N ? n0 : 0;
    const uint32_t nchunk = ((uint32_t)rows + 4 * U - 1) / (4 * U);
    const uint32_t *qw[2] = {qw0, qw1};
    const half_t *sc[2] = {sc0, sc1};
    const int32_t *qz[2] = {qz0, qz1};
    const half2_t ones = {(half_t)1.0f, (half_t)1.0f};
    const uint32_t MSK = sreg_const(UP::MSK_C), MAG = vreg_const(UP::MAG_C);
    const xrow_t *xq[MR];
#pragma unroll
    for (int m = 0; m < MR; m++) xq[m] = (const xrow_t *)(x + (size_t)(m < M ? m : M - 1) * ldx);

    float yv[MR][NS][4];
#pragma unroll
    for (int m = 0; m < MR; m++)
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) yv[m][s][j] = 0.f;

    for (uint32_t c = slice; c < nchunk; c += (uint32_t)S) {
        const uint32_t row = c * (4 * U) + wave * U;
        if (row >= (uint32_t)rows) continue;
        u32x4 w[NS][U];
        half4_t s4[NS];
        uint32_t zw[NS];
        const uint32_t g = gshift >= 0 ? (row >> gshift) : 0u;
#pragma unroll
        for (int s = 0; s < NS; s++) {
#pragma unroll
            for (int u = 0; u < U; u++)
                w[s][u] = __builtin_nontemporal_load((const u32x4 *)(qw[s] + (size_t)(row + u) * (uint32_t)N + nc));
            s4[s] = *(const half4_t *)(sc[s] + (size_t)g * (uint32_t)N + nc);
            zw[s] = (uint32_t)qz[s][(size_t)g * ((uint32_t)N / KPW) + nc / KPW];
        }
        xrow_t xr[MR][U];
#pragma unroll
        for (int m = 0; m < MR; m++)
#pragma unroll
            for (int u = 0; u < U; u++) xr[m][u] = xq[m][row + u];   // wave-uniform: scalar loads
        __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is in flight before any math

        float acc[MR][NS][4];
        float xs[MR];
#pragma unroll
        for (int m = 0; m < MR; m++) {
            xs[m] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[m][s][j] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            half2_t X[MR][NP];
#pragma unroll
            for (int m = 0; m < MR; m++)
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    const uint32_t a = xr[m][u][q / 2], b = xr[m][u][(q + NP) / 2];
                    X[m][q] = as_half2((q & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16)));
                    xs[m] = __builtin_amdgcn_fdot2(X[m][q], ones, xs[m], false);
                }
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    half2_t t[NP];
                    UP::pairs_rc(w[s][u][j], t, MSK, MAG);
#pragma unroll
                    for (int m = 0; m < MR; m++)
#pragma unroll
                        for (int q = 0; q < NP; q++) acc[m][s][j] = __builtin_amdgcn_fdot2(t[q], X[m][q], acc[m][s][j], false);
                }
        }
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float zf = (float)(((zw[s] >> (BITS * ((nc + j) % KPW))) & ((1u << BITS) - 1u)) + 1u) + UP::OFF;
#pragma unroll
                for (int m = 0; m < MR; m++) yv[m][s][j] += (float)s4[s][j] * (acc[m][s][j] - zf * xs[m]);
            }
    }

#pragma unroll
    for (int m = 0; m < MR; m++)
#pragma unroll
        for (int s = 0; s < NS; s++) *(float4_t *)&red[m][s][wave][4 * lane] = float4_t{yv[m][s][0], yv[m][s][1], yv[m][s][2], yv[m][s][3]};
    __syncthreads();
    const int t = threadIdx.x;
    const uint32_t n = tile * 256 + t;
    if (n >= (uint32_t)N) return;
#pragma unroll
    for (int m = 0; m < MR; m++) {
        if (m >= M) break;
        float t0 = red[m][0][0][t] + red[m][0][1][t] + red[m][0][2][t] + red[m][0][3][t], t1 = 0.f;
        if constexpr (FUSED2) t1 = red[m][1][0][t] + red[m][1][1][t] + red[m][1][2][t] + red[m][1][3][t];
        bool mine = true;
        if (S > 1) {
            const size_t wi = (size_t)m * (uint32_t)N + n;
            if constexpr (FUSED2) mine = splitk_add2(ws + 2 * wi, t0, t1, S, t0, t1);
            else mine = splitk_add1(ws + wi, t0, S, t0);
        }
        if (mine) {
            float v = t0;
            if constexpr (FUSED2) v = t0 * (1.0f / (1.0f + __expf(-t0))) * t1;
            half_t h = (half_t)v;
            if (bias) h = (half_t)((float)h + (float)bias[n]);
            y[(size_t)m * ldy + n] = h;
        }
    }
}

// 2 <= p.M <= 4, 4-bit: u = rows in flight per wave (MR = 2: 8 or 4; MR = 4: 4 or 2 -- x rows live in SGPRs)
int gemv_rowwave_mr_dispatch(bool fused2, int u, const GemvParams &p, hipStream_t s) {
    if (p.M < 2 || p.M > 4 || p.xperm || p.norm_w) return GPTQ_E_VARIANT;
    const int rows = p.K / 8;
    dim3 grid((p.N + 255) / 256, p.split_k), block(256);
#define GPTQ_MR_LAUNCH(U_, F_, MR_)                                                                                                         \
    hipLaunchKernelGGL((gemv_rowwave_mr_kernel<U_, F_, MR_>), grid, block, 0, s, p.qw[0], p.x, (int)p.ldx, p.sc[0], p.qz[0], p.N, rows,      \
                       p.split_k, p.upg_shift, p.qw[1], p.sc[1], p.qz[1], p.y, (int)p.ldy, p.ws, p.bias, p.M)
    if (p.M == 2) {
        if (u == 8) { if (fused2) GPTQ_MR_LAUNCH(8, true, 2); else GPTQ_MR_LAUNCH(8, false, 2); }
        else if (u == 4) { if (fused2) GPTQ_MR_LAUNCH(4, true, 2); else GPTQ_MR_LAUNCH(4, false, 2); }
        else return GPTQ_E_VARIANT;
    } else {
        if (u == 4) { if (fused2) GPTQ_MR_LAUNCH(4, true, 4); else GPTQ_MR_LAUNCH(4, false, 4); }
        else if (u == 2) { if (fused2) GPTQ_MR_LAUNCH(2, true, 4); else GPTQ_MR_LAUNCH(2, false, 4); }
        else return GPTQ_E_VARIANT;
    }
#undef GPTQ_MR_LAUNCH
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// rowwave with MFMA 4x4x4 for 2 <= M <= 8 (small decode batches).  v_mfma_f32_4x4x4_16b_f16 is 16 independent
// 4 x 4 x 4 products, one per group of four lanes: lane l supplies ONE column of B (4 k values) and receives rows
// i = 0..3 of that column of D; A is row l % 4 of x.  In the rowwave layout a lane owns 4 columns x 8 k per packed
// row, so a packed word is two B operands -- and the magic-exponent unpack already yields them in the right shape:
// t_q = {OFF + q_q, OFF + q_{q+4}} -> B1 = (t0, t1) = k order [0,4,1,5], B2 = (t2, t3) = [2,6,3,7], with
// A1 = (X0, X1), A2 = (X2, X3) the same x pairs the dot2 path uses.  Per packed word: the 7-VALU unpack + 2 MFMAs
// per block of four x rows (instead of 4 dot2 per row): the kernel stays unpack-bound up to M = 8.
// sum_k x (for the zero-point term) comes from the same MFMA against a B of ones.
// ---------------------------------------------------------------------------------------
template <int U, bool FUSED2, int MB>
__global__ void __launch_bounds__(256) gemv_rowwave_mfma_kernel(const uint32_t *__restrict__ qw0, const half_t *__restrict__ x, int ldx,
                                                                const half_t *__restrict__ sc0, const int32_t *__restrict__ qz0, int N,
                                                                int rows, int S, int gshift, const uint32_t *__restrict__ qw1,
                                                                const half_t *__restrict__ sc1, const int32_t *__restrict__ qz1,
                                                                half_t *__restrict__ y, int ldy, u64_t *__restrict__ ws,
                                                                const half_t *__restrict__ bias, int M) {
    constexpr int BITS = 4;
    using UP = Unpack<BITS>;
    constexpr int KPW = UP::KPW, NP = UP::NP;
    constexpr int NS = FUSED2 ? 2 : 1;
    constexpr int MR = 4 * MB;
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    __shared__ float red[MR][NS][4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x, slice = blockIdx.y;
    const uint32_t n0 = tile * 256 + lane * 4;
    const uint32_t nc = n0 < (uint32_t)N ? n0 : 0;
    const uint32_t nchunk = ((uint32_t)rows + 4 * U - 1) / (4 * U);
    const uint32_t *qw[2] = {qw0, qw1};
    const half_t *sc[2] = {sc0, sc1};
    const int32_t *qz[2] = {qz0, qz1};
    const uint32_t MSK = sreg_const(UP::MSK_C), MAG = vreg_const(UP::MAG_C);
    const h4_t ones = {(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
    const half_t *xrow[MB];   // this lane's A row of each block of four x rows (rows past M re-read row M-1; never written)
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        const int m = 4 * mb + (lane & 3);
        xrow[mb] = x + (size_t)(m < M ? m : M - 1) * ldx;
    }

    float4_t yv[MB][NS][4];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int c = 0; c < 4; c++) yv[mb][s][c] = (float4_t)0.f;

    for (uint32_t ch = slice; ch < nchunk; ch += (uint32_t)S) {
        const uint32_t row = ch * (4 * U) + wave * U;
        if (row >= (uint32_t)rows) continue;
        // x first (L2 hits; vector loads return in order, so behind the weights they would arrive last)
        u32x4 xv[MB][U];
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int u = 0; u < U; u++) xv[mb][u] = *(const u32x4 *)(xrow[mb] + (size_t)(row + u) * KPW);
        u32x4 w[NS][U];
        half4_t s4[NS];
        uint32_t zw[NS];
        const uint32_t g = gshift >= 0 ? (row >> gshift) : 0u;
#pragma unroll
        for (int s = 0; s < NS; s++) {
#pragma unroll
            for (int u = 0; u < U; u++)
                w[s][u] = __builtin_nontemporal_load((const u32x4 *)(qw[s] + (size_t)(row + u) * (uint32_t)N + nc));
            s4[s] = *(const half4_t *)(sc[s] + (size_t)g * (uint32_t)N + nc);
            zw[s] = (uint32_t)qz[s][(size_t)g * ((uint32_t)N / KPW) + nc / KPW];
        }
        __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is in flight before any math

        float4_t acc[MB][NS][4], xs[MB];
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            xs[mb] = (float4_t)0.f;
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[mb][s][c] = (float4_t)0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            h4_t A1[MB], A2[MB];
#pragma unroll
            for (int mb = 0; mb < MB; mb++) {
                const u32x4 v = xv[mb][u];
                // halves h0..h7 of this row's 8 k -> pairs X_q = (h_q, h_{q+4}); A1 = (X0, X1), A2 = (X2, X3)
                const uint32_t X0 = __builtin_amdgcn_perm(v[2], v[0], 0x05040100u), X1 = __builtin_amdgcn_perm(v[2], v[0], 0x07060302u);
                const uint32_t X2 = __builtin_amdgcn_perm(v[3], v[1], 0x05040100u), X3 = __builtin_amdgcn_perm(v[3], v[1], 0x07060302u);
                A1[mb] = __builtin_bit_cast(h4_t, u32x2{X0, X1});
                A2[mb] = __builtin_bit_cast(h4_t, u32x2{X2, X3});
                xs[mb] = __builtin_amdgcn_mfma_f32_4x4x4f16(A1[mb], ones, xs[mb], 0, 0, 0);
                xs[mb] = __builtin_amdgcn_mfma_f32_4x4x4f16(A2[mb], ones, xs[mb], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    half2_t t[NP];
                    UP::pairs_rc(w[s][u][c], t, MSK, MAG);
                    const h4_t B1 = __builtin_bit_cast(h4_t, u32x2{as_u32(t[0]), as_u32(t[1])});
                    const h4_t B2 = __builtin_bit_cast(h4_t, u32x2{as_u32(t[2]), as_u32(t[3])});
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) {
                        acc[mb][s][c] = __builtin_amdgcn_mfma_f32_4x4x4f16(A1[mb], B1, acc[mb][s][c], 0, 0, 0);
                        acc[mb][s][c] = __builtin_amdgcn_mfma_f32_4x4x4f16(A2[mb], B2, acc[mb][s][c], 0, 0, 0);
                    }
                }
        }
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float zf = (float)(((zw[s] >> (BITS * ((nc + c) % KPW))) & ((1u << BITS) - 1u)) + 1u) + UP::OFF;
                const float sv = (float)s4[s][c];
#pragma unroll
                for (int mb = 0; mb < MB; mb++)
#pragma unroll
                    for (int i = 0; i < 4; i++) yv[mb][s][c][i] += sv * (acc[mb][s][c][i] - zf * xs[mb][i]);
            }
    }

    // red[m][s][wave][column]: lane holds columns 4*lane + c, rows 4*mb + i
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int s = 0; s < NS; s++)
                *(float4_t *)&red[4 * mb + i][s][wave][4 * lane] = float4_t{yv[mb][s][0][i], yv[mb][s][1][i], yv[mb][s][2][i], yv[mb][s][3][i]};
    __syncthreads();
    const int t = threadIdx.x;
    const uint32_t n = tile * 256 + t;
    if (n >= (uint32_t)N) return;
#pragma unroll
    for (int m = 0; m < MR; m++) {
        if (m >= M) break;
        float t0 = red[m][0][0][t] + red[m][0][1][t] + red[m][0][2][t] + red[m][0][3][t], t1 = 0.f;
        if constexpr (FUSED2) t1 = red[m][1][0][t] + red[m][1][1][t] + red[m][1][2][t] + red[m][1][3][t];
        bool mine = true;
        if (S > 1) {
            const size_t wi = (size_t)m * (uint32_t)N + n;
            if constexpr (FUSED2) mine = splitk_add2(ws + 2 * wi, t0, t1, S, t0, t1);
            else mine = splitk_add1(ws + wi, t0, S, t0);
        }
        if (mine) {
            float v = t0;
            if constexpr (FUSED2) v = t0 * (1.0f / (1.0f + __expf(-t0))) * t1;
            half_t h = (half_t)v;
            if (bias) h = (half_t)((float)h + (float)bias[n]);
            y[(size_t)m * ldy + n] = h;
        }
    }
}

// 2 <= p.M <= 8, 4-bit: MFMA 4x4x4 rowwave (M <= 4: one block of four x rows, fused gate/up allowed; M <= 8: two blocks)
int gemv_rowwave_mfma_dispatch(bool fused2, int u, const GemvParams &p, hipStream_t s) {
    if (p.M < 2 || p.M > 8 || p.xperm || p.norm_w || (fused2 && p.M > 4)) return GPTQ_E_VARIANT;
    const int rows = p.K / 8;
    dim3 grid((p.N + 255) / 256, p.split_k), block(256);
#define GPTQ_MF_LAUNCH(U_, F_, MB_)                                                                                                        \
    hipLaunchKernelGGL((gemv_rowwave_mfma_kernel<U_, F_, MB_>), grid, block, 0, s, p.qw[0], p.x, (int)p.ldx, p.sc[0], p.qz[0], p.N, rows,   \
                       p.split_k, p.upg_shift, p.qw[1], p.sc[1], p.qz[1], p.y, (int)p.ldy, p.ws, p.bias, p.M)
    if (p.M <= 4) {
        if (u == 8) { if (fused2) GPTQ_MF_LAUNCH(8, true, 1); else GPTQ_MF_LAUNCH(8, false, 1); }
        else if (u == 4) { if (fused2) GPTQ_MF_LAUNCH(4, true, 1); else GPTQ_MF_LAUNCH(4, false, 1); }
        else return GPTQ_E_VARIANT;
    } else {
        if (u == 8) GPTQ_MF_LAUNCH(8, false, 2);
        else if (u == 4) GPTQ_MF_LAUNCH(4, false, 2);
        else return GPTQ_E_VARIANT;
    }
#undef GPTQ_MF_LAUNCH
    return (int)hipGetLastError();
}

// M == 1.  u = packed rows in flight per wave (8, 4 or 2; rows % u == 0 and a wave's u rows lie
// in one quantisation group); p.split_k = workgroups per 256-column tile; p.upg_shift = log2 of
// the packed rows per group or -1 (one group); p.ws zeroed workspace when split_k > 1.
int gemv_fast_dispatch(int bits, bool fused2, int u, const GemvParams &p, hipStream_t s) {
    if (bits == 3) {   // u = 32-k blocks in flight per wave (2 or 1); p.upg_shift = log2(blocks per group) or -1
        if (p.norm_w || p.xperm) return GPTQ_E_VARIANT;
        dim3 grid((p.N + 255) / 256, p.split_k), block(256);
        const int nblocks = p.K / 32;
#define GPTQ_R3_LAUNCH(UB_, F_)                                                                                                       \
    hipLaunchKernelGGL((gemv_rowwave3_kernel<UB_, F_>), grid, block, 0, s, p.qw[0], p.x, p.sc[0], p.qz[0], p.N, nblocks, p.split_k,    \
                       p.upg_shift, p.qw[1], p.sc[1], p.qz[1], p.y, p.ws, p.bias)
        if (u == 2) { if (fused2) GPTQ_R3_LAUNCH(2, true); else GPTQ_R3_LAUNCH(2, false); }
        else if (u == 1) { if (fused2) GPTQ_R3_LAUNCH(1, true); else GPTQ_R3_LAUNCH(1, false); }
        else return GPTQ_E_VARIANT;
#undef GPTQ_R3_LAUNCH
        return (int)hipGetLastError();
    }
    switch (bits) {
        case 2: return fused2 ? launch_rowwave_u<2, true>(u, p, s) : launch_rowwave_u<2, false>(u, p, s);
        case 4: return fused2 ? launch_rowwave_u<4, true>(u, p, s) : launch_rowwave_u<4, false>(u, p, s);
        case 8: return fused2 ? launch_rowwave_u<8, true>(u, p, s) : launch_rowwave_u<8, false>(u, p, s);
    }
    return GPTQ_E_BITS;
}

template <int BITS, bool FUSED2>
static int launch_generic_m(int nl, const GemvParams &p, hipStream_t s) {
    if (nl == 4) {
        if (p.M <= 1) return launch_generic<BITS, 4, 4, 1, FUSED2>(p, s);
        if (p.M <= 2) return launch_generic<BITS, 4, 4, 2, FUSED2>(p, s);
        if constexpr (FUSED2) return GPTQ_E_VARIANT; else return launch_generic<BITS, 4, 4, 4, FUSED2>(p, s);
    }
    if (p.M <= 1) return launch_generic<BITS, 16, 4, 1, FUSED2>(p, s);
    if (p.M <= 2) return launch_generic<BITS, 16, 4, 2, FUSED2>(p, s);
    if constexpr (FUSED2) return GPTQ_E_VARIANT; else return launch_generic<BITS, 16, 4, 4, FUSED2>(p, s);
}

int gemv_generic_dispatch(int bits, bool fused2, int nl, const GemvParams &p, hipStream_t s) {
    switch (bits) {
        case 2: return fused2 ? launch_generic_m<2, true>(nl, p, s) : launch_generic_m<2, false>(nl, p, s);
        case 3: return fused2 ? launch_generic_m<3, true>(nl, p, s) : launch_generic_m<3, false>(nl, p, s);
        case 4: return fused2 ? launch_generic_m<4, true>(nl, p, s) : launch_generic_m<4, false>(nl, p, s);
        case 8: return fused2 ? launch_generic_m<8, true>(nl, p, s) : launch_generic_m<8, false>(nl, p, s);
    }
    return GPTQ_E_BITS;
}

}  // namespace gptq
