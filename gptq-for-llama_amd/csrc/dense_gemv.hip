// dense_gemv.hip -- y[M][N] = x[M][K] . W[N][K]^T for 1 .. 16 rows of x and a dense fp16 weight stored [out, in]: the LM head of a decode step
// (one row) or of a decode batch (round 5).
//
// The reference's decode step ends in an ordinary fp16 nn.Linear (the HF model's lm_head; llama_inference.py:119-127 -> generate): 32000 x 4096
// = 262 MB per token for LLaMA-7B, streamed once -- HBM-bound like the quantised matvecs, and until round 4 the last launch of the engine that
// went to a library (torch.matmul -> hipBLASLt, 55 us).  Same recipe as the stripe decode kernel, without the unpack: x staged once per
// workgroup in LDS (optionally RMS-normalised on the way: the final norm of the model, arithmetic of triton_norm.py:22-39, fp16-rounded
// like the stand-alone launch it replaces), a wave owns whole rows -- 8 KiB contiguous each at K = 4096 -- and keeps two of them (16
// wave loads of 1 KiB) in flight, non-temporal; products by v_dot2_f32_f16 into fp32, one DPP / permlane sum per row, fp16 store.
#include <cstdlib>

#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {
namespace {

constexpr int DG_WAVES = 4;        // waves per workgroup
constexpr int DG_CHUNK = 4096;     // k per pass over a row pair: 8 sixteen-byte pieces per lane and row

GPTQ_DEV float dot8(const u32x4 a, const u32x4 b, float acc) {
#pragma unroll
    for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_fdot2(as_half2(a[q]), as_half2(b[q]), acc, false);
    return acc;
}

// MR rows of x (round 5: the LM head of a decode BATCH): the weight stream is read once for all rows -- every 16-byte piece of W meets
// MR pieces of x from LDS.  MR = 16 sits near the LDS read rate (16 KiB of x per KiB of W), still one pass over the 262 MB.
template <bool NORM, int MR>
__global__ void __launch_bounds__(DG_WAVES * 64) dense_gemv_kernel(const half_t *__restrict__ x, int64_t ldx, const half_t *__restrict__ W, int64_t ldw,
                                                                  half_t *__restrict__ y, int64_t ldy, int M, int N, int K,
                                                                  const half_t *__restrict__ nw, float eps, const half_t *__restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = DG_WAVES * 64;
    half_t *xl = (half_t *)smem;                       // [MR][Kp]: x (normalised), zero beyond K
    __shared__ float part[DG_WAVES][MR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Kp = (K + DG_CHUNK - 1) / DG_CHUNK * DG_CHUNK, np = K / 8;

    // ---- x -> LDS (every workgroup stages all of it: 8 KB of L2 hits per row against the 256 KB of weights it streams) ----
    float ss[MR];
#pragma unroll
    for (int m = 0; m < MR; m++) {
        ss[m] = 0.f;
        const half_t *xr = x + (size_t)min(m, M - 1) * ldx;
        for (int i = tid; i < Kp / 8; i += T) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < np) v = *(const u32x4 *)(xr + (size_t)i * 8);
            if constexpr (NORM) ss[m] = dot8(v, v, ss[m]);
            *(u32x4 *)(xl + (size_t)m * Kp + (size_t)i * 8) = v;
        }
    }
    if constexpr (NORM) {
#pragma unroll
        for (int m = 0; m < MR; m++) {
            ss[m] = wave_sum_xor(ss[m], 1);
            if (lane == 0) part[wave][m] = ss[m];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; m++) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < DG_WAVES; w++) tot += part[w][m];
            const float rstd = 1.0f / sqrtf(tot / (float)K + eps);
            for (int i = tid; i < np; i += T) {            // each thread re-reads the pieces it wrote itself
                const u32x4 v = *(const u32x4 *)(xl + (size_t)m * Kp + (size_t)i * 8), g = *(const u32x4 *)(nw + (size_t)i * 8);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const half2_t a = as_half2(v[q]), b = as_half2(g[q]);
                    o[q] = as_u32(half2_t{(half_t)((float)a[0] * rstd * (float)b[0]), (half_t)((float)a[1] * rstd * (float)b[1])});
                }
                *(u32x4 *)(xl + (size_t)m * Kp + (size_t)i * 8) = o;
            }
        }
    }
    __syncthreads();

    // ---- rows of W: two per wave and pass ----
    const int gw = blockIdx.x * DG_WAVES + wave, nwv = gridDim.x * DG_WAVES;
    for (int n0 = 2 * gw; n0 < N; n0 += 2 * nwv) {
        const int n1 = min(n0 + 1, N - 1);
        const half_t *r0 = W + (size_t)n0 * ldw, *r1 = W + (size_t)n1 * ldw;
        float a0[MR], a1[MR];
#pragma unroll
        for (int m = 0; m < MR; m++) a0[m] = a1[m] = 0.f;
        for (int k0 = 0; k0 < Kp; k0 += DG_CHUNK) {
            u32x4 w0[8], w1[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int c = min(k0 / 8 + i * 64 + lane, np - 1);     // past K: a valid address, x is zero there
                w0[i] = __builtin_nontemporal_load((const u32x4 *)(r0 + (size_t)c * 8));
                w1[i] = __builtin_nontemporal_load((const u32x4 *)(r1 + (size_t)c * 8));
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
#pragma unroll
                for (int m = 0; m < MR; m++) {
                    const u32x4 xv = *(const u32x4 *)(xl + (size_t)m * Kp + (size_t)(k0 / 8 + i * 64 + lane) * 8);
                    a0[m] = dot8(w0[i], xv, a0[m]);
                    a1[m] = dot8(w1[i], xv, a1[m]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MR; m++) {
            a0[m] = wave_sum_xor(a0[m], 1);
            a1[m] = wave_sum_xor(a1[m], 1);
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MR; m++) {
                if (m < M) {
                    half_t h0 = (half_t)a0[m], h1 = (half_t)a1[m];
                    if (bias) {
                        h0 = (half_t)((float)h0 + (float)bias[n0]);
                        h1 = (half_t)((float)h1 + (float)bias[n1]);
                    }
                    y[(size_t)m * ldy + n0] = h0;
                    if (n0 + 1 < N) y[(size_t)m * ldy + n0 + 1] = h1;
                }
            }
        }
    }
}

template <bool NORM, int MR>
int dense_gemv_launch_mr(const half_t *x, int64_t ldx, const half_t *W, int64_t ldw, const half_t *bias, half_t *y, int64_t ldy, int M, int N, int K,
                         const half_t *norm_w, float eps, hipStream_t s) {
    const int Kp = (K + DG_CHUNK - 1) / DG_CHUNK * DG_CHUNK;
    const size_t lds = (size_t)MR * Kp * 2;
    if (lds > 144 * 1024) return GPTQ_E_VARIANT;
    // one pass of two rows per wave keeps 16 KiB in flight; enough workgroups for ~4 passes per wave, at most 8 resident per CU
    int grid = (N + 2 * DG_WAVES * 4 - 1) / (2 * DG_WAVES * 4);
    grid = std::max(1, std::min(grid, 2048));
    static LdsOptIn opt_in;   // per instantiation; per device inside
    auto kern = dense_gemv_kernel<NORM, MR>;
    if (int rc = opt_in.ensure((const void *)kern, lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(DG_WAVES * 64), lds, s, x, ldx, W, ldw, y, ldy, M, N, K, norm_w, eps, bias);
    return (int)hipGetLastError();
}

template <int MR>
int dense_gemv_launch_rows(const half_t *x, int64_t ldx, const half_t *W, int64_t ldw, const half_t *bias, half_t *y, int64_t ldy, int M, int N, int K,
                           const half_t *norm_w, float eps, hipStream_t s) {
    return norm_w ? dense_gemv_launch_mr<true, MR>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s)
                  : dense_gemv_launch_mr<false, MR>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
}

// ---- 5 .. 16 rows on the matrix core (round 5) ----
// The dot2 kernel above meets every 16-byte piece of W with MR pieces of x from LDS: at 16 rows that is 16 KiB of LDS reads and 64 dot2 per
// KiB of weights -- 205 us for the 262 MB LLaMA-7B head against 43 at one row (gpurun_out r5a).  Here the weights still arrive row-contiguous
// (a wave instruction = 2 rows x 512 B, non-temporal), are parked in a wave-private LDS tile (16 rows x 256 k; rows padded by 32 bytes, piece p
// of row r at position p ^ (r & 3): the layout of csrc/stripe_mm.inc, conflict-free for ds_read_b128) and come back as B fragments of
// v_mfma_f32_16x16x32_f16 (lane l: W row n0 + l % 16, 8 consecutive k of slot l / 16); the A fragments are x -- every wave owns a K range of
// 1024 k and keeps ITS 32 fragments of all 16 rows of x in registers for the whole launch (RMS-normalised in the prologue: one rstd per row,
// arithmetic of triton_norm.py:22-39), so x costs no LDS traffic at all.  Two 8-KiB chunks per wave stay in flight across tile boundaries; the
// NW partial tiles of a 16-row tile of W meet through LDS (one barrier per tile).  HBM-bound for every M <= 16: 32 MFMAs per 32 KiB of weights.
constexpr int DM_KQ = 1024;             // k per wave
constexpr int DM_CH = 256;              // k per chunk (8 wave loads of 2 rows x 512 B)
constexpr int DM_RS = DM_CH + 16;       // halves per staged row

template <int NW, bool NORM>
__global__ void __launch_bounds__(NW * 64, 2) dense_mm16_kernel(const half_t *__restrict__ x, int64_t ldx, const half_t *__restrict__ W, int64_t ldw,
                                                            half_t *__restrict__ y, int64_t ldy, int M, int N, const half_t *__restrict__ nw, float eps,
                                                            const half_t *__restrict__ bias, int ntiles) {
    constexpr int NCH = DM_KQ / DM_CH, SPC = DM_CH / 32, NST = DM_KQ / 32, K = NW * DM_KQ;
    __shared__ __attribute__((aligned(16))) half_t stage[NW][16][DM_RS];
    __shared__ __attribute__((aligned(16))) float red[2][NW][256];
    __shared__ float part[NW][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mrow = lane & 15, slot = lane >> 4;
    const int kq = wave * DM_KQ;

    // ---- this wave's A fragments: x[m = lane % 16][kq + 32 j + 8 slot ..], j = 0 .. 31 (rows past M - 1 repeat the last one: masked at the store) ----
    half8_t xa[NST];
    {
        const half_t *xr = x + (size_t)min(mrow, M - 1) * ldx + kq + 8 * slot;
#pragma unroll
        for (int j = 0; j < NST; j++) xa[j] = *(const half8_t *)(xr + 32 * j);
    }
    if constexpr (NORM) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NST; j++) {
            const u32x4 v = __builtin_bit_cast(u32x4, xa[j]);
            ss = dot8(v, v, ss);
        }
        ss += __shfl_xor(ss, 16, 64);     // the four k slots of a row
        ss += __shfl_xor(ss, 32, 64);
        if (lane < 16) part[wave][lane] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) tot += part[w][mrow];
        const float rstd = 1.0f / sqrtf(tot / (float)K + eps);
        const half_t *gr = nw + kq + 8 * slot;
#pragma unroll
        for (int j = 0; j < NST; j++) {
            const half8_t g = *(const half8_t *)(gr + 32 * j);
#pragma unroll
            for (int e = 0; e < 8; e++) xa[j][e] = (half_t)((float)xa[j][e] * rstd * (float)g[e]);
        }
    }

    // ---- the stream of chunks: tile t = blockIdx.x + i gridDim.x, chunk c of the wave's K range; two chunks in flight ----
    const int lrow = lane >> 5, lpiece = lane & 31;
    u32x4 wA[8], wB[8];
    auto issue = [&](u32x4 (&w)[8], int tile, int c) {
        const int n0 = min(tile, ntiles - 1) * 16;       // past the last tile: re-read it (no branch in the load phase; the data is not used)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int n = min(n0 + 2 * i + lrow, N - 1);
            w[i] = __builtin_nontemporal_load((const u32x4 *)(W + (size_t)n * ldw + kq + c * DM_CH + 8 * lpiece));
        }
    };
    half_t *st = &stage[wave][0][0];
    auto park = [&](const u32x4 (&w)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int r = 2 * i + lrow;
            *(u32x4 *)(st + (size_t)r * DM_RS + 8 * (lpiece ^ (r & 3))) = w[i];
        }
    };
    int tile = blockIdx.x;
    issue(wA, tile, 0);
    issue(wB, tile, 1);
    // (hipcc interleaves these sixteen requests -- same base register, offsets 0 / 512: neither sched_barrier nor a compiler memory barrier keeps
    // chunk 0 entirely ahead of chunk 1 -- so the wait counters at the loop head are the merge of two orders, vmcnt(15, 13 .. 1) instead of
    // vmcnt(15 .. 8): a tile starts with most of BOTH chunks landed.  With eight waves per CU at different phases the stream stays fed.)
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            // (LDS operations of one wave execute in order: the stores of chunk c follow the fragment reads of chunk c - 1 without a barrier)
            if (c & 1) park(wB); else park(wA);
            const int tn = c + 2 < NCH ? tile : tile + (int)gridDim.x, cn = (c + 2) % NCH;
            if (c & 1) issue(wB, tn, cn); else issue(wA, tn, cn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < SPC; j++) {
                const half8_t b = *(const half8_t *)(st + (size_t)mrow * DM_RS + 8 * ((4 * j + slot) ^ (mrow & 3)));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[c * SPC + j], b, acc, 0, 0, 0);
            }
        }
        // ---- NW partial tiles -> one (two buffers: a tile's readers are a barrier away from the writers of the tile after next) ----
        *(float4_t *)(&red[buf][wave][lane * 4]) = acc;
        __syncthreads();
        if (tid < 256) {
            // element tid of the tile in the accumulator layout: lane' = tid / 4 holds column n0 + lane' % 16, rows 4 (lane' / 16) + r
            const int lp = tid >> 2, r = tid & 3;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) v += red[buf][w][tid];
            const int m = 4 * (lp >> 4) + r, n = tile * 16 + (lp & 15);
            if (m < M && n < N) {
                half_t h = (half_t)v;
                if (bias) h = (half_t)((float)h + (float)bias[n]);
                y[(size_t)m * ldy + n] = h;
            }
        }
    }
}

template <int NW>
int dense_mm16_launch(const half_t *x, int64_t ldx, const half_t *W, int64_t ldw, const half_t *bias, half_t *y, int64_t ldy, int M, int N,
                      const half_t *norm_w, float eps, hipStream_t s) {
    const int ntiles = (N + 15) / 16;
    const int grid = std::max(1, std::min(ntiles, 512));     // two workgroups per CU, ~4 tiles each for the 32000-row head
    if (norm_w) hipLaunchKernelGGL((dense_mm16_kernel<NW, true>), dim3(grid), dim3(NW * 64), 0, s, x, ldx, W, ldw, y, ldy, M, N, norm_w, eps, bias, ntiles);
    else hipLaunchKernelGGL((dense_mm16_kernel<NW, false>), dim3(grid), dim3(NW * 64), 0, s, x, ldx, W, ldw, y, ldy, M, N, norm_w, eps, bias, ntiles);
    return (int)hipGetLastError();
}

// rows from which the matrix-core kernel takes over (A/B runs: GPTQ_LM_HEAD_MFMA_MIN_ROWS; 17 = never).  Measured on the 32000 x 4096 head
// (gpurun_out r5c, us, dot2 / matrix core): 1 row 43.1 / 46.7, 2: 44.9 / 46.8, 4: 48.4 / 47.4, 5: 83 / 47.4, 8: 84 / 49.0, 16: 205 / 53.3
int dense_mm16_min_rows() {
    static const int v = [] { const char *e = getenv("GPTQ_LM_HEAD_MFMA_MIN_ROWS"); return e ? atoi(e) : 4; }();
    return v;
}

}  // namespace

// M rows of x (1 <= M <= 16; rows ldx apart, rows of y ldy apart): one pass over W for all of them
int dense_gemv_launch(const half_t *x, const half_t *W, int64_t ldw, const half_t *bias, half_t *y, int N, int K, const half_t *norm_w, float eps,
                      hipStream_t s, int M, int64_t ldx, int64_t ldy) {
    if (M >= dense_mm16_min_rows() && M <= 16 && K % DM_KQ == 0) {
        switch (K / DM_KQ) {   // K = waves x 1024 k: LLaMA-7B 4096, 13B 5120, 65B 8192
            case 4: return dense_mm16_launch<4>(x, ldx, W, ldw, bias, y, ldy, M, N, norm_w, eps, s);
            case 5: return dense_mm16_launch<5>(x, ldx, W, ldw, bias, y, ldy, M, N, norm_w, eps, s);
            case 8: return dense_mm16_launch<8>(x, ldx, W, ldw, bias, y, ldy, M, N, norm_w, eps, s);
            default: break;
        }
    }
    if (M <= 1) return dense_gemv_launch_rows<1>(x, K, W, ldw, bias, y, N, 1, N, K, norm_w, eps, s);
    if (M <= 2) return dense_gemv_launch_rows<2>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
    if (M <= 4) return dense_gemv_launch_rows<4>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
    if (M <= 8) return dense_gemv_launch_rows<8>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
    if (M <= 16) return dense_gemv_launch_rows<16>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
    return GPTQ_E_VARIANT;
}

}  // namespace gptq
