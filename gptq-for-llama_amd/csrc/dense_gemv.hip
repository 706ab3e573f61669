// dense_gemv.hip -- y[M][N] = x[M][K] . W[N][K]^T for 1 .. 16 rows of x and a dense fp16 weight stored [out, in]: the LM head of a decode step
// (one row) or of a decode batch (round 5).
//
// The reference's decode step ends in an ordinary fp16 nn.Linear (the HF model's lm_head; llama_inference.py:119-127 -> generate): 32000 x 4096
// = 262 MB per token for LLaMA-7B, streamed once -- HBM-bound like the quantised matvecs, and until round 4 the last launch of the engine that
// went to a library (torch.matmul -> hipBLASLt, 55 us).  Same recipe as the stripe decode kernel, without the unpack: x staged once per
// workgroup in LDS (optionally RMS-normalised on the way: the final norm of the model, arithmetic of triton_norm.py:22-39, fp16-rounded
// like the stand-alone launch it replaces), a wave owns whole rows -- 8 KiB contiguous each at K = 4096 -- and keeps two of them (16
// wave loads of 1 KiB) in flight, non-temporal; products by v_dot2_f32_f16 into fp32, one DPP / permlane sum per row, fp16 store.
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

}  // namespace

// M rows of x (1 <= M <= 16; rows ldx apart, rows of y ldy apart): one pass over W for all of them
int dense_gemv_launch(const half_t *x, const half_t *W, int64_t ldw, const half_t *bias, half_t *y, int N, int K, const half_t *norm_w, float eps,
                      hipStream_t s, int M, int64_t ldx, int64_t ldy) {
    if (M <= 1) return dense_gemv_launch_rows<1>(x, K, W, ldw, bias, y, N, 1, N, K, norm_w, eps, s);
    if (M <= 2) return dense_gemv_launch_rows<2>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
    if (M <= 4) return dense_gemv_launch_rows<4>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
    if (M <= 8) return dense_gemv_launch_rows<8>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
    if (M <= 16) return dense_gemv_launch_rows<16>(x, ldx, W, ldw, bias, y, ldy, M, N, K, norm_w, eps, s);
    return GPTQ_E_VARIANT;
}

}  // namespace gptq
