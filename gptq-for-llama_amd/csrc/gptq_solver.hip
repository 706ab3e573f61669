// gptq_solver.hip -- the sequential inner loop of the GPTQ solver (reference gptq.py:177-199) for one block of
// columns, all rows in parallel.  This is the CALLER side of the hot path: it produces the integer weights that
// QuantLinear.pack (quant_linear.py:325-371) packs and the matvec kernels read.
//
// The reference walks the <= 128 columns of a block one by one with a handful of torch launches per column
// (quantise, loss, outer product, subtract: gptq.py:190-198), i.e. ~1000 dependent launches per block.  Rows never
// interact, so here ONE launch per block gives every weight row to a wave: lane l keeps columns l and l + 64 of the
// block in registers, column i is broadcast with a lane read, rounded to its grid, and the error is fed forward
// into the lanes' remaining columns with the row i of the block's inverse-Hessian factor.  The arithmetic is the
// reference's, operation by operation (IEEE fp32, round-half-even, separate multiply and subtract -- this file is
// compiled without fast-math and without contraction):
//     q    = scale * (clamp(rint(w / scale) + zero, 0, maxq) - zero)         quantizer.py:28-32
//     loss = (w - q)^2 / d^2 ; err = (w - q) / d                              gptq.py:193-195
//     w[j] = w[j] - err * Hinv1[i][j]   for j >= i                            gptq.py:196
// The grid of a column is looked up in per-group (scale, zero) arrays that the host fills before the block from the
// global W, exactly as the reference fits it (gptq.py:181-183 reads W, not the in-block clone W1).
// Outputs: Q (block columns), Err1 (for the trailing update W[:, i2:] -= Err1 . Hinv[i1:i2, i2:], a GEMM the host
// issues, gptq.py:204) and the per-row loss sum (gptq.py:194, :202).
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

__global__ void __launch_bounds__(256) gptq_block_kernel(const float *__restrict__ W, int64_t ldw, const float *__restrict__ Hinv, int64_t ldh,
                                                         int rows, int i1, int count, int groupsize, float maxq,
                                                         const float *__restrict__ scale, const float *__restrict__ zero, int64_t ldg,
                                                         float *__restrict__ Q, int64_t ldq, float *__restrict__ Err, int64_t lde,
                                                         float *__restrict__ loss_rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *wrow = W + (size_t)row * ldw + i1;
    float w[2], q[2] = {0.f, 0.f}, e[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int j = lane + 64 * s;
        w[s] = j < count ? wrow[j] : 0.f;
    }
    float loss = 0.f;
    const float *hrow = Hinv + (size_t)i1 * ldh + i1;  // Hinv1[i][j] = hrow[i * ldh + j]
    // row i of the factor for this lane's two columns, requested one step ahead
    float h[2];
#pragma unroll
    for (int s = 0; s < 2; s++) h[s] = (lane + 64 * s) < count ? hrow[lane + 64 * s] : 0.f;
    int g_cur = -1;
    float sc = 1.f, zp = 0.f;
    for (int i = 0; i < count; i++) {
        float hn[2] = {0.f, 0.f};
        if (i + 1 < count) {
#pragma unroll
            for (int s = 0; s < 2; s++) hn[s] = (lane + 64 * s) < count ? hrow[(size_t)(i + 1) * ldh + lane + 64 * s] : 0.f;
        }
        const int g = (i1 + i) / groupsize;
        if (g != g_cur) {  // wave-uniform
            g_cur = g;
            sc = scale[(size_t)row * ldg + g];
            zp = zero[(size_t)row * ldg + g];
        }
        const float wi = __shfl((i & 64) ? w[1] : w[0], i & 63, 64);
        const float d = __shfl((i & 64) ? h[1] : h[0], i & 63, 64);  // Hinv1[i][i]
        const float lvl = fminf(fmaxf(rintf(wi / sc) + zp, 0.f), maxq);
        const float qi = sc * (lvl - zp);
        const float diff = wi - qi;
        loss += (diff * diff) / (d * d);
        const float err = diff / d;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int j = lane + 64 * s;
            if (j >= i && j < count) {
                const float p = err * h[s];
                w[s] = w[s] - p;
            }
            if (j == i) {
                q[s] = qi;
                e[s] = err;
            }
            h[s] = hn[s];
        }
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int j = lane + 64 * s;
        if (j < count) {
            Q[(size_t)row * ldq + i1 + j] = q[s];
            Err[(size_t)row * lde + j] = e[s];
        }
    }
    if (lane == 0) loss_rows[row] += loss * 0.5f;  // Losses1 / 2 (gptq.py:202)
}

int gptq_block_launch(const float *W, int64_t ldw, const float *Hinv, int64_t ldh, int rows, int i1, int count, int groupsize, int maxq,
                      const float *scale, const float *zero, int64_t ldg, float *Q, int64_t ldq, float *Err, int64_t lde, float *loss_rows,
                      hipStream_t s) {
    hipLaunchKernelGGL(gptq_block_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, W, ldw, Hinv, ldh, rows, i1, count, groupsize, (float)maxq, scale,
                       zero, ldg, Q, ldq, Err, lde, loss_rows);
    return (int)hipGetLastError();
}

}  // namespace gptq
