// transpose.hip -- dX[M,K] = dY[M,N] . deq(B)^T, the backward of QuantLinearFunction
// (reference quant/quant_linear.py:191-258 transpose_matmul_248_kernel, :272-279, :294-301).
// Not on the inference path: a straightforward LDS-tiled kernel that
// dequantises a [32 k][64 n] block per step with the reference's numerics (fp16 weight,
// fp32 accumulate) and serves any bits / g_idx.
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

template <int BITS>
__global__ void __launch_bounds__(256) transpose_kernel(const half_t *__restrict__ dy, int64_t lddy,
                                                        const uint32_t *__restrict__ qw, const half_t *__restrict__ sc,
                                                        const int32_t *__restrict__ qz, const int32_t *__restrict__ gi,
                                                        half_t *__restrict__ dx, int64_t lddx, int M, int K, int N, int G,
                                                        int groupsize) {
    constexpr int CH = BITS, MB = 8, NT = 64;
    __shared__ half_t wt[32][NT + 2];
    __shared__ half_t dyt[MB][NT + 2];
    __shared__ int gk[32];
    const int blk = blockIdx.x, m_base = blockIdx.y * MB, tid = threadIdx.x;
    const int ldz = N / 32 * BITS;
    if (tid < 32) {
        const int k = blk * 32 + tid;
        int g = gi ? gi[k] : k / groupsize;
        gk[tid] = (g < 0 || g >= G) ? 0 : g;
    }
    __syncthreads();
    const int kk = tid % 32, mm = tid / 32;
    float acc = 0.f;
    for (int nb = 0; nb < N; nb += NT) {
        // dequantise [32][NT]: thread -> (k = tid/8 .. , 8 columns)
        for (int idx = tid; idx < 32 * NT; idx += 256) {
            const int k = idx / NT, n = nb + idx % NT;
            half_t w = (half_t)0;
            if (n < N) {
                uint32_t col[CH];
#pragma unroll
                for (int i = 0; i < CH; i++) col[i] = qw[((size_t)blk * CH + i) * N + n];
                const int q = field_of_block<BITS>(col, k);
                const int g = gk[k];
                const int z = zero_of<BITS>(qz + (size_t)g * ldz, n);
                w = (half_t)(float)(q - z) * sc[(size_t)g * N + n];
            }
            wt[k][idx % NT] = w;
        }
        for (int idx = tid; idx < MB * NT; idx += 256) {
            const int m = m_base + idx / NT, n = nb + idx % NT;
            dyt[idx / NT][idx % NT] = (m < M && n < N) ? dy[(size_t)m * lddy + n] : (half_t)0;
        }
        __syncthreads();
#pragma unroll 8
        for (int n = 0; n < NT; n++) acc += (float)dyt[mm][n] * (float)wt[kk][n];
        __syncthreads();
    }
    const int m = m_base + mm;
    if (m < M) dx[(size_t)m * lddx + (size_t)blk * 32 + kk] = (half_t)acc;
}

int transpose_dispatch(int bits, const half_t *dy, int64_t lddy, const uint32_t *qw, const half_t *sc, const int32_t *qz,
                       const int32_t *gi, half_t *dx, int64_t lddx, int M, int K, int N, int G, int groupsize,
                       hipStream_t s) {
    dim3 grid(K / 32, (M + 7) / 8), block(256);
#define LAUNCH(B) hipLaunchKernelGGL(transpose_kernel<B>, grid, block, 0, s, dy, lddy, qw, sc, qz, gi, dx, lddx, M, K, N, G, groupsize)
    switch (bits) {
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        default: return GPTQ_E_BITS;
    }
#undef LAUNCH
    return (int)hipGetLastError();
}

}  // namespace gptq
