// decode_attn.hip -- the two non-GEMV pieces of a batch-1 decode step that the reference does with
// torch ops between its Triton kernels (quant/fused_attn.py:126-155): RoPE on q,k + KV-cache
// append (triton_rotate_half_ :126, torch.cat :142-143) and the single-query attention
// (F.scaled_dot_product_attention :155).  Written as plain HIP so that the whole decode step can
// be captured in ONE hipGraph: the current position is read from device memory, the KV cache is
// a preallocated [t_max, heads*head_dim] buffer (no torch.cat, no shape change per token).
//
// HBM-bound elementwise / reduction work (K,V rows are read once per token): no MFMA.
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

// ---------------------------------------------------------------------------------------
// RoPE (rotate-half, fp32 trig exactly like rope_kernel / reference :43-57) on q (in place) and
// k, then k,v -> cache row `pos`.  grid = heads, block = head_dim/2 threads.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rope_kv_kernel(half_t *__restrict__ qkv, const int64_t *__restrict__ pos_ptr,
                                                      half_t *__restrict__ kc, half_t *__restrict__ vc, int heads, int head_dim,
                                                      int t_max, float inv_base) {
    const int h = blockIdx.x, c = threadIdx.x, half = head_dim / 2;
    if (c >= half) return;
    const int64_t pos = pos_ptr[0];
    if (pos < 0 || pos >= t_max) return;
    const float freq = expf((float)c * inv_base) * (float)pos;
    const float cs = cosf(freq), sn = sinf(freq);
    const int hd = heads * head_dim;
    half_t *q = qkv + (size_t)h * head_dim + c;
    half_t *k = qkv + hd + (size_t)h * head_dim + c;
    const half_t *v = qkv + 2 * hd + (size_t)h * head_dim + c;
    const float qx = (float)q[0], qy = (float)q[half];
    q[0] = (half_t)(qx * cs - qy * sn);
    q[half] = (half_t)(qx * sn + qy * cs);
    const float kx = (float)k[0], ky = (float)k[half];
    half_t *kd = kc + (size_t)pos * hd + (size_t)h * head_dim + c;
    kd[0] = (half_t)(kx * cs - ky * sn);
    kd[half] = (half_t)(kx * sn + ky * cs);
    half_t *vd = vc + (size_t)pos * hd + (size_t)h * head_dim + c;
    vd[0] = v[0];
    vd[half] = v[half];
}

// ---------------------------------------------------------------------------------------
// Single-query attention over the cache rows [0, pos], head_dim == 128.
// grid = (heads, nsplit): split s owns timesteps [s*TS, (s+1)*TS); splits past the current length
// exit at once.  Partial = {max, sum, acc[128]} in fp32; attn_combine_kernel merges the splits.
// ---------------------------------------------------------------------------------------
constexpr int ATT_TS = 128;   // timesteps per split
constexpr int ATT_HD = 128;   // head_dim served
constexpr int ATT_REC = ATT_HD + 2;

__global__ void __launch_bounds__(256) attn_partial_kernel(const half_t *__restrict__ q, const half_t *__restrict__ kc,
                                                           const half_t *__restrict__ vc, const int64_t *__restrict__ pos_ptr,
                                                           float *__restrict__ ws, int heads, int t_max, float scale) {
    __shared__ float sc[ATT_TS];
    __shared__ float red[8];
    __shared__ float accs[4][ATT_HD];
    const int h = blockIdx.x, s = blockIdx.y, nsplit = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t len = pos_ptr[0] + 1;
    if (len > t_max) len = t_max;
    const int t0 = s * ATT_TS;
    if (t0 >= len) return;
    const int nact = (int)((len - t0) < ATT_TS ? (len - t0) : ATT_TS);
    const int hd = heads * ATT_HD;

    // ---- scores: 16 lanes x 8 dims per timestep, 16 timesteps per pass ------------------------
    const int d8 = tid & 15, tsub = tid >> 4;
    const half8_t q8 = *(const half8_t *)(q + (size_t)h * ATT_HD + d8 * 8);
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) qf[j] = (float)q8[j];
#pragma unroll
    for (int it = 0; it < ATT_TS / 16; it++) {
        const int tl = it * 16 + tsub;
        float dot = 0.f;
        if (tl < nact) {
            const half8_t k8 = *(const half8_t *)(kc + (size_t)(t0 + tl) * hd + (size_t)h * ATT_HD + d8 * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) dot += qf[j] * (float)k8[j];
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        dot += __shfl_xor(dot, 8, 64);
        if (d8 == 0) sc[tl] = (tl < nact) ? dot * scale : -INFINITY;
    }
    __syncthreads();

    // ---- softmax statistics of the chunk ------------------------------------------------------
    float sv = (tid < ATT_TS) ? sc[tid] : -INFINITY;
    float m = sv;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float p = (tid < nact) ? __expf(sv - m) : 0.f;
    float l = p;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) l += __shfl_xor(l, off, 64);
    __syncthreads();                 // everyone has read sc[] and red[0..3]
    if (tid < ATT_TS) sc[tid] = p;
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    l = red[4] + red[5] + red[6] + red[7];

    // ---- acc[d] = sum_t p_t V[t][d]: 64 lanes x 2 dims, 4 waves over the timesteps --------------
    float a0 = 0.f, a1 = 0.f;
    const half_t *vb = vc + (size_t)t0 * hd + (size_t)h * ATT_HD + 2 * lane;
    for (int tl = wave; tl < nact; tl += 4) {
        const half2_t v2 = *(const half2_t *)(vb + (size_t)tl * hd);
        const float pt = sc[tl];
        a0 += pt * (float)v2[0];
        a1 += pt * (float)v2[1];
    }
    accs[wave][2 * lane] = a0;
    accs[wave][2 * lane + 1] = a1;
    __syncthreads();
    float *rec = ws + ((size_t)h * nsplit + s) * ATT_REC;
    if (tid < ATT_HD) rec[2 + tid] = accs[0][tid] + accs[1][tid] + accs[2][tid] + accs[3][tid];
    if (tid == 0) {
        rec[0] = m;
        rec[1] = l;
    }
}

__global__ void __launch_bounds__(ATT_HD) attn_combine_kernel(const float *__restrict__ ws, const int64_t *__restrict__ pos_ptr,
                                                              half_t *__restrict__ out, int nsplit, int t_max) {
    const int h = blockIdx.x, d = threadIdx.x;
    int64_t len = pos_ptr[0] + 1;
    if (len > t_max) len = t_max;
    if (len < 1) len = 1;
    const int nact = (int)((len + ATT_TS - 1) / ATT_TS);
    const float *base = ws + (size_t)h * nsplit * ATT_REC;
    float M = -INFINITY;
    for (int s = 0; s < nact; s++) M = fmaxf(M, base[(size_t)s * ATT_REC]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nact; s++) {
        const float *rec = base + (size_t)s * ATT_REC;
        const float w = __expf(rec[0] - M);
        num += w * rec[2 + d];
        den += w * rec[1];
    }
    out[(size_t)h * ATT_HD + d] = (half_t)(num / den);
}

int decode_rope_kv_launch(half_t *qkv, const int64_t *pos, half_t *kc, half_t *vc, int heads, int head_dim, int t_max, float base,
                          hipStream_t s) {
    const float inv_base = -2.0f * logf(base) / (float)head_dim;   // reference fused_attn.py:91
    hipLaunchKernelGGL(rope_kv_kernel, dim3(heads), dim3(head_dim / 2), 0, s, qkv, pos, kc, vc, heads, head_dim, t_max, inv_base);
    return (int)hipGetLastError();
}

int decode_attn_launch(const half_t *q, const half_t *kc, const half_t *vc, const int64_t *pos, half_t *out, float *ws, int heads,
                       int t_max, float scale, hipStream_t s) {
    const int nsplit = (t_max + ATT_TS - 1) / ATT_TS;
    hipLaunchKernelGGL(attn_partial_kernel, dim3(heads, nsplit), dim3(256), 0, s, q, kc, vc, pos, ws, heads, t_max, scale);
    hipLaunchKernelGGL(attn_combine_kernel, dim3(heads), dim3(ATT_HD), 0, s, ws, pos, out, nsplit, t_max);
    return (int)hipGetLastError();
}

size_t decode_attn_ws_bytes(int heads, int t_max) {
    return (size_t)heads * ((t_max + ATT_TS - 1) / ATT_TS) * ATT_REC * sizeof(float);
}

}  // namespace gptq
