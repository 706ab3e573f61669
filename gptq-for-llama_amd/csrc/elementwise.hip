// elementwise.hip -- the small fused LLaMA ops around QuantLinear, as plain HIP for gfx950:
//   rmsnorm   reference quant/triton_norm.py:7-39  (rms_norm_fwd_fused)
//   rope      reference quant/fused_attn.py:8-58,91 (rotate_half_kernel)
//   pack      reference quant/quant_linear.py:325-371 (QuantLinear.pack, CPU upstream)
//   g_idx triviality check (load-time helper, no reference counterpart)
// All are bandwidth/latency bound: 16-byte vector accesses, wave shuffles + one LDS hop.
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

// ------------------------------------------------------------------------------ RMSNorm
// One workgroup per row; the row stays in registers between the two passes.  The norm weight is requested WITH the row (round 5; rounds 1-4 loaded
// it after the barrier: a second dependent memory round trip; same arithmetic, same bits).  Measured in the batched decode engine's 16-row step
// (two launches per layer): 4.83 us per launch before and after, minimum 2.0 -- what rocprofv3 reports for a 16-workgroup launch inside a graph is
// the kernel-to-kernel dependency (the 4.9 us of stripe_mm_reduce_kernel and the 4.1 us of a 64-byte copyBuffer say the same), not this code.
template <int VPT>
__global__ void __launch_bounds__(256) rmsnorm_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                      const half_t *__restrict__ w, half_t *__restrict__ y,
                                                      int64_t ldy, int N, float eps) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const half_t *xr = x + (size_t)row * ldx;
    half8_t v[VPT], wv[VPT];
    float ss = 0.f;
    const int nv = N / 8;
#pragma unroll
    for (int i = 0; i < VPT; i++) {
        const int c = tid + i * 256;
        v[i] = (half8_t)(half_t)0;
        wv[i] = (half8_t)(half_t)0;
        if (c < nv) {
            v[i] = *(const half8_t *)(xr + (size_t)c * 8);
            wv[i] = *(const half8_t *)(w + (size_t)c * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < VPT; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float f = (float)v[i][j];
            ss += f * f;
        }
    }
    ss = wave_sum_xor(ss, 1);
    __shared__ float part[4];
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float var = (part[0] + part[1] + part[2] + part[3]) / (float)N;
    const float rstd = 1.0f / sqrtf(var + eps);
    half_t *yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < VPT; i++) {
        const int c = tid + i * 256;
        if (c < nv) {
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (half_t)((float)v[i][j] * rstd * (float)wv[i][j]);
            *(half8_t *)(yr + (size_t)c * 8) = o;
        }
    }
}

// any N / alignment: two passes over global memory like the reference kernel.
__global__ void __launch_bounds__(256) rmsnorm_scalar_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                             const half_t *__restrict__ w, half_t *__restrict__ y,
                                                             int64_t ldy, int N, float eps) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const half_t *xr = x + (size_t)row * ldx;
    float ss = 0.f;
    for (int c = tid; c < N; c += 256) {
        const float f = (float)xr[c];
        ss += f * f;
    }
    ss = wave_sum_xor(ss, 1);
    __shared__ float part[4];
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float var = (part[0] + part[1] + part[2] + part[3]) / (float)N;
    const float rstd = 1.0f / sqrtf(var + eps);
    half_t *yr = y + (size_t)row * ldy;
    for (int c = tid; c < N; c += 256) yr[c] = (half_t)((float)xr[c] * rstd * (float)w[c]);
}

int rmsnorm_launch(const half_t *x, int64_t ldx, const half_t *w, half_t *y, int64_t ldy, int M, int N,
                   float eps, hipStream_t s) {
    if (M == 0) return 0;
    const bool vec = (N % 8 == 0) && (ldx % 8 == 0) && (ldy % 8 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) % 16 == 0);
    dim3 grid(M), block(256);
    const int nv = N / 8;
    if (!vec || nv > 16 * 256) {
        hipLaunchKernelGGL(rmsnorm_scalar_kernel, grid, block, 0, s, x, ldx, w, y, ldy, N, eps);
    } else if (nv <= 256) {
        hipLaunchKernelGGL(rmsnorm_kernel<1>, grid, block, 0, s, x, ldx, w, y, ldy, N, eps);
    } else if (nv <= 512) {
        hipLaunchKernelGGL(rmsnorm_kernel<2>, grid, block, 0, s, x, ldx, w, y, ldy, N, eps);
    } else if (nv <= 1024) {
        hipLaunchKernelGGL(rmsnorm_kernel<4>, grid, block, 0, s, x, ldx, w, y, ldy, N, eps);
    } else if (nv <= 2048) {
        hipLaunchKernelGGL(rmsnorm_kernel<8>, grid, block, 0, s, x, ldx, w, y, ldy, N, eps);
    } else {
        hipLaunchKernelGGL(rmsnorm_kernel<16>, grid, block, 0, s, x, ldx, w, y, ldy, N, eps);
    }
    return (int)hipGetLastError();
}

// ---- round 6: K slices' combine + residual + the NEXT RMSNorm in one launch (16-row tiles, stripe_mm.inc) ----
// A decode batch of 9 .. 16 rows runs LLaMA's down_proj as K slices + a combine launch, and the next block starts with a stand-alone RMSNorm launch
// of the same rows (at 16 rows the norm fused into the qkv launch costs more than a launch: every workgroup would normalise all 16 x 4096 elements
// itself).  Here the combine owns whole ROWS -- one workgroup per row, the slices' partials arrive as fp32 rows [S][16][N] -- so it also writes
// h = rmsnorm(y) * w for the consumer: one launch less per decoder block.  Arithmetic: the slices are added in slice order and the residual / bias is
// added to the rounded sum exactly as stripe_mm_reduce_kernel does; the norm is rmsnorm_kernel's, thread for thread (same pieces per thread, same
// order of the sum of squares, this file's strict floating-point flags): y and h are the bits the two launches produce.
template <int VPT>
__global__ void __launch_bounds__(256) slices_combine_norm_kernel(const float *__restrict__ partials, int S, int N, const half_t *__restrict__ add, int64_t ldb,
                                                                  half_t *__restrict__ y, int64_t ldy, const half_t *__restrict__ nw, float eps,
                                                                  half_t *__restrict__ h, int64_t ldh) {
    const int row = blockIdx.x, tid = threadIdx.x, nv = N / 8;
    half8_t v[VPT], wv[VPT];
#pragma unroll
    for (int i = 0; i < VPT; i++) {
        const int c = tid + i * 256;
        v[i] = (half8_t)(half_t)0;
        wv[i] = (half8_t)(half_t)0;
        if (c < nv) {
            const float *first = partials + (size_t)row * N + (size_t)c * 8;
            wv[i] = *(const half8_t *)(nw + (size_t)c * 8);
            half8_t av = (half8_t)(half_t)0;
            if (add) av = *(const half8_t *)(add + (ldb ? (size_t)row * ldb : (size_t)0) + (size_t)c * 8);
            // (four slices requested together, clamped and masked like stripe_mm_reduce_kernel's eight: with a run-time trip count the loads of one
            // slice were requested only after the previous slice's had been added -- S x VPT dependent round trips to partials another XCD wrote:
            // 6.7 us per launch in the B = 16 engine profile against 4.9 for the reduce kernel it replaced)
            float4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
            for (int sl0 = 0; sl0 < S; sl0 += 4) {
                float4_t vl[4], vh[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float *src = first + (size_t)min(sl0 + q, S - 1) * 16 * N;
                    vl[q] = *(const float4_t *)src;
                    vh[q] = *(const float4_t *)(src + 4);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (sl0 + q < S) {
                        lo += vl[q];
                        hi += vh[q];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                half_t hh = (half_t)(j < 4 ? lo[j & 3] : hi[j & 3]);
                if (add) hh = (half_t)((float)hh + (float)av[j]);
                v[i][j] = hh;
            }
            *(half8_t *)(y + (size_t)row * ldy + (size_t)c * 8) = v[i];
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float f = (float)v[i][j];
            ss += f * f;
        }
    }
    ss = wave_sum_xor(ss, 1);
    __shared__ float part[4];
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float var = (part[0] + part[1] + part[2] + part[3]) / (float)N;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < VPT; i++) {
        const int c = tid + i * 256;
        if (c < nv) {
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (half_t)((float)v[i][j] * rstd * (float)wv[i][j]);
            *(half8_t *)(h + (size_t)row * ldh + (size_t)c * 8) = o;
        }
    }
}

int slices_combine_norm_launch(const float *partials, int S, int M, int N, const half_t *add, int64_t ldb, half_t *y, int64_t ldy, const half_t *nw, float eps,
                               half_t *h, int64_t ldh, hipStream_t s) {
    const int nv = N / 8;
    dim3 grid(M), block(256);
    if (nv <= 256) hipLaunchKernelGGL(slices_combine_norm_kernel<1>, grid, block, 0, s, partials, S, N, add, ldb, y, ldy, nw, eps, h, ldh);
    else if (nv <= 512) hipLaunchKernelGGL(slices_combine_norm_kernel<2>, grid, block, 0, s, partials, S, N, add, ldb, y, ldy, nw, eps, h, ldh);
    else if (nv <= 1024) hipLaunchKernelGGL(slices_combine_norm_kernel<4>, grid, block, 0, s, partials, S, N, add, ldb, y, ldy, nw, eps, h, ldh);
    else if (nv <= 2048) hipLaunchKernelGGL(slices_combine_norm_kernel<8>, grid, block, 0, s, partials, S, N, add, ldb, y, ldy, nw, eps, h, ldh);
    else hipLaunchKernelGGL(slices_combine_norm_kernel<16>, grid, block, 0, s, partials, S, N, add, ldb, y, ldy, nw, eps, h, ldh);
    return (int)hipGetLastError();
}

// --------------------------------------------------------------------------------- RoPE
// One workgroup per (batch, position) row.  A thread owns VW adjacent rotary columns: it
// computes their cos/sin once (fp32, accurate expf/cosf/sinf like the reference's libdevice
// calls) and walks over the 2*heads head slots of q and k.
template <int VW>
__global__ void __launch_bounds__(256) rope_kernel(half_t *qk, int64_t row_stride, const int64_t *pos,
                                                   int64_t pos_batch_stride, int seq, int nheads2, int head_dim,
                                                   float inv_base) {
    typedef half_t vec_t __attribute__((ext_vector_type(VW)));
    const int row = blockIdx.x, tid = threadIdx.x;
    const int half = head_dim / 2;
    const int cv = half / VW;           // column vectors per head
    const int hstep = 256 / cv;         // heads processed per pass (cv <= 256 checked on host)
    const int c = (tid % cv) * VW, h0 = tid / cv;
    if (h0 >= hstep) return;
    const int b = row / seq, t = row % seq;
    const float p = (float)pos[(size_t)b * pos_batch_stride + t];
    float cs[VW], sn[VW];
#pragma unroll
    for (int j = 0; j < VW; j++) {
        const float freq = expf((float)(c + j) * inv_base) * p;
        cs[j] = cosf(freq);
        sn[j] = sinf(freq);
    }
    half_t *base = qk + (size_t)row * row_stride + c;
    for (int h = h0; h < nheads2; h += hstep) {
        half_t *px = base + (size_t)h * head_dim;
        vec_t xv = *(const vec_t *)px, yv = *(const vec_t *)(px + half), ox, oy;
#pragma unroll
        for (int j = 0; j < VW; j++) {
            const float xf = (float)xv[j], yf = (float)yv[j];
            ox[j] = (half_t)(xf * cs[j] - yf * sn[j]);
            oy[j] = (half_t)(xf * sn[j] + yf * cs[j]);
        }
        *(vec_t *)px = ox;
        *(vec_t *)(px + half) = oy;
    }
}

int rope_launch(half_t *qk, int64_t row_stride, const int64_t *pos, int64_t pos_batch_stride, int bsz, int seq,
                int heads, int head_dim, float base, hipStream_t s) {
    const int rows = bsz * seq;
    if (rows == 0) return 0;
    const int half = head_dim / 2;
    const float inv_base = -2.0f * logf(base) / (float)head_dim;
    dim3 grid(rows), block(256);
    const bool a16 = ((uintptr_t)qk % 16 == 0) && (row_stride % 8 == 0);
    if (half % 8 == 0 && a16 && half / 8 <= 256) {
        hipLaunchKernelGGL(rope_kernel<8>, grid, block, 0, s, qk, row_stride, pos, pos_batch_stride, seq, 2 * heads,
                           head_dim, inv_base);
    } else if (half % 2 == 0 && ((uintptr_t)qk % 4 == 0) && (row_stride % 2 == 0) && half / 2 <= 256) {
        hipLaunchKernelGGL(rope_kernel<2>, grid, block, 0, s, qk, row_stride, pos, pos_batch_stride, seq, 2 * heads,
                           head_dim, inv_base);
    } else if (half <= 256) {
        hipLaunchKernelGGL(rope_kernel<1>, grid, block, 0, s, qk, row_stride, pos, pos_batch_stride, seq, 2 * heads,
                           head_dim, inv_base);
    } else {
        return GPTQ_E_SHAPE;
    }
    return (int)hipGetLastError();
}

// --------------------------------------------------------------------- g_idx triviality
__global__ void __launch_bounds__(1024) gidx_trivial_kernel(const int32_t *g_idx, int K, int groupsize, int32_t *out) {
    int ok = 1;
    for (int k = threadIdx.x; k < K; k += 1024) ok &= (g_idx[k] == k / groupsize);
    const int all = __syncthreads_and(ok);
    if (threadIdx.x == 0) *out = all;
}

int gidx_trivial_launch(const int32_t *g_idx, int K, int groupsize, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(gidx_trivial_kernel, dim3(1), dim3(1024), 0, s, g_idx, K, groupsize, out);
    return (int)hipGetLastError();
}

// --------------------------------------------------------------------------------- pack
// Bit-exact restatement of QuantLinear.pack on the GPU.  The float ops are written with
// explicit round-to-nearest intrinsics so that no FMA contraction changes the rounding:
//   sz = z * s ; v = (W + sz) / fp16(s) ; iw = (int) rint(v) ; word |= (uint)iw << (bits*j)
// (the OR is unmasked, exactly like the numpy code; 3-bit fields are masked -- extension).
template <int BITS>
__global__ void __launch_bounds__(256) pack_qweight_kernel(const float *__restrict__ weight, const float *__restrict__ scales,
                                                           const float *__restrict__ zeros, const int32_t *__restrict__ g_idx,
                                                           int K, int N, int G, int groupsize, uint32_t *__restrict__ qweight) {
    // workgroup: one 32-k block x 64 columns.  LDS transposes the [n][k] weight tile.
    __shared__ float tile[64][33];
    const int blk = blockIdx.x, n_base = blockIdx.y * 64, tid = threadIdx.x;
    for (int idx = tid; idx < 64 * 32; idx += 256) {
        const int nn = idx / 32, kk = idx % 32;
        const int n = n_base + nn;
        tile[nn][kk] = (n < N) ? weight[(size_t)n * K + (size_t)blk * 32 + kk] : 0.f;
    }
    __syncthreads();
    if (tid >= 64) return;
    const int n = n_base + tid;
    if (n >= N) return;
    uint32_t words[BITS];
#pragma unroll
    for (int i = 0; i < BITS; i++) words[i] = 0u;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int k = blk * 32 + j;
        const int g = g_idx ? g_idx[k] : k / groupsize;
        const float s = scales[(size_t)n * G + g], z = zeros[(size_t)n * G + g];
        const float s16 = (float)(half_t)s;
        const float v = __fdiv_rn(__fadd_rn(tile[tid][j], __fmul_rn(z, s)), s16);
        const uint32_t u = (uint32_t)(int32_t)rintf(v);
        if constexpr (BITS == 3) {
            const int bit = 3 * j, wi = bit >> 5, o = bit & 31;
            const uint64_t sh = (uint64_t)(u & 7u) << o;
            words[wi] |= (uint32_t)sh;
            if (wi < 2) words[wi + 1] |= (uint32_t)(sh >> 32);
        } else {
            constexpr int KPW = 32 / BITS;
            words[j / KPW] |= u << (BITS * (j % KPW));
        }
    }
#pragma unroll
    for (int i = 0; i < BITS; i++) qweight[((size_t)blk * BITS + i) * N + n] = words[i];
}

template <int BITS>
__global__ void __launch_bounds__(256) pack_qzeros_kernel(const float *__restrict__ scales, const float *__restrict__ zeros,
                                                          int N, int G, uint32_t *__restrict__ qzeros,
                                                          half_t *__restrict__ scales16) {
    // one thread per (g, 32-column block)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nb = N / 32;
    if (idx >= G * nb) return;
    const int g = idx / nb, b = idx % nb;
    uint32_t words[BITS];
#pragma unroll
    for (int i = 0; i < BITS; i++) words[i] = 0u;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int n = b * 32 + j;
        scales16[(size_t)g * N + n] = (half_t)scales[(size_t)n * G + g];
        const float zf = __fadd_rn(zeros[(size_t)n * G + g], -1.0f);
        const uint32_t u = (uint32_t)(int64_t)zf;  // -1.0 -> 0xFFFFFFFF like numpy's astype(uint32)
        if constexpr (BITS == 3) {
            const int bit = 3 * j, wi = bit >> 5, o = bit & 31;
            const uint64_t sh = (uint64_t)(u & 7u) << o;
            words[wi] |= (uint32_t)sh;
            if (wi < 2) words[wi + 1] |= (uint32_t)(sh >> 32);
        } else {
            constexpr int KPW = 32 / BITS;
            words[j / KPW] |= u << (BITS * (j % KPW));
        }
    }
#pragma unroll
    for (int i = 0; i < BITS; i++) qzeros[(size_t)g * (nb * BITS) + (size_t)b * BITS + i] = words[i];
}

template <int BITS>
static int pack_bits(const float *weight, const float *scales, const float *zeros, const int32_t *g_idx, int K, int N,
                     int G, int groupsize, int32_t *qweight, int32_t *qzeros, half_t *scales16, hipStream_t s) {
    dim3 grid(K / 32, (N + 63) / 64);
    hipLaunchKernelGGL(pack_qweight_kernel<BITS>, grid, dim3(256), 0, s, weight, scales, zeros, g_idx, K, N, G, groupsize,
                       (uint32_t *)qweight);
    const int total = G * (N / 32);
    hipLaunchKernelGGL(pack_qzeros_kernel<BITS>, dim3((total + 255) / 256), dim3(256), 0, s, scales, zeros, N, G,
                       (uint32_t *)qzeros, scales16);
    return (int)hipGetLastError();
}

int pack_launch(const float *weight, const float *scales, const float *zeros, const int32_t *g_idx, int K, int N, int G,
                int bits, int groupsize, int32_t *qweight, int32_t *qzeros, half_t *scales16, hipStream_t s) {
    switch (bits) {
        case 2: return pack_bits<2>(weight, scales, zeros, g_idx, K, N, G, groupsize, qweight, qzeros, scales16, s);
        case 3: return pack_bits<3>(weight, scales, zeros, g_idx, K, N, G, groupsize, qweight, qzeros, scales16, s);
        case 4: return pack_bits<4>(weight, scales, zeros, g_idx, K, N, G, groupsize, qweight, qzeros, scales16, s);
        case 8: return pack_bits<8>(weight, scales, zeros, g_idx, K, N, G, groupsize, qweight, qzeros, scales16, s);
    }
    return GPTQ_E_BITS;
}

// ------------------------------------------------------------------------------ dequantise
// W[k][n] = fp16(q - z) * fp16 scale (one fp16 rounding), the weight the reference's kernel builds
// on the fly (quant_linear.py:114-128), materialised as a dense fp16 [K, N] matrix.  Any bits in
// {2,3,4,8}, any g_idx.  A workgroup converts a 32-k block x 256 columns; writes are row-contiguous.
template <int BITS>
__global__ void __launch_bounds__(256) dequant_kernel(const uint32_t *__restrict__ qw, const half_t *__restrict__ sc,
                                                      const int32_t *__restrict__ qz, const int32_t *__restrict__ gi, int K, int N,
                                                      int G, int groupsize, half_t *__restrict__ out, int64_t ldo) {
    const int blk = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int ldz = N / 32 * BITS;
    uint32_t col[BITS];
#pragma unroll
    for (int i = 0; i < BITS; i++) col[i] = qw[((size_t)blk * BITS + i) * N + n];
    int g_prev = -1;
    half_t s = (half_t)0, z = (half_t)0;
#pragma unroll 8
    for (int j = 0; j < 32; j++) {
        const int k = blk * 32 + j;
        int g = gi ? gi[k] : k / groupsize;
        g = (g < 0 || g >= G) ? 0 : g;
        if (g != g_prev) {
            g_prev = g;
            s = sc[(size_t)g * N + n];
            z = (half_t)(float)zero_of<BITS>(qz + (size_t)g * ldz, n);
        }
        const int q = field_of_block<BITS>(col, j);
        out[(size_t)k * ldo + n] = (half_t)((half_t)(float)q - z) * s;
    }
}

// 4-bit, trivial g_idx, groupsize % 8 == 0: a thread owns 8 adjacent columns (two dwordx4 of one packed row,
// one qzeros word, 16 B of scales) and writes 8 rows x 16 B; same arithmetic, an order of magnitude fewer
// store instructions than the generic kernel's 2-byte stores.
__global__ void __launch_bounds__(256) dequant4_fast_kernel(const uint32_t *__restrict__ qw, const half_t *__restrict__ sc,
                                                            const int32_t *__restrict__ qz, int K, int N, int groupsize,
                                                            half_t *__restrict__ out, int64_t ldo) {
    const int n8 = (blockIdx.x * 256 + threadIdx.x) * 8, r = blockIdx.y;   // packed row r = k / 8
    if (n8 >= N) return;
    const int g = (r * 8) / groupsize;
    const u32x4 w0 = *(const u32x4 *)(qw + (size_t)r * N + n8), w1 = *(const u32x4 *)(qw + (size_t)r * N + n8 + 4);
    const half8_t s8 = *(const half8_t *)(sc + (size_t)g * N + n8);
    const uint32_t zw = (uint32_t)qz[(size_t)g * (N / 8) + n8 / 8];
    const uint32_t words[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
    half_t z[8];
#pragma unroll
    for (int c = 0; c < 8; c++) z[c] = (half_t)(float)(((zw >> (4 * c)) & 15u) + 1u);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        half8_t o;
#pragma unroll
        for (int c = 0; c < 8; c++) o[c] = (half_t)((half_t)(float)((words[c] >> (4 * j)) & 15u) - z[c]) * s8[c];
        *(half8_t *)(out + (size_t)(r * 8 + j) * ldo + n8) = o;
    }
}

// The same weight TRANSPOSED: Wt[n][k] (k contiguous, row stride ldo >= K) -- the operand layout of the hand-written prefill GEMM
// (gemm8.hip: both operands by LDS-DMA, a fragment = 16 contiguous bytes of one row).  Identical arithmetic, element for element.
// Workgroup tile: 64 columns x 256 k.  A thread owns one column and two of the tile's eight 32-k blocks: its BITS packed words per
// block are read coalesced across the 64 columns, its 32 values go to LDS as row n of the transposed tile; the tile then leaves as
// full 512-byte row segments (32 lanes x 16 B per row: whole 128-byte lines).  The first version stored straight from registers --
// 64 rows x 16 B per wave store -- and ran at 1.7 TB/s (20.6 us for a 4096^2 layer, profiles/r3f_final/prefill_kernel_stats.csv).
constexpr int DQT_COLS = 64, DQT_K = 256, DQT_PITCH = DQT_K + 8;   // halves; 16 bytes of padding per LDS row
template <int BITS, bool UNIFORM>   // UNIFORM: trivial g_idx and groupsize % 32 == 0 -> one (scale, zero) per thread and block
__global__ void __launch_bounds__(256) dequant_t_kernel(const uint32_t *__restrict__ qw, const half_t *__restrict__ sc,
                                                        const int32_t *__restrict__ qz, const int32_t *__restrict__ gi, int K, int N,
                                                        int G, int groupsize, half_t *__restrict__ out, int64_t ldo) {
    __shared__ __attribute__((aligned(16))) half_t tile[DQT_COLS * DQT_PITCH];
    const int tid = threadIdx.x, c = tid & 63, n = blockIdx.x * DQT_COLS + c, k0 = blockIdx.y * DQT_K;
    const int ldz = N / 32 * BITS;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int b32 = (tid >> 6) + 4 * h, blk = k0 / 32 + b32;   // 32-k block inside the tile / inside the layer
        if (n >= N || blk * 32 >= K) continue;
        uint32_t col[BITS];
#pragma unroll
        for (int i = 0; i < BITS; i++) col[i] = qw[((size_t)blk * BITS + i) * N + n];
        int g_prev = -1;
        half_t s = (half_t)0, z = (half_t)0;
        half_t v[32];
        if constexpr (UNIFORM) {
            const int g = (blk * 32) / groupsize;
            g_prev = g;
            s = sc[(size_t)g * N + n];
            z = (half_t)(float)zero_of<BITS>(qz + (size_t)g * ldz, n);
        }
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const int k = blk * 32 + j;
            int g = g_prev;
            if constexpr (!UNIFORM) {
                g = gi ? gi[k] : k / groupsize;
                g = (g < 0 || g >= G) ? 0 : g;
            }
            if (g != g_prev) {
                g_prev = g;
                s = sc[(size_t)g * N + n];
                z = (half_t)(float)zero_of<BITS>(qz + (size_t)g * ldz, n);
            }
            const int q = field_of_block<BITS>(col, j);
            v[j] = (half_t)((half_t)(float)q - z) * s;
        }
        half_t *dst = tile + c * DQT_PITCH + b32 * 32;
#pragma unroll
        for (int i = 0; i < 4; i++)
            *(half8_t *)(dst + 8 * i) = half8_t{v[8 * i], v[8 * i + 1], v[8 * i + 2], v[8 * i + 3], v[8 * i + 4], v[8 * i + 5], v[8 * i + 6], v[8 * i + 7]};
    }
    __syncthreads();
    // 64 rows x 32 pieces of 16 bytes; thread t: piece t % 32 of rows t / 32 + 8 i
    const int pc = tid & 31;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int r = (tid >> 5) + 8 * i, nn = blockIdx.x * DQT_COLS + r, k = k0 + pc * 8;
        if (nn < N && k < K) *(half8_t *)(out + (size_t)nn * ldo + k) = *(const half8_t *)(tile + r * DQT_PITCH + pc * 8);
    }
}

int dequant_t_launch(const uint32_t *qw, const half_t *sc, const int32_t *qz, const int32_t *gi, int K, int N, int G, int groupsize,
                     int bits, half_t *out, int64_t ldo, hipStream_t s) {
    if (ldo % 8 != 0 || ((uintptr_t)out % 16) != 0) return GPTQ_E_ALIGN;
    dim3 grid((N + DQT_COLS - 1) / DQT_COLS, (K + DQT_K - 1) / DQT_K), block(256);
    const bool uni = !gi && groupsize % 32 == 0;
#define GPTQ_DQT(B)                                                                                                                  \
    if (uni) hipLaunchKernelGGL((dequant_t_kernel<B, true>), grid, block, 0, s, qw, sc, qz, gi, K, N, G, groupsize, out, ldo);      \
    else hipLaunchKernelGGL((dequant_t_kernel<B, false>), grid, block, 0, s, qw, sc, qz, gi, K, N, G, groupsize, out, ldo)
    switch (bits) {
        case 2: GPTQ_DQT(2); break;
        case 3: GPTQ_DQT(3); break;
        case 4: GPTQ_DQT(4); break;
        case 8: GPTQ_DQT(8); break;
        default: return GPTQ_E_BITS;
    }
#undef GPTQ_DQT
    return (int)hipGetLastError();
}

int dequant_launch(const uint32_t *qw, const half_t *sc, const int32_t *qz, const int32_t *gi, int K, int N, int G, int groupsize,
                   int bits, half_t *out, int64_t ldo, hipStream_t s) {
    if (bits == 4 && !gi && groupsize % 8 == 0 && N % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)qw % 16) == 0 && ((uintptr_t)sc % 16) == 0 &&
        ((uintptr_t)out % 16) == 0) {
        hipLaunchKernelGGL(dequant4_fast_kernel, dim3((N / 8 + 255) / 256, K / 8), dim3(256), 0, s, qw, sc, qz, K, N, groupsize, out, ldo);
        return (int)hipGetLastError();
    }
    dim3 grid((N + 255) / 256, K / 32), block(256);
    switch (bits) {
        case 2: hipLaunchKernelGGL(dequant_kernel<2>, grid, block, 0, s, qw, sc, qz, gi, K, N, G, groupsize, out, ldo); break;
        case 3: hipLaunchKernelGGL(dequant_kernel<3>, grid, block, 0, s, qw, sc, qz, gi, K, N, G, groupsize, out, ldo); break;
        case 4: hipLaunchKernelGGL(dequant_kernel<4>, grid, block, 0, s, qw, sc, qz, gi, K, N, G, groupsize, out, ldo); break;
        case 8: hipLaunchKernelGGL(dequant_kernel<8>, grid, block, 0, s, qw, sc, qz, gi, K, N, G, groupsize, out, ldo); break;
        default: return GPTQ_E_BITS;
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------ silu(gate) * up
// c[m][n] = fp16(silu(fp32 gate[m][n]) * fp32 up[m][n]) -- the epilogue of the reference's fused MLP kernel (fused_mlp.py:160-165)
// as a pass of its own, for the route that hands the two products to a library GEMM (prefill).  HBM-bound: 6 B per element,
// 16-byte accesses, one row of the batch per blockIdx.y.
__global__ void __launch_bounds__(256) silu_mul_kernel(const half_t *__restrict__ g, int64_t ldg, const half_t *__restrict__ u, int64_t ldu,
                                                       half_t *__restrict__ c, int64_t ldc, int N) {
    const int n8 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (n8 >= N) return;
    const size_t m = blockIdx.y;
    const half8_t gv = *(const half8_t *)(g + m * ldg + n8), uv = *(const half8_t *)(u + m * ldu + n8);
    half8_t o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float a = (float)gv[i];
        o[i] = (half_t)(a * (1.0f / (1.0f + __expf(-a))) * (float)uv[i]);
    }
    *(half8_t *)(c + m * ldc + n8) = o;
}

int silu_mul_launch(const half_t *g, int64_t ldg, const half_t *u, int64_t ldu, half_t *c, int64_t ldc, int M, int N, hipStream_t s) {
    for (int m0 = 0; m0 < M; m0 += 65535) {       // gridDim.y limit
        const int rows = M - m0 < 65535 ? M - m0 : 65535;
        hipLaunchKernelGGL(silu_mul_kernel, dim3((N / 8 + 255) / 256, rows), dim3(256), 0, s, g + (size_t)m0 * ldg, ldg, u + (size_t)m0 * ldu, ldu,
                           c + (size_t)m0 * ldc, ldc, N);
    }
    return (int)hipGetLastError();
}

// the same epilogue on FP32 products (the prefill route's gate | up GEMM leaves its sums unrounded, exactly like the accumulators
// the reference feeds to SiLU, fused_mlp.py:160-165): 10 B per element
__global__ void __launch_bounds__(256) silu_mul_f32_kernel(const float *__restrict__ g, int64_t ldg, const float *__restrict__ u, int64_t ldu,
                                                           half_t *__restrict__ c, int64_t ldc, int N) {
    const int n8 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (n8 >= N) return;
    const size_t m = blockIdx.y;
    const float4_t g0 = *(const float4_t *)(g + m * ldg + n8), g1 = *(const float4_t *)(g + m * ldg + n8 + 4);
    const float4_t u0 = *(const float4_t *)(u + m * ldu + n8), u1 = *(const float4_t *)(u + m * ldu + n8 + 4);
    half8_t o;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        o[i] = (half_t)(g0[i] * (1.0f / (1.0f + __expf(-g0[i]))) * u0[i]);
        o[i + 4] = (half_t)(g1[i] * (1.0f / (1.0f + __expf(-g1[i]))) * u1[i]);
    }
    *(half8_t *)(c + m * ldc + n8) = o;
}

int silu_mul_f32_launch(const float *g, int64_t ldg, const float *u, int64_t ldu, half_t *c, int64_t ldc, int M, int N, hipStream_t s) {
    for (int m0 = 0; m0 < M; m0 += 65535) {       // gridDim.y limit
        const int rows = M - m0 < 65535 ? M - m0 : 65535;
        hipLaunchKernelGGL(silu_mul_f32_kernel, dim3((N / 8 + 255) / 256, rows), dim3(256), 0, s, g + (size_t)m0 * ldg, ldg, u + (size_t)m0 * ldu,
                           ldu, c + (size_t)m0 * ldc, ldc, N);
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------- x[:, perm] (act-order batches)
// xg[m][k] = x[m][perm[k]]: the input of a group-sorted act-order layer (the decode kernel fuses this gather; batches pay one
// pass over M K halves).  Thread = 8 consecutive k of one row: eight 2-byte gathers (L2-resident x), one 16-byte store.
__global__ void __launch_bounds__(256) gather_cols_kernel(const half_t *__restrict__ x, int64_t ldx, const int32_t *__restrict__ perm,
                                                          half_t *__restrict__ xg, int64_t ldg, int K) {
    const int k8 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (k8 >= K) return;
    const size_t m = blockIdx.y;
    const half_t *row = x + m * ldx;
    half8_t o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = row[perm[k8 + i]];
    *(half8_t *)(xg + m * ldg + k8) = o;
}

int gather_cols_launch(const half_t *x, int64_t ldx, const int32_t *perm, half_t *xg, int64_t ldg, int M, int K, hipStream_t s) {
    if (K % 8 != 0 || ldg % 8 != 0 || ((uintptr_t)xg % 16) != 0) return GPTQ_E_ALIGN;
    for (int m0 = 0; m0 < M; m0 += 65535) {
        const int rows = M - m0 < 65535 ? M - m0 : 65535;
        hipLaunchKernelGGL(gather_cols_kernel, dim3((K / 8 + 255) / 256, rows), dim3(256), 0, s, x + (size_t)m0 * ldx, ldx, perm, xg + (size_t)m0 * ldg,
                           ldg, K);
    }
    return (int)hipGetLastError();
}

// y[m][n] = fp16(y[m][n] + r[m][n]): the residual add of a decoder layer (HF: hidden = residual + proj(..), one fp16 add) for the routes of a
// decode batch whose kernel has no residual epilogue.  Thread = 8 consecutive n of one row.
__global__ void __launch_bounds__(256) add_rows_kernel(half_t *__restrict__ y, int64_t ldy, const half_t *__restrict__ r, int64_t ldr, int N) {
    const int n8 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (n8 >= N) return;
    const size_t m = blockIdx.y;
    half8_t a = *(const half8_t *)(y + m * ldy + n8);
    const half8_t b = *(const half8_t *)(r + m * ldr + n8);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (half_t)((float)a[i] + (float)b[i]);
    *(half8_t *)(y + m * ldy + n8) = a;
}

int add_rows_launch(half_t *y, int64_t ldy, const half_t *r, int64_t ldr, int M, int N, hipStream_t s) {
    if (N % 8 != 0 || ldy % 8 != 0 || ldr % 8 != 0 || ((uintptr_t)y % 16) != 0 || ((uintptr_t)r % 16) != 0) return GPTQ_E_ALIGN;
    if (M <= 0 || M > 65535) return GPTQ_E_SHAPE;
    hipLaunchKernelGGL(add_rows_kernel, dim3((N / 8 + 255) / 256, M), dim3(256), 0, s, y, ldy, r, ldr, N);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------- act-order row sort
// qweight_out row r', field j  <-  the field of k = perm[r' * f + j] in qweight (f = 32 / bits).
// With perm = stable argsort(g_idx) every packed row of the output holds f consecutive members of
// ONE group and the group of sorted position k' is k' / groupsize: the act-order layer becomes a
// trivial-g_idx layer applied to x[perm] (load-time re-sort).  One-off, at load.
template <int BITS>
__global__ void __launch_bounds__(256) act_order_repack_kernel(const uint32_t *__restrict__ qw, const int32_t *__restrict__ perm,
                                                               int K, int N, uint32_t *__restrict__ out) {
    constexpr int F = 32 / BITS;
    const int n = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (n >= N) return;
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < F; j++) {
        const int k = perm[r * F + j];
        const uint32_t src = qw[(size_t)(k / F) * N + n];
        w |= ((src >> (BITS * (k % F))) & ((1u << BITS) - 1u)) << (BITS * j);
    }
    out[(size_t)r * N + n] = w;
}

int act_order_repack_launch(const uint32_t *qw, const int32_t *perm, int K, int N, int bits, uint32_t *out, hipStream_t s) {
    dim3 block(256);
    switch (bits) {
        case 2: hipLaunchKernelGGL(act_order_repack_kernel<2>, dim3((N + 255) / 256, K / 16), block, 0, s, qw, perm, K, N, out); break;
        case 4: hipLaunchKernelGGL(act_order_repack_kernel<4>, dim3((N + 255) / 256, K / 8), block, 0, s, qw, perm, K, N, out); break;
        case 8: hipLaunchKernelGGL(act_order_repack_kernel<8>, dim3((N + 255) / 256, K / 4), block, 0, s, qw, perm, K, N, out); break;
        default: return GPTQ_E_BITS;
    }
    return (int)hipGetLastError();
}

// ---- test support: overwrite every CU's LDS (soak tests: a decode launch must not depend on what the previous kernel left there) ----
__global__ void __launch_bounds__(256) dirty_lds_kernel(uint32_t pattern, uint32_t *sink, int words) {
    extern __shared__ uint32_t lds_all[];
    for (int i = threadIdx.x; i < words; i += 256) lds_all[i] = pattern ^ (uint32_t)i;
    __syncthreads();
    if (sink && lds_all[(threadIdx.x * 97) % words] == 0x12345678u) sink[0] = 1;   // (keeps the stores alive)
}

int dirty_lds_launch(uint32_t pattern, hipStream_t s) {
    constexpr int BYTES = 160 * 1024 - 256;
    static LdsOptIn opt_in;
    if (int rc = opt_in.ensure((const void *)dirty_lds_kernel, BYTES)) return rc;
    hipLaunchKernelGGL(dirty_lds_kernel, dim3(1024), dim3(256), BYTES, s, pattern, (uint32_t *)nullptr, BYTES / 4);   // one workgroup per CU at a time: four rounds
    return (int)hipGetLastError();
}

}  // namespace gptq
