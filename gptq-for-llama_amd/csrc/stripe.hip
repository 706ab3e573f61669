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
// Bit widths: 4 (described below), 8 and 2 use the same geometry with KPW = 32 / bits k per word: a row block is 16 packed rows
// = 16 KPW k, a lane's dwordx4 = 4 KPW consecutive k of one column; fields are re-ordered (even k in the low half-word, odd k in
// the high one) so that ONE mask per field position yields a natural k pair against an fp16 magic whose ulp equals the field's
// bit weight: 8-bit 0x00FF00FF|1024 on w and w>>8 (1 shift + 2 and_or per 4 weights); 2-bit five positions [1:0]..[9:8] with the
// magics 1024, 256, 64, 16, 4 on w and three more on w>>6 (1 shift + 8 and_or per 16 weights).
// 3 bits: 32 k per lane = a dwordx3 (768-byte wave loads); ten fields per word at half-word bits [2:0] .. [14:12] (magics 1024, 128, 16 on
// w and 1024, 128 on w >> 9: 1 shift + 5 and_or per 10 weights), k 30 / k 31 assembled from the spare bits 15 / 31 of the three words.
// MR = 4 (2 <= M <= 4 rows of x): the MFMA computes four rows anyway -- lane l supplies row l%4 of x as the A operand and reads
// result row i -- so a small decode batch costs the same weight stream, unpack and MFMA count as M = 1.
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
#include <algorithm>

#include "gptq_device.h"
#include "gptq_internal.h"
#include "stripe_common.h"

namespace gptq {

namespace {

// perm != NULL (round 4): the image of the GROUP-SORTED rows of an act-order layer straight from the checkpoint layout -- packed row r of
// the sorted matrix holds k' = F r .. F r + F - 1, whose source is field perm[k'] % F of checkpoint row perm[k'] / F (what
// act_order_repack_kernel writes into a full sorted copy; that copy, one more qweight per layer, is no longer needed)
__global__ void __launch_bounds__(256) stripe_repack_kernel(const uint32_t *__restrict__ qw0, const uint32_t *__restrict__ qw1,
                                                            uint32_t *__restrict__ R, int N, int nrb, int NS, int bits,
                                                            const int32_t *__restrict__ perm) {
    // one thread per output word; consecutive threads -> consecutive j (rows), then lanes (columns): 64-byte reads
    const size_t total = (size_t)(N / 16) * nrb * NS * 256;
    const int F = 32 / bits;
    const uint32_t fm = (1u << bits) - 1u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), l = (int)((i >> 2) & 63);
        size_t b = i >> 8;
        const int set = (int)(b % NS); b /= NS;
        const int rb = (int)(b % nrb);
        const int stripe = (int)(b / nrb);
        const int row = rb * 16 + 4 * (l >> 4) + j, col = 16 * stripe + (l & 15);
        uint32_t w;
        if (perm) {
            w = 0;
            for (int f = 0; f < F; f++) {
                const int k = perm[row * F + f];
                w |= (((set ? qw1 : qw0)[(size_t)(k / F) * N + col] >> (bits * (k % F))) & fm) << (bits * f);
            }
        } else {
            w = (set ? qw1 : qw0)[(size_t)row * N + col];
        }
        uint32_t o = 0;
        for (int p = 0; p < F; p++) o |= ((w >> (bits * stripe_k_of_pos(p, F))) & fm) << (bits * p);
        R[i] = o;
    }
}

// 3-bit: a lane's 32 k = three words.  Word j holds k = 10 j .. 10 j + 9 as five 3-bit fields per half-word (even k low, odd k
// high: pair p = bits [3p+2:3p] of both halves) and bit j of k 30 / k 31 in its spare bits 15 / 31.  Source: the reference's
// 96-bit blocks (rows 3 b .. 3 b + 2 of qweight hold the 32 k of block b, fields straddling the word boundaries).
GPTQ_DEV void put3(uint32_t (&c3)[3], int j, uint32_t v) {
    const int bit = 3 * j, wi = bit >> 5, off = bit & 31;
    c3[wi] |= v << off;
    if (off > 29) c3[wi + 1] |= v >> (32 - off);
}
GPTQ_DEV uint32_t field3_at(const uint32_t *__restrict__ qw, int N, int col, int k) {   // 3-bit field k of one column in the 96-bit block layout
    const int bit = 3 * (k & 31), wi = bit >> 5, off = bit & 31;
    const uint32_t *src = qw + ((size_t)(k >> 5) * 3 + wi) * N + col;
    uint32_t v = src[0] >> off;
    if (off > 29) v |= src[N] << (32 - off);
    return v & 7u;
}
// perm != NULL: the group-sorted rows of an act-order layer, gathered field by field (k' = 32 blk + j comes from checkpoint k = perm[k'])
__global__ void __launch_bounds__(256) stripe_repack3_kernel(const uint32_t *__restrict__ qw0, const uint32_t *__restrict__ qw1,
                                                             uint32_t *__restrict__ R, int N, int nrb, int NS, const int32_t *__restrict__ perm) {
    const size_t total = (size_t)(N / 16) * nrb * NS * 64;   // one thread per (stripe, row block, set, lane): three output words
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63);
        size_t b = i >> 6;
        const int set = (int)(b % NS); b /= NS;
        const int rb = (int)(b % nrb);
        const int stripe = (int)(b / nrb);
        const int blk = rb * 4 + (l >> 4), col = 16 * stripe + (l & 15);
        const uint32_t *qw = set ? qw1 : qw0;
        uint32_t c3[3];
        if (perm) {
            c3[0] = c3[1] = c3[2] = 0u;
            for (int j = 0; j < 32; j++) put3(c3, j, field3_at(qw, N, col, perm[blk * 32 + j]));
        } else {
            const uint32_t *src = qw + (size_t)blk * 3 * N + col;
            c3[0] = src[0], c3[1] = src[N], c3[2] = src[2 * (size_t)N];
        }
        const uint32_t q30 = (uint32_t)field_of_block<3>(c3, 30), q31 = (uint32_t)field_of_block<3>(c3, 31);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            uint32_t o = (((q30 >> j) & 1u) << 15) | (((q31 >> j) & 1u) << 31);
#pragma unroll
            for (int pp = 0; pp < 5; pp++) {
                o |= (uint32_t)field_of_block<3>(c3, 10 * j + 2 * pp) << (3 * pp);
                o |= (uint32_t)field_of_block<3>(c3, 10 * j + 2 * pp + 1) << (16 + 3 * pp);
            }
            R[i * 3 + j] = o;
        }
    }
}

template <int BITS>
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
        const int z = zero_of<BITS>((set ? qz1 : qz0) + (size_t)g * (N / 32 * BITS), n);   // stored + 1, not re-masked (quant_linear.py:120-121)
        const half2_t e = {s, (half_t)(float)z};
        tab[i] = as_u32(e);
    }
}

// ---- the inverse (bits 2 / 4 / 8): checkpoint buffers of ONE weight set back out of an image.  The image is a bijection of
// (qweight, scales, qzeros) -- oracle.stripe16_unpack states it, tests hold both directions bit-exact -- so a model that has
// released its checkpoint buffers (GPTQ_RELEASE_CHECKPOINT, DESIGN.md "memory") can still produce its state_dict and feed the
// per-call dequantise pass of the prefill route.
// invperm != NULL: the image holds the group-sorted rows (see stripe_repack_kernel); field f of checkpoint row r is value k' = invperm[F r + f]
// of the sorted matrix, i.e. field position pos(k' % F) of the image word of packed row k' / F
__global__ void __launch_bounds__(256) stripe_unpack_kernel(const uint32_t *__restrict__ R, uint32_t *__restrict__ qw, int N, int nrb, int NS, int set,
                                                            int bits, size_t total, const int32_t *__restrict__ invperm) {
    const int F = 32 / bits;
    const uint32_t fm = (1u << bits) - 1u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % N), row = (int)(i / N);
        const int stripe = col >> 4, c = col & 15;
        auto image_word = [&](int srow) {
            const int rb = srow >> 4, rr = srow & 15;
            const int l = (rr >> 2) * 16 + c, j = rr & 3;
            return R[((((size_t)stripe * nrb + rb) * NS + set) * 64 + l) * 4 + j];
        };
        uint32_t w = 0;
        if (invperm) {
            for (int f = 0; f < F; f++) {
                const int ks = invperm[row * F + f], kk = ks % F;
                const int pos = (kk & 1) ? (kk - 1) / 2 + F / 2 : kk / 2;        // inverse of stripe_k_of_pos
                w |= ((image_word(ks / F) >> (bits * pos)) & fm) << (bits * f);
            }
        } else {
            const uint32_t o = image_word(row);
            for (int p = 0; p < F; p++) w |= ((o >> (bits * p)) & fm) << (bits * stripe_k_of_pos(p, F));
        }
        qw[i] = w;
    }
}

template <int BITS>
__global__ void __launch_bounds__(256) stripe_untable_kernel(const uint32_t *__restrict__ tab, half_t *__restrict__ sc, uint32_t *__restrict__ qz, int N,
                                                             int G, int NS, int set) {
    constexpr int F = 32 / BITS;
    const size_t total = (size_t)G * (N / F);   // one thread per qzeros word = F columns of one group
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int wcol = (int)(i % (N / F)), g = (int)(i / (N / F));
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < F; j++) {
            const int n = wcol * F + j;
            const half2_t e = as_half2(tab[(((size_t)(n >> 4) * NS + set) * G + g) * 16 + (n & 15)]);
            sc[(size_t)g * N + n] = e[0];
            const uint32_t stored = ((uint32_t)(int)(float)e[1] - 1u) & ((1u << BITS) - 1u);   // the table holds zero + 1 (not re-masked: 16 for a stored 15)
            word |= stored << (BITS * j);
        }
        qz[i] = word;
    }
}

// ---- 3-bit inverse (round 4): the lane's three image words of a 32-k block -> its 32 fields -> the reference's 96-bit block (rows 3 b .. 3 b + 2
// of qweight, value j at bits [3 j, 3 j + 3) of the little-endian stream); the zero stream along N likewise
GPTQ_DEV uint32_t image3_field(const uint32_t *__restrict__ o, int j) {   // field j (0..31) of a lane's three image words
    if (j >= 30) {
        const int sh = j == 30 ? 15 : 31;
        return ((o[0] >> sh) & 1u) | (((o[1] >> sh) & 1u) << 1) | (((o[2] >> sh) & 1u) << 2);
    }
    const int p = j % 10;
    return (o[j / 10] >> (3 * (p >> 1) + 16 * (p & 1))) & 7u;
}
// invperm != NULL: the image holds the group-sorted rows; checkpoint k sits at sorted position invperm[k]
__global__ void __launch_bounds__(256) stripe_unpack3_kernel(const uint32_t *__restrict__ R, uint32_t *__restrict__ qw, int N, int nrb, int NS, int set,
                                                             size_t total, const int32_t *__restrict__ invperm) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % N), blk = (int)(i / N);
        const int stripe = col >> 4, c = col & 15;
        uint32_t c3[3] = {0u, 0u, 0u};
        for (int j = 0; j < 32; j++) {
            const int k = invperm ? invperm[blk * 32 + j] : blk * 32 + j;
            const int b = k >> 5, l = (b & 3) * 16 + c;
            put3(c3, j, image3_field(R + ((((size_t)stripe * nrb + (b >> 2)) * NS + set) * 64 + l) * 3, k & 31));
        }
#pragma unroll
        for (int j = 0; j < 3; j++) qw[((size_t)blk * 3 + j) * N + col] = c3[j];
    }
}
__global__ void __launch_bounds__(256) stripe_untable3_kernel(const uint32_t *__restrict__ tab, half_t *__restrict__ sc, uint32_t *__restrict__ qz, int N,
                                                              int G, int NS, int set) {
    const size_t total = (size_t)G * (N / 32);   // one thread per 96-bit block of the zero stream = 32 columns of one group
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int wb = (int)(i % (N / 32)), g = (int)(i / (N / 32));
        uint32_t c3[3] = {0u, 0u, 0u};
        for (int j = 0; j < 32; j++) {
            const int n = wb * 32 + j;
            const half2_t e = as_half2(tab[(((size_t)(n >> 4) * NS + set) * G + g) * 16 + (n & 15)]);
            sc[(size_t)g * N + n] = e[0];
            put3(c3, j, ((uint32_t)(int)(float)e[1] - 1u) & 7u);   // the table holds zero + 1 (not re-masked: 8 for a stored 7)
        }
        for (int j = 0; j < 3; j++) qz[((size_t)g * (N / 32) + wb) * 3 + j] = c3[j];
    }
}

// ---- image -> dense fp16 W^T (round 5): the operand of the prefill tile GEMM straight from the stripe16 image ----
// A released layer (memory mode: the image is the only copy) used to rebuild the checkpoint layout per call (stripe_unpack_kernel) and then
// dequantise THAT (dequant_t_kernel): two passes over the weights in front of every prompt.  One pass: thread = (stripe, row block, set, lane)
// owns the lane's block of LK consecutive k of ONE column -- exactly its WPL image words -- and writes Wt[n][k .. k + LK) (2 LK bytes, the four
// lanes of a column 8 LK bytes contiguous).  The arithmetic is dequant_t_kernel's, element for element: fp16(q - (zero + 1)) * fp16 scale, one
// rounding (reference quant_linear.py:128) -- bit-identical weights, hence bit-identical products.  Trivial g_idx (the image of an act-order
// layer holds sorted rows: its W^T would be a 2-byte scatter; those layers keep the two-pass route).
// Workgroup = one stripe (16 columns) x 4 consecutive row blocks x one set; wave = row block.  The 16 x (4 BK) tile goes through LDS so that it
// leaves as whole rows: 8 BK bytes contiguous per column (1 KiB for 4 bits) instead of 2 LK-byte pieces per lane (the first version: 64-byte
// stores, +7 .. 14 % on a 4096 x 4096 prompt against the two-copy route; gpurun_out r5f).
template <int BITS>
__global__ void __launch_bounds__(256) stripe_dequant_t_kernel(const uint32_t *__restrict__ R, const uint32_t *__restrict__ tab, half_t *__restrict__ out,
                                                               int64_t ldo, int K, int N, int nrb, int NS, int G, int groupsize) {
    constexpr int F = BITS == 3 ? 1 : 32 / BITS, WPLv = BITS == 3 ? 3 : 4, LKv = stripe_lk(BITS), BKv = 4 * LKv, TK = 4 * BKv, PITCH = TK + 8;
    __shared__ __attribute__((aligned(16))) half_t tile[16 * PITCH];
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    const int nq = (nrb + 3) / 4;                                 // groups of four row blocks
    const size_t ntiles = (size_t)(N / 16) * nq * NS;
    for (size_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        size_t b = tix;
        const int set = (int)(b % NS); b /= NS;
        const int q4 = (int)(b % nq);
        const int stripe = (int)(b / nq);
        const int rb = q4 * 4 + wv;
        const int c = l & 15, kl = wv * BKv + (l >> 4) * LKv;      // k inside the tile
        if (rb < nrb) {
            const size_t i = (((size_t)stripe * nrb + rb) * NS + set) * 64 + l;
            uint32_t w[WPLv];
#pragma unroll
            for (int j = 0; j < WPLv; j++) w[j] = R[i * WPLv + j];
            const int k0 = q4 * TK + kl;
            const int g = groupsize >= K ? 0 : k0 / groupsize;
            const half2_t e = as_half2(tab[(((size_t)stripe * NS + set) * G + g) * 16 + c]);
#pragma unroll
            for (int j8 = 0; j8 < LKv / 8; j8++) {
                half8_t v;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int kk = 8 * j8 + t;                       // k inside the lane block
                    uint32_t q;
                    if constexpr (BITS == 3) {
                        q = image3_field(w, kk);
                    } else {
                        const int word = kk / F, f = kk % F;          // packed row of the lane, field (natural k order) inside it
                        const int pos = (f & 1) ? (f - 1) / 2 + F / 2 : f / 2;   // inverse of stripe_k_of_pos
                        q = (w[word] >> (BITS * pos)) & ((1u << BITS) - 1u);
                    }
                    v[t] = (half_t)((half_t)(float)q - e[1]) * e[0];
                }
                *(half8_t *)(tile + c * PITCH + kl + 8 * j8) = v;
            }
        }
        __syncthreads();
        // 16 rows x TK / 8 pieces of 16 bytes: consecutive threads -> consecutive pieces of one row
        constexpr int PPR = TK / 8;
        for (int p = tid; p < 16 * PPR; p += 256) {
            const int r = p / PPR, pc = p % PPR, k = q4 * TK + pc * 8;
            if (k < K) *(half8_t *)(out + ((size_t)set * N + 16 * stripe + r) * ldo + k) = *(const half8_t *)(tile + r * PITCH + pc * 8);
        }
        __syncthreads();
    }
}

}  // namespace

int stripe_unpack_launch(const void *image, int K, int N, int bits, int groupsize, int nsets, int set, uint32_t *qw, half_t *sc, int32_t *qz,
                         hipStream_t s, const int32_t *invperm) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_VARIANT;
    if (stripe_gq_shift(K, N, bits, groupsize) == -2 || set < 0 || set >= nsets || N % 32 != 0) return GPTQ_E_VARIANT;
    const int G = groupsize >= K ? 1 : K / groupsize;
    const uint32_t *R = (const uint32_t *)image;
    const uint32_t *tab = (const uint32_t *)((const char *)image + stripe_tab_offset(K, N, bits, nsets));
    if (bits == 3) {
        hipLaunchKernelGGL(stripe_unpack3_kernel, dim3(2048), dim3(256), 0, s, R, qw, N, K / 128, nsets, set, (size_t)(K / 32) * N, invperm);
        hipLaunchKernelGGL(stripe_untable3_kernel, dim3(512), dim3(256), 0, s, tab, sc, (uint32_t *)qz, N, G, nsets, set);
        return (int)hipGetLastError();
    }
    const size_t words = (size_t)(K / 32 * bits) * N;
    hipLaunchKernelGGL(stripe_unpack_kernel, dim3(2048), dim3(256), 0, s, R, qw, N, K / (16 * (32 / bits)), nsets, set, bits, words, invperm);
    if (bits == 2) hipLaunchKernelGGL(stripe_untable_kernel<2>, dim3(512), dim3(256), 0, s, tab, sc, (uint32_t *)qz, N, G, nsets, set);
    else if (bits == 4) hipLaunchKernelGGL(stripe_untable_kernel<4>, dim3(512), dim3(256), 0, s, tab, sc, (uint32_t *)qz, N, G, nsets, set);
    else hipLaunchKernelGGL(stripe_untable_kernel<8>, dim3(512), dim3(256), 0, s, tab, sc, (uint32_t *)qz, N, G, nsets, set);
    return (int)hipGetLastError();
}

// Wt[set * N + n][k] (row stride ldo) = the dense fp16 weight of every set of the image, k contiguous
int stripe_dequant_t_launch(const void *image, int K, int N, int bits, int groupsize, int nsets, half_t *out, int64_t ldo, hipStream_t s) {
    if (stripe_gq_shift(K, N, bits, groupsize) == -2 || nsets < 1 || nsets > 2) return GPTQ_E_VARIANT;
    if (ldo % 8 != 0 || ((uintptr_t)out % 16) != 0 || ldo < K) return GPTQ_E_ALIGN;
    const int G = groupsize >= K ? 1 : K / groupsize, nrb = K / (4 * stripe_lk(bits));
    const uint32_t *R = (const uint32_t *)image;
    const uint32_t *tab = (const uint32_t *)((const char *)image + stripe_tab_offset(K, N, bits, nsets));
    const size_t ntiles = (size_t)(N / 16) * ((nrb + 3) / 4) * nsets;
    const int grid = (int)std::min<size_t>(ntiles, 8192);
    switch (bits) {
        case 2: hipLaunchKernelGGL(stripe_dequant_t_kernel<2>, dim3(grid), dim3(256), 0, s, R, tab, out, ldo, K, N, nrb, nsets, G, groupsize); break;
        case 3: hipLaunchKernelGGL(stripe_dequant_t_kernel<3>, dim3(grid), dim3(256), 0, s, R, tab, out, ldo, K, N, nrb, nsets, G, groupsize); break;
        case 4: hipLaunchKernelGGL(stripe_dequant_t_kernel<4>, dim3(grid), dim3(256), 0, s, R, tab, out, ldo, K, N, nrb, nsets, G, groupsize); break;
        default: hipLaunchKernelGGL(stripe_dequant_t_kernel<8>, dim3(grid), dim3(256), 0, s, R, tab, out, ldo, K, N, nrb, nsets, G, groupsize); break;
    }
    return (int)hipGetLastError();
}

// groupsize is the effective one (K for the reference's -1).  Returns log2(groupsize / (4 KPW)), -1 for one group, -2 if ineligible.
int stripe_gq_shift(int K, int N, int bits, int groupsize) {
    if ((bits != 2 && bits != 3 && bits != 4 && bits != 8) || K <= 0 || N <= 0 || N % 16 != 0) return -2;
    const int lk = stripe_lk(bits), blk = 4 * lk;
    if (K % blk != 0) return -2;
    if ((K / blk + STRIPE_NW - 1) / STRIPE_NW > stripe_max_nu(bits)) return -2;
    if (groupsize >= K) return -1;
    if (groupsize < lk || groupsize % lk != 0 || K % groupsize != 0) return -2;
    const int q = groupsize / lk;
    for (int sft = 0; sft < 16; sft++)
        if ((1 << sft) == q) return sft;
    return -2;
}

size_t stripe_tab_offset(int K, int N, int bits, int nsets) { return (size_t)(K / 32 * bits) * N * nsets * 4; }

size_t stripe_total_bytes(int K, int N, int bits, int groupsize, int nsets) {
    if (stripe_gq_shift(K, N, bits, groupsize) == -2 || nsets < 1 || nsets > 2) return 0;
    const int G = groupsize >= K ? 1 : K / groupsize;
    return stripe_tab_offset(K, N, bits, nsets) + (size_t)nsets * G * N * 4;
}

int stripe_repack_launch(const uint32_t *qw0, const half_t *sc0, const int32_t *qz0, const uint32_t *qw1, const half_t *sc1, const int32_t *qz1,
                         void *out, int K, int N, int bits, int groupsize, hipStream_t s, const int32_t *perm) {
    const int NS = qw1 ? 2 : 1;
    const int G = groupsize >= K ? 1 : K / groupsize;
    uint32_t *R = (uint32_t *)out;
    uint32_t *tab = (uint32_t *)((char *)out + stripe_tab_offset(K, N, bits, NS));
    if (bits == 3) hipLaunchKernelGGL(stripe_repack3_kernel, dim3(2048), dim3(256), 0, s, qw0, qw1, R, N, K / 128, NS, perm);
    else hipLaunchKernelGGL(stripe_repack_kernel, dim3(2048), dim3(256), 0, s, qw0, qw1, R, N, K / (16 * (32 / bits)), NS, bits, perm);
    if (bits == 3) hipLaunchKernelGGL(stripe_table_kernel<3>, dim3(512), dim3(256), 0, s, sc0, qz0, sc1, qz1, tab, N, G, NS);
    else if (bits == 2) hipLaunchKernelGGL(stripe_table_kernel<2>, dim3(512), dim3(256), 0, s, sc0, qz0, sc1, qz1, tab, N, G, NS);
    else if (bits == 4) hipLaunchKernelGGL(stripe_table_kernel<4>, dim3(512), dim3(256), 0, s, sc0, qz0, sc1, qz1, tab, N, G, NS);
    else hipLaunchKernelGGL(stripe_table_kernel<8>, dim3(512), dim3(256), 0, s, sc0, qz0, sc1, qz1, tab, N, G, NS);
    return (int)hipGetLastError();
}

int stripe_gemv_dispatch(const StripeParams &p, hipStream_t s) {
    switch (p.bits) {
        case 2: return stripe_gemv_dispatch_b2(p, s);
        case 3: return stripe_gemv_dispatch_b3(p, s);
        case 4: return stripe_gemv_dispatch_b4(p, s);
        case 8: return stripe_gemv_dispatch_b8(p, s);
    }
    return GPTQ_E_VARIANT;
}

}  // namespace gptq
