// gemm_mfma.hip -- MFMA-tiled dequant-GEMM for batched prefill (M > 64) on gfx950:
// C[M,N] = A[M,K] . deq(B) (+ bias), the large-M regime of the reference's matmul_248_kernel
// (quant/quant_linear.py:72-137).  Bound: fp16 MFMA (2.5 PFLOP/s dense); flops = 2*M*N*K.
//
//  * workgroup tile 256(M) x 256(N) x 64(K), 4 waves in a 2 x 2 grid, each wave owns 128 x 128 =
//    4 x 4 tiles of v_mfma_f32_32x32x16_f16 (256 fp32 accumulators per lane; one wave per SIMD,
//    the register file is the occupancy limit by design);
//  * B is dequantised ONCE per (workgroup, K slab) on the way into LDS, with the reference's own
//    numerics -- fp16(q - z) exact (magic-exponent unpack + packed fp16 subtract), times the fp16
//    scale, one fp16 rounding (quant_linear.py:128) -- so the MFMA consumes exactly the weights
//    the reference's tl.dot does; a packed word (8 consecutive k of one column) IS one B fragment
//    (8 halves of one column per lane), so the LDS layout is [n][k8-block] rows of 16-byte
//    fragments (row stride 144 B: conflict-free ds_write_b128 and ds_read_b128);
//  * A goes global -> registers -> LDS in the same [m][k8-block] layout, 128-byte row segments
//    per 8 lanes; LDS is double buffered: one barrier per K slab;
//  * operands are swapped in the MFMA (D = Bfrag^T-major) so that each lane ends with 4
//    consecutive n of one m: the epilogue transposes through LDS and writes 16 B per lane,
//    256 contiguous bytes per row;
//  * workgroups are numbered so that all N tiles of an M tile run on the same XCD (block b lands
//    on XCD b % 8): the A slab is fetched from HBM once and shared through that XCD's L2.
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

struct GemmParams {
    const half_t *a;
    int64_t lda;
    const uint32_t *qw;
    const half_t *sc;
    const int32_t *qz;
    const half_t *bias;
    half_t *c;
    int64_t ldc;
    int M, K, N, groupsize;
    int ntm, ntn;  // tiles along M and N
};

constexpr int GM = 256, GN = 256, GK = 64;
constexpr int KB = GK / 8;                 // k8-blocks per slab
constexpr int ROWB = KB * 16 + 16;         // LDS row stride in bytes (144): 16-B pad kills bank conflicts
constexpr int TILE_BYTES = GM * ROWB;      // one A (or B) buffer: 36 864 B
constexpr int CROW = 128 * 2 + 16;         // epilogue row stride (bytes) of a wave's 128 x 128 fp16 block

// 8 consecutive k (one k8-block) of one column -> 8 halves with the reference's numerics.
//   words : the packed words covering the block (4-bit: 1, 8-bit: 2; 2-bit handled by the caller)
template <int BITS>
GPTQ_DEV half8_t dequant8(const uint32_t *w, half2_t zc, half2_t s2, uint32_t msk, uint32_t mag);

template <>
GPTQ_DEV half8_t dequant8<4>(const uint32_t *w, half2_t zc, half2_t s2, uint32_t msk, uint32_t mag) {
    half2_t t[4];
    Unpack<4>::pairs_rc(w[0], t, msk, mag);  // {OFF+q_i, OFF+q_{i+4}}
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = (t[i] - zc) * s2;  // exact subtract, one fp16 rounding
    return half8_t{t[0][0], t[1][0], t[2][0], t[3][0], t[0][1], t[1][1], t[2][1], t[3][1]};
}

template <>
GPTQ_DEV half8_t dequant8<8>(const uint32_t *w, half2_t zc, half2_t s2, uint32_t msk, uint32_t mag) {
    half2_t a[2], b[2];
    Unpack<8>::pairs_rc(w[0], a, msk, mag);  // bytes (0,2), (1,3)
    Unpack<8>::pairs_rc(w[1], b, msk, mag);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        a[i] = (a[i] - zc) * s2;
        b[i] = (b[i] - zc) * s2;
    }
    return half8_t{a[0][0], a[1][0], a[0][1], a[1][1], b[0][0], b[1][0], b[0][1], b[1][1]};
}

template <int BITS>
__global__ void __launch_bounds__(256) gemm_mfma_kernel(const GemmParams p) {
    using UP = Unpack<BITS>;
    constexpr int KPW = UP::KPW;
    constexpr int WPB = 8 / KPW;      // words per k8-block (4-bit: 1, 8-bit: 2)
    constexpr int NW = KB * WPB;      // words per thread per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *As = smem;                       // [2][GM][ROWB]
    char *Bs = smem + 2 * TILE_BYTES;      // [2][GN][ROWB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: XCD x (= block % 8) walks M tiles x, x+8, ... and all N tiles of each
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tm = (j / p.ntn) * 8 + xcd, tn = j % p.ntn;
    if (tm >= p.ntm) return;
    const int m0 = tm * GM, n0 = tn * GN;
    const int M = p.M, N = p.N, K = p.K;

    // ---- global -> register staging maps ------------------------------------------------------
    // A: chunk c = tid + 256*i (i < 8): row = c / 8, kb = c % 8 -> 8 lanes cover one 128-B row segment
    const half_t *aptr[8];
    int aoff[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = tid + 256 * i, row = c >> 3, kb = c & 7;
        const int m = min(m0 + row, M - 1);
        aptr[i] = p.a + (size_t)m * p.lda + kb * 8;
        aoff[i] = row * ROWB + kb * 16;
    }
    // B: thread = column n0 + tid, all KB blocks of the slab
    const int nb = min(n0 + tid, N - 1);
    const uint32_t *bptr = p.qw + nb;
    const int boff = tid * ROWB;
    const uint32_t MSK = sreg_const(UP::MSK_C), MAG = vreg_const(UP::MAG_C);
    const int ldz = N / KPW;

    u32x4 areg[8];
    uint32_t breg[NW];
    auto load_global = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; i++) areg[i] = *(const u32x4 *)(aptr[i] + k0);
#pragma unroll
        for (int w = 0; w < NW; w++) breg[w] = bptr[(size_t)(k0 / KPW + w) * N];
    };
    int g_cur = -1;
    half2_t zc = {(half_t)0, (half_t)0}, s2 = {(half_t)0, (half_t)0};
    auto store_lds = [&](int buf, int k0) {
        char *ad = As + buf * TILE_BYTES, *bd = Bs + buf * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 8; i++) *(u32x4 *)(ad + aoff[i]) = areg[i];
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            const int g = (k0 + kb * 8) / p.groupsize;  // uniform over the workgroup
            if (g != g_cur) {
                g_cur = g;
                const half_t s = p.sc[(size_t)g * N + nb];
                const float z = (float)zero_of<BITS>(p.qz + (size_t)g * ldz, nb) + UP::OFF;
                s2 = half2_t{s, s};
                zc = half2_t{(half_t)z, (half_t)z};
            }
            const half8_t v = dequant8<BITS>(&breg[kb * WPB], zc, s2, MSK, MAG);
            *(half8_t *)(bd + boff + kb * 16) = v;
        }
    };

    float16_t acc[4][4];  // [n tile][m tile] (operands swapped: rows of D are n)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) acc[i][jj] = (float16_t)0.f;

    const int nslab = K / GK;
    load_global(0);
    store_lds(0, 0);
    __syncthreads();

    // fragment read addresses: lane l -> row (l & 31) of the 32-row tile, k8-block (l >> 5) of the K16 step
    const int frow = lane & 31, fkb = lane >> 5;
    for (int it = 0; it < nslab; it++) {
        const int buf = it & 1;
        if (it + 1 < nslab) load_global((it + 1) * GK);
        const char *ab = As + buf * TILE_BYTES + (wm * 128 + frow) * ROWB + fkb * 16;
        const char *bb = Bs + buf * TILE_BYTES + (wn * 128 + frow) * ROWB + fkb * 16;
#pragma unroll
        for (int ks = 0; ks < GK / 16; ks++) {
            half8_t af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                af[t] = *(const half8_t *)(ab + t * 32 * ROWB + ks * 32);
                bf[t] = *(const half8_t *)(bb + t * 32 * ROWB + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[i], af[jj], acc[i][jj], 0, 0, 0);
        }
        if (it + 1 < nslab) store_lds(buf ^ 1, (it + 1) * GK);
        __syncthreads();
    }

    // ---- epilogue: fp32 -> fp16, transpose through LDS (wave-private region), 16-B row stores ----
    char *cs = smem + wave * (128 * CROW);
    const int ml = lane & 31, nq = (lane >> 5) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nl = i * 32 + 8 * r + nq;  // D row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
                const half4_t h = {(half_t)acc[i][jj][4 * r + 0], (half_t)acc[i][jj][4 * r + 1], (half_t)acc[i][jj][4 * r + 2],
                                   (half_t)acc[i][jj][4 * r + 3]};
                *(half4_t *)(cs + (jj * 32 + ml) * CROW + nl * 2) = h;
            }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the region is private to this wave
    __builtin_amdgcn_wave_barrier();
    const int c16 = lane & 15, rsub = lane >> 4;
    const int ncol = n0 + wn * 128 + c16 * 8;
#pragma unroll 4
    for (int r = 0; r < 32; r++) {
        const int mloc = r * 4 + rsub;
        const int m = m0 + wm * 128 + mloc;
        half8_t v = *(const half8_t *)(cs + mloc * CROW + c16 * 16);
        if (m < M && ncol < N) {
            if (p.bias) {
                const half8_t b = *(const half8_t *)(p.bias + ncol);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (half_t)((float)v[e] + (float)b[e]);
            }
            *(half8_t *)(p.c + (size_t)m * p.ldc + ncol) = v;
        }
    }
}

template <int BITS>
static int launch_gemm(const GemmParams &p, hipStream_t s) {
    auto kern = gemm_mfma_kernel<BITS>;
    const size_t lds = 4 * (size_t)TILE_BYTES;  // 147 456 B (>= 4 * 128 * CROW for the epilogue)
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int groups_of_8 = (p.ntm + 7) / 8;
    dim3 grid(groups_of_8 * 8 * p.ntn), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
    return (int)hipGetLastError();
}

// Eligibility: bits in {4, 8}, trivial g_idx (checked by the caller), K % 64 == 0, groupsize % 8 == 0,
// N % 8 == 0 (always: N % 32 == 0), rows 16-byte aligned.  Everything else -> GPTQ_E_VARIANT and
// the caller falls back to the weight-streaming kernel.
int gemm_dispatch(int bits, bool fused2, const GemvParams &q, hipStream_t s) {
    if (fused2) return GPTQ_E_VARIANT;
    if (bits != 4 && bits != 8) return GPTQ_E_VARIANT;
    if (q.K % GK != 0 || q.groupsize % 8 != 0 || q.ldx % 8 != 0 || q.ldy % 8 != 0) return GPTQ_E_VARIANT;
    if (((uintptr_t)q.y % 16) != 0 || (q.bias && ((uintptr_t)q.bias % 16) != 0)) return GPTQ_E_VARIANT;
    GemmParams p;
    p.a = q.x;
    p.lda = q.ldx;
    p.qw = q.qw[0];
    p.sc = q.sc[0];
    p.qz = q.qz[0];
    p.bias = q.bias;
    p.c = q.y;
    p.ldc = q.ldy;
    p.M = q.M;
    p.K = q.K;
    p.N = q.N;
    p.groupsize = q.groupsize;
    p.ntm = (q.M + GM - 1) / GM;
    p.ntn = (q.N + GN - 1) / GN;
    return bits == 4 ? launch_gemm<4>(p, s) : launch_gemm<8>(p, s);
}

}  // namespace gptq
