// gemv.hip -- batch-1 (M <= 4) dequant-matvec for gfx950: the decode hot path behind
// QuantLinear.forward (reference quant/quant_linear.py:72-137, 263-269, 373-377) and the fused
// gate/up + SiLU*mul of QuantLlamaMLP (reference quant/fused_mlp.py:84-168).
//
// Design (HBM-bound; see DESIGN.md "GEMV"):
//  * a workgroup owns a tile of 4*NL columns; lanes are NL column-lanes x (64/NL) k-lanes per
//    wave; every lane streams `global_load_dwordx4` = 4 columns x one packed row, U*CH rows in
//    flight, straight to VGPRs (no LDS round trip for the weights -- they are read once);
//  * x is staged once per workgroup in LDS in the order the magic-exponent unpack produces
//    field pairs, so a word costs 3 shifts + 4 v_and_or + 4 v_dot2c_f32_f16 (4-bit);
//  * with the trivial group map the scale and zero are applied once per 32-k chunk:
//    y += s * (sum_k x_k (OFF+q_k) - (OFF+z) * sum_k x_k), all in fp32;
//  * k-lanes are reduced with wavefront __shfl_xor, waves through LDS, and (optional) K-slices
//    of different workgroups through fp32 atomics + an arrival ticket; the last arriver
//    swaps the workspace back to zero while reading it.
//  * act-order / odd group sizes / 3-bit go through gemv_generic_kernel, which keeps a
//    per-tile {scale, zero} table in LDS indexed by g_idx[k].
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

// ---------------------------------------------------------------------------------------
// fast path: trivial g_idx, groupsize % 32 == 0, bits in {2,4,8}
// ---------------------------------------------------------------------------------------
// Loads in flight per lane are bounded by registers: U 32-k chunks per pipeline stage, and a
// second (prefetch) stage only where the VGPR budget of the variant allows it.
template <int BITS, int WAVES, bool FUSED2>
constexpr int gemv_u() {
    return (BITS == 8 || FUSED2 || WAVES >= 16) ? 1 : 2;
}
template <int BITS, int WAVES, int MR, bool FUSED2>
constexpr bool gemv_double_buffer() {
    if (WAVES >= 16) return false;            // 128-VGPR budget at 1024 threads
    if (BITS == 8 && FUSED2) return false;    // 16 dwordx4 per stage already
    if (MR == 4 && (FUSED2 || BITS == 8)) return false;
    return true;
}

template <int BITS, int CH, int U, bool FUSED2>
struct Batch {
    u32x4 w[FUSED2 ? 2 : 1][U][CH];
    half4_t s[FUSED2 ? 2 : 1][U];
    uint32_t zw[FUSED2 ? 2 : 1][U];
};

template <int BITS, int NL, int WAVES, int MR, bool FUSED2>
__global__ void __launch_bounds__(WAVES * 64) gemv_fast_kernel(const GemvParams p) {
    using UP = Unpack<BITS>;
    constexpr int KPW = UP::KPW, NP = UP::NP;
    constexpr int CH = 32 / KPW;  // packed rows per 32-k chunk
    constexpr int KLW = 64 / NL, KL = WAVES * KLW, T = WAVES * 64;
    constexpr int NS = FUSED2 ? 2 : 1;
    constexpr int TILE = 4 * NL;
    constexpr int U = gemv_u<BITS, WAVES, FUSED2>();
    constexpr bool DB = gemv_double_buffer<BITS, WAVES, MR, FUSED2>();
    using BatchT = Batch<BITS, CH, U, FUSED2>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid / p.split_k, slice = bid % p.split_k;
    const int chunk_begin = slice * p.chunks_per_slice;
    const int chunk_end = min(p.nchunks, chunk_begin + p.chunks_per_slice);
    const int cg = lane % NL, kl = wave * KLW + lane / NL;
    const int n0 = tile * TILE + 4 * cg;
    const bool active = n0 < p.N;
    const int N = p.N;
    const int ldz = N / KPW;
    const int zshift0 = BITS * (n0 % KPW);

    // ---- issue the x loads for staging first (oldest in the vmcnt queue) ---------------------
    const int nk = (chunk_end - chunk_begin) * 32;
    half_t *lx = (half_t *)smem;  // [MR][nk], each word-row permuted into pair order
    constexpr int XV = (KPW * 2 >= 16) ? 8 : 4;  // halves per staging load (16 B or 8 B)
    typedef half_t xvec_t __attribute__((ext_vector_type(XV)));
    const int nxv = MR * nk / XV;
    constexpr int XPT = 2;  // staging loads kept in flight per thread per round

    // ---- first weight batch -------------------------------------------------------------
    auto load_batch = [&](BatchT &b, int c0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = c0 + u * KL;
            if (active && c < chunk_end) {
                const int g = (c * 32) / p.groupsize;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const uint32_t *qw = p.qw[s] + (size_t)c * CH * N + n0;
#pragma unroll
                    for (int i = 0; i < CH; i++)
                        b.w[s][u][i] = __builtin_nontemporal_load((const u32x4 *)(qw + (size_t)i * N));
                    b.s[s][u] = *(const half4_t *)(p.sc[s] + (size_t)g * N + n0);
                    b.zw[s][u] = (uint32_t)p.qz[s][(size_t)g * ldz + n0 / KPW];
                }
            }
        }
    };

    float y[NS][MR][4];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int m = 0; m < MR; m++)
#pragma unroll
            for (int j = 0; j < 4; j++) y[s][m][j] = 0.f;

    auto compute_batch = [&](const BatchT &b, int c0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = c0 + u * KL;
            if (active && c < chunk_end) {
                float acc[NS][MR][4];
                float xs[MR];
#pragma unroll
                for (int m = 0; m < MR; m++) {
                    xs[m] = 0.f;
#pragma unroll
                    for (int s = 0; s < NS; s++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[s][m][j] = 0.f;
                }
                const int row0 = (c - chunk_begin) * CH;
#pragma unroll
                for (int i = 0; i < CH; i++) {
                    half2_t X[MR][NP];
#pragma unroll
                    for (int m = 0; m < MR; m++) {
                        const half2_t *px = (const half2_t *)(lx + (size_t)m * nk + (size_t)(row0 + i) * KPW);
#pragma unroll
                        for (int q = 0; q < NP; q++) X[m][q] = px[q];
                        const half2_t ones = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
                        for (int q = 0; q < NP; q++) xs[m] = __builtin_amdgcn_fdot2(X[m][q], ones, xs[m], false);
                    }
#pragma unroll
                    for (int s = 0; s < NS; s++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            half2_t t[NP];
                            UP::pairs(b.w[s][u][i][j], t);
#pragma unroll
                            for (int m = 0; m < MR; m++)
#pragma unroll
                                for (int q = 0; q < NP; q++)
                                    acc[s][m][j] = __builtin_amdgcn_fdot2(t[q], X[m][q], acc[s][m][j], false);
                        }
                }
#pragma unroll
                for (int s = 0; s < NS; s++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float zf = (float)(((b.zw[s][u] >> (zshift0 + BITS * j)) & ((1u << BITS) - 1u)) + 1u) + UP::OFF;
                        const float sf = (float)b.s[s][u][j];
#pragma unroll
                        for (int m = 0; m < MR; m++) y[s][m][j] += sf * (acc[s][m][j] - zf * xs[m]);
                    }
            }
        }
    };

    // x loads -> registers (issued before the weights so their wait leaves the weights in flight)
    xvec_t xr[XPT];
    int xidx[XPT];
#pragma unroll
    for (int r = 0; r < XPT; r++) {
        xidx[r] = tid + r * T;
        if (xidx[r] < nxv) {
            const int m = xidx[r] / (nk / XV), e = (xidx[r] % (nk / XV)) * XV;
            if (m < p.M)
                xr[r] = *(const xvec_t *)(p.x + (size_t)m * p.ldx + (size_t)chunk_begin * 32 + e);
            else
                xr[r] = (xvec_t)(half_t)0;
        }
    }

    BatchT cur;
    int c0 = chunk_begin + kl;
    load_batch(cur, c0);

    // permute + write the staged x.  One word-row = KPW halves; staged position of field f is
    // staged_pos<BITS>(f).
    auto stage_write = [&](const xvec_t &v, int idx) {
        const int m = idx / (nk / XV), e = (idx % (nk / XV)) * XV;
        half_t *dst = lx + (size_t)m * nk;
        if constexpr (XV >= KPW) {
            // vector covers XV/KPW whole word-rows
            xvec_t o;
#pragma unroll
            for (int q = 0; q < XV; q++) {
                const int wr = q / KPW, f = q % KPW;
                o[wr * KPW + staged_pos<BITS>(f)] = v[q];
            }
            *(xvec_t *)(dst + e) = o;
        } else {
            // word-row spans several vectors (2-bit: 16 halves = 2 x 8): scatter element-wise
            const int wr0 = e / KPW * KPW, f0 = e % KPW;
#pragma unroll
            for (int q = 0; q < XV; q++) dst[wr0 + staged_pos<BITS>(f0 + q)] = v[q];
        }
    };
#pragma unroll
    for (int r = 0; r < XPT; r++)
        if (xidx[r] < nxv) stage_write(xr[r], xidx[r]);
    for (int idx = tid + XPT * T; idx < nxv; idx += T) {
        const int m = idx / (nk / XV), e = (idx % (nk / XV)) * XV;
        xvec_t v = (xvec_t)(half_t)0;
        if (m < p.M) v = *(const xvec_t *)(p.x + (size_t)m * p.ldx + (size_t)chunk_begin * 32 + e);
        stage_write(v, idx);
    }
    __syncthreads();

    // ---- main loop, register double-buffered -------------------------------------------------
    if constexpr (DB) {
        BatchT nxt;
        while (true) {
            const int cn = c0 + U * KL;
            const bool more = cn < chunk_end;  // per lane; no barrier inside the loop
            if (more) load_batch(nxt, cn);
            compute_batch(cur, c0);
            if (!more) break;
            cur = nxt;
            c0 = cn;
        }
    } else {
        while (true) {
            compute_batch(cur, c0);
            c0 += U * KL;
            if (c0 >= chunk_end) break;
            load_batch(cur, c0);
        }
    }

    // ---- reduce over k-lanes (wave shuffles), waves (LDS), K-slices (atomics) -----------------
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int m = 0; m < MR; m++)
#pragma unroll
            for (int j = 0; j < 4; j++) y[s][m][j] = wave_sum_xor(y[s][m][j], NL);

    __syncthreads();  // everyone is done reading staged x
    float *red = (float *)smem;  // [WAVES][NS][MR][TILE]
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
    float tot[NS][(NOUT + T - 1) / T];
#pragma unroll
    for (int r = 0; r < (NOUT + T - 1) / T; r++) {
        const int e = tid + r * T;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            float a = 0.f;
            if (e < NOUT) {
#pragma unroll
                for (int w = 0; w < WAVES; w++) a += red[((size_t)w * NS + s) * NOUT + e];
            }
            tot[s][r] = a;
        }
    }

    // ---- outputs: direct, or combined over K-slices in one atomic round trip -------------------
#pragma unroll
    for (int r = 0; r < (NOUT + T - 1) / T; r++) {
        const int e = tid + r * T;
        const int m = e / TILE, n = tile * TILE + e % TILE;
        if (e < NOUT && m < p.M && n < N) {
            float t0 = tot[0][r], t1 = 0.f;
            if constexpr (FUSED2) t1 = tot[1][r];
            bool mine = true;
            if (p.split_k > 1) {
                u64_t *word = p.ws + (size_t)m * N + n;
                if constexpr (FUSED2) mine = splitk_add2(word, t0, t1, p.split_k, t0, t1);
                else mine = splitk_add1(word, t0, p.split_k, t0);
            }
            if (mine) {
                float v = t0;
                if constexpr (FUSED2) v = t0 * (1.0f / (1.0f + __expf(-t0))) * t1;  // silu on the fp32 accumulator
                half_t h = (half_t)v;
                if (p.bias) h = (half_t)((float)h + (float)p.bias[n]);
                p.y[(size_t)m * p.ldy + n] = h;
            }
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
            half_t h = (half_t)v;
            if (p.bias) h = (half_t)((float)h + (float)p.bias[n]);
            p.y[(size_t)m * p.ldy + n] = h;
        }
    }
}

// ---------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------
template <int BITS, int NL, int WAVES, int MR, bool FUSED2>
static int launch_fast(const GemvParams &p, hipStream_t stream) {
    constexpr int NS = FUSED2 ? 2 : 1;
    constexpr int TILE = 4 * NL;
    const size_t x_bytes = (size_t)MR * p.chunks_per_slice * 32 * 2;
    const size_t red_bytes = (size_t)WAVES * NS * MR * TILE * 4;
    const size_t lds = (((x_bytes > red_bytes ? x_bytes : red_bytes) + 15) & ~(size_t)15) + 16;
    if (lds > 160 * 1024 - 64) return GPTQ_E_SHAPE;
    auto kern = gemv_fast_kernel<BITS, NL, WAVES, MR, FUSED2>;
    static size_t configured = 0;  // per instantiation
    if (lds > 48 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = lds;
    }
    dim3 grid(p.ntiles * p.split_k), block(WAVES * 64);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
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
    static size_t configured = 0;
    if (lds > 48 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = lds;
    }
    dim3 grid(p.ntiles), block(WAVES * 64);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

// Variant table for the fast path.  A variant fixes (NL, WAVES); MR and FUSED2 come from the
// call.  Index = variant id (stable: tests and the autotune warm-up refer to it).
const GemvVariant g_gemv_variants[GEMV_NUM_VARIANTS] = {
    {4, 4},   // 0: 16-col tiles (64-B row segments), 256 threads
    {4, 8},   // 1: 16-col tiles, 512 threads
    {4, 16},  // 2: 16-col tiles, 1024 threads
    {8, 4},   // 3: 32-col tiles (128-B segments), 256 threads
    {8, 8},   // 4
    {16, 4},  // 5: 64-col tiles (256-B segments), 256 threads
    {16, 8},  // 6
    {64, 4},  // 7: 256-col tiles (full 1-KiB rows per wave), 256 threads
    {32, 4},  // 8: 128-col tiles (512-B segments), 256 threads
    {32, 8},  // 9
    {64, 8},  // 10
    {16, 16}, // 11: 64-col tiles, 1024 threads
};

template <int BITS, int MR, bool FUSED2>
static int launch_fast_variant(int variant, const GemvParams &p, hipStream_t s) {
    switch (variant) {
        case 0: return launch_fast<BITS, 4, 4, MR, FUSED2>(p, s);
        case 1: return launch_fast<BITS, 4, 8, MR, FUSED2>(p, s);
        case 2:  // 1024 threads cap the kernel at 128 VGPRs: the two-set kernel does not fit
            if constexpr (FUSED2) return launch_fast<BITS, 4, 8, MR, FUSED2>(p, s);
            else return launch_fast<BITS, 4, 16, MR, FUSED2>(p, s);
        case 3: return launch_fast<BITS, 8, 4, MR, FUSED2>(p, s);
        case 4: return launch_fast<BITS, 8, 8, MR, FUSED2>(p, s);
        case 5: return launch_fast<BITS, 16, 4, MR, FUSED2>(p, s);
        case 6: return launch_fast<BITS, 16, 8, MR, FUSED2>(p, s);
        case 7: return launch_fast<BITS, 64, 4, MR, FUSED2>(p, s);
        case 8: return launch_fast<BITS, 32, 4, MR, FUSED2>(p, s);
        case 9: return launch_fast<BITS, 32, 8, MR, FUSED2>(p, s);
        case 10: return launch_fast<BITS, 64, 8, MR, FUSED2>(p, s);
        case 11:
            if constexpr (FUSED2) return launch_fast<BITS, 16, 8, MR, FUSED2>(p, s);
            else return launch_fast<BITS, 16, 16, MR, FUSED2>(p, s);
    }
    return GPTQ_E_VARIANT;
}

// the fused (two weight sets) kernel is instantiated for M <= 2 only; capi.hip splits larger M
template <int BITS, bool FUSED2>
static int launch_fast_m(int variant, const GemvParams &p, hipStream_t s) {
    if (p.M <= 1) return launch_fast_variant<BITS, 1, FUSED2>(variant, p, s);
    if (p.M <= 2) return launch_fast_variant<BITS, 2, FUSED2>(variant, p, s);
    if constexpr (FUSED2) {
        return GPTQ_E_VARIANT;
    } else {
        return launch_fast_variant<BITS, 4, FUSED2>(variant, p, s);
    }
}

int gemv_fast_dispatch(int bits, bool fused2, int variant, const GemvParams &p, hipStream_t s) {
    switch (bits) {
        case 2: return fused2 ? launch_fast_m<2, true>(variant, p, s) : launch_fast_m<2, false>(variant, p, s);
        case 4: return fused2 ? launch_fast_m<4, true>(variant, p, s) : launch_fast_m<4, false>(variant, p, s);
        case 8: return fused2 ? launch_fast_m<8, true>(variant, p, s) : launch_fast_m<8, false>(variant, p, s);
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
