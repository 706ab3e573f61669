// skinny_mfma.hip -- weight-streaming dequant-matmul for small batches (M <= 64) on gfx950.
//
// Same operator as gemv.hip (reference quant/quant_linear.py:72-137; fused variant
// quant/fused_mlp.py:84-168) for the batch range where the weights are still read exactly once
// and HBM is the roofline, but the per-row VALU cost of the GEMV would dominate.  The k-lane
// reduction is done by the matrix core instead of shuffles:
//
//  * a wave owns 64 columns; lane l (cl = l & 15, kg = l >> 4) loads ONE dwordx4 per unit =
//    4 adjacent columns x the packed row of k-group kg, i.e. a wave instruction fetches
//    4 rows x 256 contiguous bytes;
//  * each loaded word is exactly the B fragment of v_mfma_f32_16x16x32_f16 for its column
//    (8 consecutive k of one column per lane): fields are expanded with the magic-exponent
//    trick and the zero point is removed EXACTLY in fp16 (v_pk_add_f16), so B holds the
//    integers (q - z); one MFMA per column set j multiplies them with the x fragment;
//  * the x fragment is loaded from global (L2-resident) in natural order and permuted in
//    registers to the field order the unpack produces (A and B only have to agree on k);
//  * accumulators are flushed through the fp32 scale once per quantisation group;
//  * waves split K inside the workgroup (LDS reduce); workgroups may split K further
//    (fp32 atomics + arrival ticket, identical to gemv.hip).
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

template <int BITS>
struct SkinnyGeom;
template <>
struct SkinnyGeom<4> {
    static constexpr int LROWS = 1, UK = 32, STEPS = 1;  // rows per lane, k per unit, MFMAs per column
};
template <>
struct SkinnyGeom<2> {
    static constexpr int LROWS = 1, UK = 64, STEPS = 2;
};
template <>
struct SkinnyGeom<8> {
    static constexpr int LROWS = 2, UK = 32, STEPS = 1;
};

// B fragments (integers q - z as fp16) of one column for one unit.
template <int BITS>
GPTQ_DEV void make_b(const uint32_t (&w)[SkinnyGeom<BITS>::LROWS], half2_t zneg, half8_t (&b)[SkinnyGeom<BITS>::STEPS]) {
    using UP = Unpack<BITS>;
    if constexpr (BITS == 4) {
        half2_t t[4];
        UP::pairs(w[0], t);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            t[q] += zneg;
            b[0][2 * q] = t[q][0];
            b[0][2 * q + 1] = t[q][1];
        }
    } else if constexpr (BITS == 2) {
        half2_t t[8];
        UP::pairs(w[0], t);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            t[q] += zneg;
            b[q / 4][2 * (q % 4)] = t[q][0];
            b[q / 4][2 * (q % 4) + 1] = t[q][1];
        }
    } else {
        half2_t t0[2], t1[2];
        UP::pairs(w[0], t0);
        UP::pairs(w[1], t1);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            t0[q] += zneg;
            t1[q] += zneg;
            b[0][2 * q] = t0[q][0];
            b[0][2 * q + 1] = t0[q][1];
            b[0][4 + 2 * q] = t1[q][0];
            b[0][4 + 2 * q + 1] = t1[q][1];
        }
    }
}

// A fragments: natural-order x (UK/4 halves per lane) -> the k order make_b produces.
template <int BITS>
GPTQ_DEV void make_a(const half8_t (&xin)[SkinnyGeom<BITS>::STEPS], half8_t (&a)[SkinnyGeom<BITS>::STEPS]) {
    if constexpr (BITS == 4) {
        // fields [0,4,1,5,2,6,3,7]
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a[0][2 * q] = xin[0][q];
            a[0][2 * q + 1] = xin[0][q + 4];
        }
    } else if constexpr (BITS == 2) {
        // pair q = fields (q, q+8); step q/4
#pragma unroll
        for (int q = 0; q < 8; q++) {
            a[q / 4][2 * (q % 4)] = xin[0][q];
            a[q / 4][2 * (q % 4) + 1] = xin[1][q];
        }
    } else {
        // two words of 4 bytes: [0,2,1,3 | 4,6,5,7]
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                a[0][4 * h + 2 * q] = xin[0][4 * h + q];
                a[0][4 * h + 2 * q + 1] = xin[0][4 * h + q + 2];
            }
    }
}

template <int BITS, int MT, bool FUSED2>
struct SkinnyStage {
    uint32_t w[FUSED2 ? 2 : 1][SkinnyGeom<BITS>::LROWS][4];
    half4_t s[FUSED2 ? 2 : 1];
    uint32_t zw[FUSED2 ? 2 : 1];
    half8_t x[MT][SkinnyGeom<BITS>::STEPS];
};

template <int BITS, int MT, int WAVES, bool FUSED2>
__global__ void __launch_bounds__(WAVES * 64) skinny_kernel(const GemvParams p) {
    using GEO = SkinnyGeom<BITS>;
    using UP = Unpack<BITS>;
    constexpr int KPW = UP::KPW, LROWS = GEO::LROWS, UK = GEO::UK, STEPS = GEO::STEPS;
    constexpr int NS = FUSED2 ? 2 : 1, T = WAVES * 64, TILE = 64;
    constexpr int ROWS_PER_UNIT = UK / KPW;  // packed rows covered by one unit (all 4 k-groups)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15, kg = lane >> 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid / p.split_k, slice = bid % p.split_k;
    const int N = p.N;
    const int n0 = tile * TILE + 4 * cl;
    const bool active = n0 < N;
    const int ldz = N / KPW;
    const int zshift0 = BITS * (n0 % KPW);

    // units of this slice, split contiguously over the waves
    const int nunits = p.K / UK;
    const int ups = (nunits + p.split_k - 1) / p.split_k;
    const int ub_s = slice * ups, ue_s = min(nunits, ub_s + ups);
    const int upw = (ue_s - ub_s + WAVES - 1) / WAVES;
    const int ub = ub_s + wave * upw, ue = min(ue_s, ub + upw);

    float4_t acc[NS][MT][4], yv[NS][MT][4];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[s][mt][j] = (float4_t)0.f;
                yv[s][mt][j] = (float4_t)0.f;
            }

    auto load_stage = [&](SkinnyStage<BITS, MT, FUSED2> &st, int u) {
        if (u < ue) {
            const int g = (u * UK) / p.groupsize;
            const int row = u * ROWS_PER_UNIT + kg * LROWS;
            if (active) {
#pragma unroll
                for (int s = 0; s < NS; s++) {
#pragma unroll
                    for (int r = 0; r < LROWS; r++) {
                        u32x4 v = __builtin_nontemporal_load((const u32x4 *)(p.qw[s] + (size_t)(row + r) * N + n0));
#pragma unroll
                        for (int j = 0; j < 4; j++) st.w[s][r][j] = v[j];
                    }
                    st.s[s] = *(const half4_t *)(p.sc[s] + (size_t)g * N + n0);
                    st.zw[s] = (uint32_t)p.qz[s][(size_t)g * ldz + n0 / KPW];
                }
            }
            const int k0 = u * UK + kg * (UK / 4);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int m = mt * 16 + cl;
#pragma unroll
                for (int t = 0; t < STEPS; t++) {
                    half8_t v = (half8_t)(half_t)0;
                    if (m < p.M) v = *(const half8_t *)(p.x + (size_t)m * p.ldx + k0 + 8 * t);
                    st.x[mt][t] = v;
                }
            }
        }
    };

    int cur_g = -1;
    float sf[NS][4];
    half2_t zneg[NS][4];

    auto flush = [&]() {
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    yv[s][mt][j] += acc[s][mt][j] * sf[s][j];
                    acc[s][mt][j] = (float4_t)0.f;
                }
    };

    auto compute_stage = [&](const SkinnyStage<BITS, MT, FUSED2> &st, int u) {
        if (u >= ue) return;
        const int g = (u * UK) / p.groupsize;
        if (g != cur_g) {  // wave-uniform
            if (cur_g >= 0) flush();
            cur_g = g;
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned z = ((st.zw[s] >> (zshift0 + BITS * j)) & ((1u << BITS) - 1u)) + 1u;
                    const half_t zn = (half_t)(-(float)z - UP::OFF);
                    zneg[s][j] = half2_t{zn, zn};
                    sf[s][j] = active ? (float)st.s[s][j] : 0.f;
                }
        }
        half8_t a[MT][STEPS];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) make_a<BITS>(st.x[mt], a[mt]);
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t wj[LROWS];
#pragma unroll
                for (int r = 0; r < LROWS; r++) wj[r] = active ? st.w[s][r][j] : 0u;
                half8_t b[STEPS];
                make_b<BITS>(wj, zneg[s][j], b);
#pragma unroll
                for (int t = 0; t < STEPS; t++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        acc[s][mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][t], b[t], acc[s][mt][j], 0, 0, 0);
            }
    };

    constexpr int D = (MT <= 1) ? 4 : (MT == 2 ? 3 : 2);  // units in flight per buffer
    SkinnyStage<BITS, MT, FUSED2> cur[D], nxt[D];
#pragma unroll
    for (int d = 0; d < D; d++) load_stage(cur[d], ub + d);
    for (int u = ub; u < ue; u += D) {
        const bool more = (u + D) < ue;
        if (more) {
#pragma unroll
            for (int d = 0; d < D; d++) load_stage(nxt[d], u + D + d);
        }
#pragma unroll
        for (int d = 0; d < D; d++) compute_stage(cur[d], u + d);
        if (more) {
#pragma unroll
            for (int d = 0; d < D; d++) cur[d] = nxt[d];
        }
    }
    if (cur_g >= 0) flush();

    // ---- cross-wave reduction through LDS: red[WAVES][NS][MT*16][64] fp32 ---------------------
    float *red = (float *)smem;
    int *flag = (int *)(smem + (size_t)WAVES * NS * MT * 16 * TILE * 4);
    constexpr int MP = MT * 16;
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = mt * 16 + kg * 4 + r;
                float4_t v = {yv[s][mt][0][r], yv[s][mt][1][r], yv[s][mt][2][r], yv[s][mt][3][r]};
                *(float4_t *)(red + (((size_t)wave * NS + s) * MP + m) * TILE + 4 * cl) = v;
            }
    __syncthreads();

    // each thread finalises 4 adjacent columns of some rows
    constexpr int NQ = MP * (TILE / 4);  // float4 outputs per set
    for (int e = tid; e < NQ; e += T) {
        const int m = e / (TILE / 4), c4 = e % (TILE / 4);
        const int n = tile * TILE + 4 * c4;
        float4_t tot[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            tot[s] = (float4_t)0.f;
#pragma unroll
            for (int w = 0; w < WAVES; w++) tot[s] += *(const float4_t *)(red + (((size_t)w * NS + s) * MP + m) * TILE + 4 * c4);
        }
        const bool ok = (m < p.M) && (n < N);
        if (p.split_k > 1) {
            if (ok) {
#pragma unroll
                for (int s = 0; s < NS; s++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float old = __hip_atomic_fetch_add(p.ws + ((size_t)s * p.M + m) * N + n + j, tot[s][j],
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("" ::"v"(old));
                    }
            }
        } else if (ok) {
            half4_t h;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = tot[0][j];
                if constexpr (FUSED2) v = v * (1.0f / (1.0f + __expf(-v))) * tot[1][j];
                half_t hv = (half_t)v;
                if (p.bias) hv = (half_t)((float)hv + (float)p.bias[n + j]);
                h[j] = hv;
            }
            *(half4_t *)(p.y + (size_t)m * p.ldy + n) = h;
        }
    }
    if (p.split_k <= 1) return;

    __syncthreads();
    if (tid == 0) {
        unsigned t = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)p.split_k - 1);
        if (last) __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    for (int e = tid; e < NQ; e += T) {
        const int m = e / (TILE / 4), c4 = e % (TILE / 4);
        const int n = tile * TILE + 4 * c4;
        if (m < p.M && n < N) {
            half4_t h;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = __hip_atomic_exchange(p.ws + ((size_t)0 * p.M + m) * N + n + j, 0.0f, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
                if constexpr (FUSED2) {
                    float v2 = __hip_atomic_exchange(p.ws + ((size_t)1 * p.M + m) * N + n + j, 0.0f, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                    v = v * (1.0f / (1.0f + __expf(-v))) * v2;
                }
                half_t hv = (half_t)v;
                if (p.bias) hv = (half_t)((float)hv + (float)p.bias[n + j]);
                h[j] = hv;
            }
            *(half4_t *)(p.y + (size_t)m * p.ldy + n) = h;
        }
    }
}

template <int BITS, int MT, int WAVES, bool FUSED2>
static int launch_skinny(const GemvParams &p, hipStream_t stream) {
    constexpr int NS = FUSED2 ? 2 : 1;
    const size_t lds = (size_t)WAVES * NS * MT * 16 * 64 * 4 + 16;
    auto kern = skinny_kernel<BITS, MT, WAVES, FUSED2>;
    static size_t configured = 0;
    if (lds > 48 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = lds;
    }
    dim3 grid(p.ntiles * p.split_k), block(WAVES * 64);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

template <int BITS, bool FUSED2>
static int skinny_m(const GemvParams &p, hipStream_t s) {
    if (p.M <= 16) return launch_skinny<BITS, 1, 4, FUSED2>(p, s);
    if (p.M <= 32) return launch_skinny<BITS, 2, 4, FUSED2>(p, s);
    if constexpr (FUSED2) {
        return GPTQ_E_VARIANT;  // fused: two accumulator sets; capi.hip feeds it 32 rows at a time
    } else {
        return launch_skinny<BITS, 4, 4, FUSED2>(p, s);
    }
}

// p.ntiles must be ceil(N/64); p.split_k >= 1.
int skinny_dispatch(int bits, bool fused2, const GemvParams &p, hipStream_t s) {
    switch (bits) {
        case 2: return fused2 ? skinny_m<2, true>(p, s) : skinny_m<2, false>(p, s);
        case 4: return fused2 ? skinny_m<4, true>(p, s) : skinny_m<4, false>(p, s);
        case 8: return fused2 ? skinny_m<8, true>(p, s) : skinny_m<8, false>(p, s);
    }
    return GPTQ_E_BITS;
}

}  // namespace gptq
