// skinny_mfma.hip -- weight-streaming dequant-matmul for small batches (M <= 64) on the CHECKPOINT layout (reference
// quant/quant_linear.py:72-137, fused gate/up quant/fused_mlp.py:84-168).  Round 1's small-batch kernel; since round 2 the
// stripe16 kernels (stripe_kernel.inc, stripe_mm.inc) serve every layer that has a stripe image, and this one keeps what they
// do not take: K or group sizes the image does not cover, 3-bit act-order, family = 'abi' / 'skinny' runs.  Weights are read
// exactly once; HBM is the roofline.
//
//  * a wave owns 64 columns; lane l (cl = l & 15, kg = l >> 4) loads ONE dwordx4 per unit =
//    4 adjacent columns x the packed row of k-group kg: a wave instruction fetches 4 rows x 256
//    contiguous bytes (tools/membench.hip: >= 256-byte row segments are needed for HBM speed);
//  * each loaded word IS the B fragment of v_mfma_f32_16x16x32_f16 for its column (8 consecutive
//    k of one column per lane).  It is expanded with the fp16 magic-exponent trick --
//    v_and_or_b32 of the field into the mantissa of a constant whose ulp equals the field's bit
//    weight -- at 1 shift + 4 and_or per 8 weights; the constants' offsets are NOT subtracted:
//    two extra MFMAs per unit against constant fragments give J = sum_k x_k*OFF_k and
//    XS = sum_k x_k, and once per quantisation group
//         y += s * (D - J - z * XS)                      (fp32)
//    so the matrix core does both the multiply and the k-lane reduction and the VALU only
//    unpacks (no shuffles, no int->float converts, no per-weight zero/scale arithmetic);
//  * x is staged per workgroup in LDS in the field order the unpack produces (M <= 16), or read
//    from L2 and permuted in registers (16 < M <= 64);
//  * waves split K inside the workgroup (LDS reduce); workgroups split K further through the
//    one-round-trip fixed-point combine of gptq_device.h (bit-reproducible).
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

// D = (a & mask) | magic should be ONE v_and_or_b32.  With literal operands the compiler emits
// v_and + v_or (VOP3 has no literals on gfx9), so the constants are made opaque once per kernel:
// masks live in SGPRs, magics in VGPRs, and the plain C expression selects v_and_or_b32.  (The
// instruction itself is NOT written as inline asm: its result feeds an MFMA operand and hipcc
// pads VALU->MFMA hazards only for instructions it scheduled itself.)
GPTQ_DEV uint32_t and_or(uint32_t a, uint32_t mask, uint32_t magic) { return (a & mask) | magic; }
typedef uint32_t frag_u32 __attribute__((ext_vector_type(4)));  // 8 halves as 4 dwords
GPTQ_DEV half8_t as_half8(frag_u32 v) { return __builtin_bit_cast(half8_t, v); }
GPTQ_DEV uint32_t pair_bits(float f) { return as_u32(half2_t{(half_t)f, (half_t)f}); }

struct Magics {
    uint32_t m[5];  // magic exponents (VGPRs)
    uint32_t k[5];  // field masks (SGPRs)
};

template <int BITS>
struct Stream;

// 4-bit: one row of 8 k per lane; fragment order fields [0,4 | 1,5 | 2,6 | 3,7]; OFF = 1024 for
// even fields (mantissa bits [3:0], ulp 1) and 64 for odd fields (bits [7:4], ulp 1/16).
template <>
struct Stream<4> {
    static constexpr int LROWS = 1, UK = 32, STEPS = 1, KPW = 8;
    GPTQ_DEV void magics(Magics &g) {
        g.m[0] = vreg_const(0x64006400u);
        g.m[1] = vreg_const(0x54005400u);
        g.k[0] = sreg_const(0x000F000Fu);
        g.k[1] = sreg_const(0x00F000F0u);
    }
    GPTQ_DEV void unpack(const uint32_t (&w)[LROWS], frag_u32 (&b)[STEPS], const Magics &g) {
        const uint32_t w8 = w[0] >> 8;
        b[0][0] = and_or(w[0], g.k[0], g.m[0]);
        b[0][1] = and_or(w[0], g.k[1], g.m[1]);
        b[0][2] = and_or(w8, g.k[0], g.m[0]);
        b[0][3] = and_or(w8, g.k[1], g.m[1]);
    }
    GPTQ_DEV void offsets(frag_u32 (&o)[STEPS]) {
        const uint32_t a = pair_bits(1024.f), c = pair_bits(64.f);
        o[0] = frag_u32{a, c, a, c};
    }
    GPTQ_DEV int field_at(int, int e) { return (e >> 1) + 4 * (e & 1); }
};

// 8-bit: two rows of 4 k per lane; bytes [0,2 | 1,3 | 4,6 | 5,7]; OFF = 1024 everywhere.
template <>
struct Stream<8> {
    static constexpr int LROWS = 2, UK = 32, STEPS = 1, KPW = 4;
    GPTQ_DEV void magics(Magics &g) {
        g.m[0] = vreg_const(0x64006400u);
        g.k[0] = sreg_const(0x00FF00FFu);
    }
    GPTQ_DEV void unpack(const uint32_t (&w)[LROWS], frag_u32 (&b)[STEPS], const Magics &g) {
        b[0][0] = and_or(w[0], g.k[0], g.m[0]);
        b[0][1] = and_or(w[0] >> 8, g.k[0], g.m[0]);
        b[0][2] = and_or(w[1], g.k[0], g.m[0]);
        b[0][3] = and_or(w[1] >> 8, g.k[0], g.m[0]);
    }
    GPTQ_DEV void offsets(frag_u32 (&o)[STEPS]) {
        const uint32_t a = pair_bits(1024.f);
        o[0] = frag_u32{a, a, a, a};
    }
    GPTQ_DEV int field_at(int, int e) { return 4 * (e >> 2) + ((e >> 1) & 1) + 2 * (e & 1); }
};

// 2-bit: one row of 16 k per lane, two MFMA steps; pair f = fields (f, f+8), step f/4.
// fields 0..4 sit in mantissa bits [9:0] (five different ulps), 5..7 after a shift by 10.
template <>
struct Stream<2> {
    static constexpr int LROWS = 1, UK = 64, STEPS = 2, KPW = 16;
    GPTQ_DEV void magics(Magics &g) {
        g.m[0] = vreg_const(0x64006400u);  // 1024, ulp 1
        g.m[1] = vreg_const(0x5C005C00u);  // 256,  ulp 1/4
        g.m[2] = vreg_const(0x54005400u);  // 64,   ulp 1/16
        g.m[3] = vreg_const(0x4C004C00u);  // 16,   ulp 1/64
        g.m[4] = vreg_const(0x44004400u);  // 4,    ulp 1/256
        g.k[0] = sreg_const(0x00030003u);
        g.k[1] = sreg_const(0x000C000Cu);
        g.k[2] = sreg_const(0x00300030u);
        g.k[3] = sreg_const(0x00C000C0u);
        g.k[4] = sreg_const(0x03000300u);
    }
    GPTQ_DEV void unpack(const uint32_t (&w)[LROWS], frag_u32 (&b)[STEPS], const Magics &g) {
        const uint32_t wh = w[0] >> 10;
        b[0][0] = and_or(w[0], g.k[0], g.m[0]);
        b[0][1] = and_or(w[0], g.k[1], g.m[1]);
        b[0][2] = and_or(w[0], g.k[2], g.m[2]);
        b[0][3] = and_or(w[0], g.k[3], g.m[3]);
        b[1][0] = and_or(w[0], g.k[4], g.m[4]);
        b[1][1] = and_or(wh, g.k[0], g.m[0]);
        b[1][2] = and_or(wh, g.k[1], g.m[1]);
        b[1][3] = and_or(wh, g.k[2], g.m[2]);
    }
    GPTQ_DEV void offsets(frag_u32 (&o)[STEPS]) {
        o[0] = frag_u32{pair_bits(1024.f), pair_bits(256.f), pair_bits(64.f), pair_bits(16.f)};
        o[1] = frag_u32{pair_bits(4.f), pair_bits(1024.f), pair_bits(256.f), pair_bits(64.f)};
    }
    GPTQ_DEV int field_at(int t, int e) { return 4 * t + (e >> 1) + 8 * (e & 1); }
};

// natural-order x (8*STEPS halves of the lane's k range) -> fragment order
template <int BITS>
GPTQ_DEV void permute_a(const half8_t (&xin)[Stream<BITS>::STEPS], half8_t (&a)[Stream<BITS>::STEPS]) {
    constexpr int STEPS = Stream<BITS>::STEPS;
#pragma unroll
    for (int t = 0; t < STEPS; t++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int f = Stream<BITS>::field_at(t, e);
            a[t][e] = xin[f / 8][f % 8];
        }
}

// STG units form one pipeline stage; a stage never straddles a quantisation group.
template <int BITS, int STG, int MT, bool XLDS, bool FUSED2>
struct StreamStage {
    uint32_t w[FUSED2 ? 2 : 1][STG][Stream<BITS>::LROWS][4];
    half4_t s[FUSED2 ? 2 : 1];
    uint32_t zw[FUSED2 ? 2 : 1];
    half8_t x[XLDS ? 1 : STG][XLDS ? 1 : MT][Stream<BITS>::STEPS];  // only when A comes from global
};

template <int BITS, int STG, int MT, int WAVES, bool XLDS, bool FUSED2>
__global__ void __launch_bounds__(WAVES * 64) stream_kernel(const GemvParams p) {
    using ST = Stream<BITS>;
    constexpr int KPW = ST::KPW, LROWS = ST::LROWS, UK = ST::UK, STEPS = ST::STEPS;
    constexpr int NS = FUSED2 ? 2 : 1, T = WAVES * 64, TILE = 64;
    constexpr int ROWS_PER_UNIT = UK / KPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 15, kg = lane >> 4;
    u64_t *dbg = p.dbg ? p.dbg + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave) * 8 : nullptr;
    auto stamp = [&](int i) {
        if (dbg && lane == 0) dbg[i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int N = p.N;
    const int n0 = tile * TILE + 4 * cl;
    const bool active = n0 < N;
    const int nc = active ? n0 : 0;  // ragged N: idle lanes read column 0, results are dropped
    const int ldz = N / KPW;
    const int zshift0 = BITS * (nc % KPW);

    // this slice's units (a multiple of STG), split contiguously over the waves in whole stages
    const int ub_s = slice * p.chunks_per_slice, ue_s = min(p.nchunks, ub_s + p.chunks_per_slice);
    const int stages_s = (ue_s - ub_s) / STG;
    const int spw = (stages_s + WAVES - 1) / WAVES;
    const int ub = min(ue_s, ub_s + wave * spw * STG), ue = min(ue_s, ub + spw * STG);
    const int upg = p.units_per_group, upg_shift = p.upg_shift;

    // ---- x in LDS (XLDS): [rows][xstride] halves in fragment order + one zero chunk -----------
    const int nk = (ue_s - ub_s) * UK;
    const int xstride = nk + 8;  // +16 B so that rows start on different bank groups
    half_t *lx = (half_t *)smem;
    const int xrows = XLDS ? min(p.M, MT * 16) : 0;
    half_t *zero_chunk = lx + (size_t)xrows * xstride;

    using Stage = StreamStage<BITS, STG, MT, XLDS, FUSED2>;
    auto load_stage = [&](Stage &st, int u0) {
        const int g = (upg_shift >= 0) ? (u0 >> upg_shift) : (u0 / upg);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const uint32_t *base = p.qw[s] + (size_t)(u0 * ROWS_PER_UNIT + kg * LROWS) * N + nc;
#pragma unroll
            for (int i = 0; i < STG; i++)
#pragma unroll
                for (int r = 0; r < LROWS; r++) {
                    const u32x4 v = __builtin_nontemporal_load((const u32x4 *)(base + (size_t)(i * ROWS_PER_UNIT + r) * N));
#pragma unroll
                    for (int j = 0; j < 4; j++) st.w[s][i][r][j] = v[j];
                }
            st.s[s] = *(const half4_t *)(p.sc[s] + (size_t)g * N + nc);
            st.zw[s] = (uint32_t)p.qz[s][(size_t)g * ldz + nc / KPW];
        }
        if constexpr (!XLDS) {
#pragma unroll
            for (int i = 0; i < STG; i++) {
                const int k0 = (u0 + i) * UK + kg * (UK / 4);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const int m = mt * 16 + cl;
#pragma unroll
                    for (int t = 0; t < STEPS; t++) {
                        half8_t v = (half8_t)(half_t)0;
                        if (m < p.M) v = *(const half8_t *)(p.x + (size_t)m * p.ldx + k0 + 8 * t);
                        st.x[i][mt][t] = v;
                    }
                }
            }
        }
    };

    // ---- issue order: x first (oldest in the vmcnt queue, so its wait leaves the weights in
    // flight and it is not stuck behind the weight flood in the memory system), then DEPTH-1
    // stages of weights, then the LDS write + barrier.
    constexpr int XR = 2;  // x spans held in registers per thread before the weights are issued
    const int spans_per_row = XLDS ? nk / (8 * STEPS) : 0;
    const int total_spans = xrows * spans_per_row;
    half8_t xreg[XR][STEPS];
    if constexpr (XLDS) {
#pragma unroll
        for (int r = 0; r < XR; r++) {
            const int idx = tid + r * T;
            if (idx < total_spans) {
                int m = 0, sp = idx;
                if (xrows > 1) {
                    m = idx / spans_per_row;
                    sp = idx - m * spans_per_row;
                }
                const half_t *src = p.x + (size_t)m * p.ldx + (size_t)ub_s * UK + (size_t)sp * 8 * STEPS;
#pragma unroll
                for (int t = 0; t < STEPS; t++) xreg[r][t] = *(const half8_t *)(src + 8 * t);
            }
        }
    }

    constexpr int DEPTH = FUSED2 ? 2 : 3;  // stages resident per wave (registers bound it)
    Stage sa, sb, sc;
    if (ub < ue) load_stage(sa, ub);
    if (DEPTH == 3 && ub + STG < ue) load_stage(sb, ub + STG);
    stamp(1);

    if constexpr (XLDS) {
        auto put = [&](int idx, const half8_t (&xin)[STEPS]) {
            int m = 0, sp = idx;
            if (xrows > 1) {
                m = idx / spans_per_row;
                sp = idx - m * spans_per_row;
            }
            half8_t a[STEPS];
            permute_a<BITS>(xin, a);
            half_t *dst = lx + (size_t)m * xstride + (size_t)sp * 8 * STEPS;
#pragma unroll
            for (int t = 0; t < STEPS; t++) *(half8_t *)(dst + 8 * t) = a[t];
        };
#pragma unroll
        for (int r = 0; r < XR; r++)
            if (tid + r * T < total_spans) put(tid + r * T, xreg[r]);
        for (int idx = tid + XR * T; idx < total_spans; idx += T) {
            int m = 0, sp = idx;
            if (xrows > 1) {
                m = idx / spans_per_row;
                sp = idx - m * spans_per_row;
            }
            half8_t xin[STEPS];
            const half_t *src = p.x + (size_t)m * p.ldx + (size_t)ub_s * UK + (size_t)sp * 8 * STEPS;
#pragma unroll
            for (int t = 0; t < STEPS; t++) xin[t] = *(const half8_t *)(src + 8 * t);
            put(idx, xin);
        }
        if (tid < 8) zero_chunk[tid] = (half_t)0;
        __syncthreads();
    }
    stamp(2);

    // constant fragments: magic constants, offset pattern, ones
    Magics mg;
    ST::magics(mg);
    frag_u32 offs[STEPS];
    ST::offsets(offs);
    const uint32_t one2 = pair_bits(1.f);
    const frag_u32 ones = frag_u32{one2, one2, one2, one2};

    float4_t yv[NS][MT][4], acc[NS][MT][4], accj[MT], accx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        accj[mt] = (float4_t)0.f;
        accx[mt] = (float4_t)0.f;
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                yv[s][mt][j] = (float4_t)0.f;
                acc[s][mt][j] = (float4_t)0.f;
            }
    }

    // LDS address of this lane's A fragment: row cl of the staged x, or the zero chunk
    const half_t *arow[MT];
    int astep[MT], tstep[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const bool live = XLDS && (mt * 16 + cl) < xrows;
        arow[mt] = live ? (lx + (size_t)(mt * 16 + cl) * xstride + kg * (UK / 4)) : zero_chunk;
        astep[mt] = live ? UK : 0;
        tstep[mt] = live ? 8 : 0;
    }

    auto compute_stage = [&](const Stage &st, int u0) {
#pragma unroll
        for (int i = 0; i < STG; i++) {
            half8_t a[MT][STEPS];
            if constexpr (XLDS) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const half_t *ap = arow[mt] + (size_t)(u0 + i - ub_s) * astep[mt];
#pragma unroll
                    for (int t = 0; t < STEPS; t++) a[mt][t] = *(const half8_t *)(ap + t * tstep[mt]);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) permute_a<BITS>(st.x[i][mt], a[mt]);
            }
#pragma unroll
            for (int t = 0; t < STEPS; t++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    accj[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][t], as_half8(offs[t]), accj[mt], 0, 0, 0);
                    accx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][t], as_half8(ones), accx[mt], 0, 0, 0);
                }
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t wj[LROWS];
#pragma unroll
                    for (int r = 0; r < LROWS; r++) wj[r] = st.w[s][i][r][j];
                    frag_u32 b[STEPS];
                    ST::unpack(wj, b, mg);
#pragma unroll
                    for (int t = 0; t < STEPS; t++)
#pragma unroll
                        for (int mt = 0; mt < MT; mt++)
                            acc[s][mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][t], as_half8(b[t]), acc[s][mt][j], 0, 0, 0);
                }
        }
        // end of a quantisation group (wave-uniform): fold scale and zero, restart accumulators
        const int un = u0 + STG;
        const bool group_end = (upg_shift >= 0) ? ((un & (upg - 1)) == 0) : ((un % upg) == 0);
        if (group_end || un >= ue) {
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float zf = (float)(((st.zw[s] >> (zshift0 + BITS * j)) & ((1u << BITS) - 1u)) + 1u);
                    const float sf = (float)st.s[s][j];
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        yv[s][mt][j] += sf * (acc[s][mt][j] - accj[mt] - zf * accx[mt]);
                        acc[s][mt][j] = (float4_t)0.f;
                    }
                }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                accj[mt] = (float4_t)0.f;
                accx[mt] = (float4_t)0.f;
            }
        }
    };

    if (ub < ue) {
        int u = ub;
        if constexpr (DEPTH == 3) {
            while (true) {
                if (u + 2 * STG < ue) load_stage(sc, u + 2 * STG);
                compute_stage(sa, u);
                if (u == ub) stamp(3);
                u += STG;
                if (u >= ue) break;
                if (u + 2 * STG < ue) load_stage(sa, u + 2 * STG);
                compute_stage(sb, u);
                u += STG;
                if (u >= ue) break;
                if (u + 2 * STG < ue) load_stage(sb, u + 2 * STG);
                compute_stage(sc, u);
                u += STG;
                if (u >= ue) break;
            }
        } else {
            while (true) {
                if (u + STG < ue) load_stage(sb, u + STG);
                compute_stage(sa, u);
                if (u == ub) stamp(3);
                u += STG;
                if (u >= ue) break;
                if (u + STG < ue) load_stage(sa, u + STG);
                compute_stage(sb, u);
                u += STG;
                if (u >= ue) break;
            }
        }
    }
    stamp(4);

    // ---- cross-wave reduction through LDS: red[WAVES][NS][mrows][64] fp32 ---------------------
    constexpr int MP = MT * 16;
    const int mrows = min(p.M, MP);
    if constexpr (XLDS) __syncthreads();  // staged x no longer needed
    float *red = (float *)smem;
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = mt * 16 + kg * 4 + r;
                if (m < mrows) {
                    float4_t v = {yv[s][mt][0][r], yv[s][mt][1][r], yv[s][mt][2][r], yv[s][mt][3][r]};
                    *(float4_t *)(red + (((size_t)wave * NS + s) * mrows + m) * TILE + 4 * cl) = v;
                }
            }
    __syncthreads();
    stamp(5);

    const int nq = mrows * (TILE / 4);
    // K slices are combined through per-slice partial tiles in the workspace + ONE arrival ticket per
    // tile (not per-element atomics: M*N*S returning atomics saturate at ~50 G/s, tools/timeline_stream.py).
    // Publication follows MI355X_MICROARCH.md "Valid forms": system-scope (write-through) stores, vmcnt(0),
    // agent-scope ticket; the last slice reads every partial with system-scope loads, in slice order
    // (bit-reproducible), and applies the epilogue.
    __shared__ int last_slice;
    const int S = p.split_k;
    float *part_base = (float *)((char *)p.ws + SPLITK_PART_OFFSET) + (size_t)tile * S * NS * mrows * TILE;
    if (S > 1) {
        float *mine = part_base + (size_t)slice * NS * mrows * TILE;
        for (int e = tid; e < nq; e += T) {
            const int m = e >> 4, c4 = e & 15;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                float4_t tot = (float4_t)0.f;
#pragma unroll
                for (int w = 0; w < WAVES; w++) tot += *(const float4_t *)(red + (((size_t)w * NS + s) * mrows + m) * TILE + 4 * c4);
                store_sys16(mine + ((size_t)s * mrows + m) * TILE + 4 * c4, tot);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned *ticket = (unsigned *)((char *)p.ws + SPLITK_TICKET_OFFSET) + tile;
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == (unsigned)(S - 1));
            if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_slice = last;
        }
        __syncthreads();
        if (!last_slice) return;
    }
    for (int e = tid; e < nq; e += T) {
        const int m = e >> 4, c4 = e & 15;
        const int n = tile * TILE + 4 * c4;
        float4_t tot[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            tot[s] = (float4_t)0.f;
            if (S > 1) {
                for (int sl0 = 0; sl0 < S; sl0 += 8) {   // 8 slices in flight per wait, summed in slice order
                    float4_t v[8];
                    const float *src[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int sl = min(sl0 + i, S - 1);
                        src[i] = part_base + ((size_t)sl * NS + s) * mrows * TILE + (size_t)m * TILE + 4 * c4;
                    }
                    load_sys16_x8(v, src);
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (sl0 + i < S) tot[s] += v[i];
                }
            } else {
#pragma unroll
                for (int w = 0; w < WAVES; w++) tot[s] += *(const float4_t *)(red + (((size_t)w * NS + s) * mrows + m) * TILE + 4 * c4);
            }
        }
        if (n < N) {
            half4_t h;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float t0 = tot[0][j];
                float v = t0;
                if constexpr (FUSED2) v = t0 * (1.0f / (1.0f + __expf(-t0))) * tot[1][j];  // silu on the fp32 sum
                half_t hv = (half_t)v;
                if (p.bias) hv = (half_t)((float)hv + (float)p.bias[n + j]);
                h[j] = hv;
            }
            *(half4_t *)(p.y + (size_t)m * p.ldy + n) = h;
        }
    }
    stamp(6);
}

template <int BITS, int STG, int MT, int WAVES, bool XLDS, bool FUSED2>
static int launch_stream(const GemvParams &p, hipStream_t stream) {
    constexpr int NS = FUSED2 ? 2 : 1;
    constexpr int UK = Stream<BITS>::UK;
    const int mrows = p.M < MT * 16 ? p.M : MT * 16;
    const size_t red = (size_t)WAVES * NS * mrows * 64 * 4;
    const size_t xb = XLDS ? ((size_t)mrows * ((size_t)p.chunks_per_slice * UK + 8) * 2 + 16) : 0;
    const size_t lds = ((red > xb ? red : xb) + 15) & ~(size_t)15;
    if (lds > 160 * 1024 - 64) return GPTQ_E_SHAPE;
    auto kern = stream_kernel<BITS, STG, MT, WAVES, XLDS, FUSED2>;
    static LdsOptIn opt_in;   // per instantiation; per device inside
    if (int rc = opt_in.ensure((const void *)kern, lds)) return rc;
    dim3 grid(p.ntiles, p.split_k), block(WAVES * 64);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return (int)hipGetLastError();
}

template <int BITS, bool FUSED2>
static int stream_m(const GemvParams &p, int stg, int waves, bool xlds, hipStream_t s) {
    if (p.M <= 16) {
        if (xlds) {
            if (stg == 4) {
                switch (waves) {
                    case 2: return launch_stream<BITS, 4, 1, 2, true, FUSED2>(p, s);
                    case 8: return launch_stream<BITS, 4, 1, 8, true, FUSED2>(p, s);
                    default: return launch_stream<BITS, 4, 1, 4, true, FUSED2>(p, s);
                }
            }
            if (stg == 2) return launch_stream<BITS, 2, 1, 4, true, FUSED2>(p, s);
            return launch_stream<BITS, 1, 1, 4, true, FUSED2>(p, s);
        }
        if (stg == 4) return launch_stream<BITS, 4, 1, 4, false, FUSED2>(p, s);
        return launch_stream<BITS, 1, 1, 4, false, FUSED2>(p, s);
    }
    if (p.M <= 32) {
        if (stg == 4) return xlds ? launch_stream<BITS, 4, 2, 4, true, FUSED2>(p, s) : launch_stream<BITS, 4, 2, 4, false, FUSED2>(p, s);
        return launch_stream<BITS, 1, 2, 4, false, FUSED2>(p, s);
    }
    if constexpr (FUSED2) {
        return GPTQ_E_VARIANT;  // fused: two accumulator sets; capi.hip feeds it 32 rows at a time
    } else {
        if (stg == 4) return xlds ? launch_stream<BITS, 2, 4, 4, true, FUSED2>(p, s) : launch_stream<BITS, 2, 4, 4, false, FUSED2>(p, s);
        return launch_stream<BITS, 1, 4, 4, false, FUSED2>(p, s);
    }
}

// p.ntiles = ceil(N/64); p.nchunks = K/UK units; p.chunks_per_slice (multiple of stg) units per
// K slice; p.units_per_group = groupsize/UK (a multiple of stg), p.upg_shift = log2 or -1.
// stg in {1,2,4} (M > 16 or !xlds: {1,4}; M > 32: {1,2}); waves in {2,4,8} (stg == 4, xlds only).
int skinny_dispatch(int bits, bool fused2, int stg, int waves, bool xlds, const GemvParams &p, hipStream_t s) {
    switch (bits) {
        case 2: return fused2 ? stream_m<2, true>(p, stg, waves, xlds, s) : stream_m<2, false>(p, stg, waves, xlds, s);
        case 4: return fused2 ? stream_m<4, true>(p, stg, waves, xlds, s) : stream_m<4, false>(p, stg, waves, xlds, s);
        case 8: return fused2 ? stream_m<8, true>(p, stg, waves, xlds, s) : stream_m<8, false>(p, stg, waves, xlds, s);
    }
    return GPTQ_E_BITS;
}

}  // namespace gptq
