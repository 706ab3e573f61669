// attn_split.h -- how the decode attention cuts a row's history into splits, shared by the producer (decode_attn.hip) and by the consumer that
// merges the splits' records while it stages x (the decode kernel of o_proj, stripe_kernel.inc, ATT instances).  Reference: the single-query
// F.scaled_dot_product_attention of quant/fused_attn.py:154-155 -- one softmax over the whole history; here online-softmax partials.
//
// Round 6.  A split is ONE workgroup that walks its range of the history in tiles of 128 timesteps with a running {max, sum, acc[128]} per wave
// (no LDS and no barrier inside the loop); it leaves a record: {M, den} in fp32 (M in the log2 domain) and its normalised partial output o[128] = acc / den in fp16.  Who merges the records:
//   * nobody (one split): the workgroup stores the fp16 row itself;
//   * the NEXT launch: o_proj's decode kernel reads the records of up to ATT_MAX_SPLITS splits instead of an fp16 x (a kernel boundary is the
//     cheapest grid-wide hand-off on this chip, DESIGN 3.6 -- the in-kernel merge below cost 5.5 us per layer from 129 tokens of context on);
//   * the last split to arrive (system-scope records + an arrival ticket): callers without a merging consumer.
#pragma once
#include <stdint.h>

namespace gptq {

constexpr int ATT_TILE = 128;        // timesteps per tile; a split's range is a multiple of it (the last one: what is left)
constexpr int ATT_MAX_SPLITS = 8;    // splits per (row, head) a launch is ever cut into
constexpr float ATT_M_FLOOR = -1.0e30f;   // running maximum of a wave that has seen no timestep (finite: exp2(floor - M) = 0, never NaN)

struct AttnSplit {
    int nsp, chunk;                  // active splits of this row, timesteps per split (split s = [s * chunk, min((s + 1) * chunk, len)))
};

// len = tokens the row attends to (history + the new one), cap = splits of the launch's grid (<= ATT_MAX_SPLITS), tps = tokens a split should
// at least own.  Both sides evaluate this on the critical path of their first load (the consumer's record addresses depend on nsp), so it is
// written without integer divisions by run-time values (three of them cost ~150 scalar instructions): want = min(cap, ceil(len / tps)) by
// comparisons, ceil(len / want) through constant divisors, nsp = ceil(len / chunk) by comparisons.
__host__ __device__ inline AttnSplit attn_split(int len, int cap, int tps) {
    int want = 1;
#pragma unroll
    for (int i = 1; i < ATT_MAX_SPLITS; i++) want += len > i * tps;
    want = want > cap ? cap : want;
    int per;
    switch (want) {
        case 1: per = len; break;
        case 2: per = (len + 1) >> 1; break;
        case 3: per = (len + 2) / 3; break;
        case 4: per = (len + 3) >> 2; break;
        case 5: per = (len + 4) / 5; break;
        case 6: per = (len + 5) / 6; break;
        case 7: per = (len + 6) / 7; break;
        default: per = (len + 7) >> 3; break;
    }
    const int chunk = (per + ATT_TILE - 1) & ~(ATT_TILE - 1);
    int nsp = 1;
#pragma unroll
    for (int i = 1; i < ATT_MAX_SPLITS; i++) nsp += i * chunk < len;
    return AttnSplit{nsp, chunk};
}

// what a merging consumer needs: the records of `batch` rows -- [batch][S][hidden] fp16 partial outputs o_s = num_s / den_s and
// [batch][S][heads] fp32 {M_s, den_s} (M in the log2 domain).  fp16 partials: a record costs the consumer what an fp16 x costs it (8 KB per
// 4096 k), one split IS the output row (no second rounding), and with several splits the merge below adds one fp32 sum + one rounding.
struct AttnMerge {
    const _Float16 *o16;             // NULL: the consumer reads an ordinary fp16 x
    const float *md;
    const int64_t *pos;              // per row: position of the new token (history length); < 0 = idle row
    int S, tps, heads, t_max;
};

#if defined(__HIPCC__)
// The merge of a row's records, shared by the in-kernel merge (decode_attn.hip) and the consumer-side merge (stripe_kernel.inc) so that both
// produce the same bits whatever the translation unit's floating-point flags are: explicit fma / v_exp_f32 / v_rcp_f32, no reassociation.
//   x = sum_s c_s o_s,  c_s = w_s den_s / sum_t w_t den_t,  w_s = 2^(M_s - max M)
// Slots i >= nsp hold anything finite (zeros, or copies of a live record) and get coefficient 0.
static __device__ inline __attribute__((always_inline)) void attn_merge_coeffs(const float (&M)[ATT_MAX_SPLITS], const float (&den)[ATT_MAX_SPLITS], int nsp,
                                                                               float (&c)[ATT_MAX_SPLITS]) {
#pragma clang fp reassociate(off) contract(off)
    float Mx = M[0];
#pragma unroll
    for (int i = 1; i < ATT_MAX_SPLITS; i++)
        if (i < nsp) Mx = fmaxf(Mx, M[i]);
    float D = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_MAX_SPLITS; i++) {
        c[i] = i < nsp ? __builtin_amdgcn_exp2f(M[i] - Mx) * den[i] : 0.f;
        // (a compiler fence per product: with -ffast-math the backend contracts c[0] + c[1] into fma(w0, den0, c1) although the pragma above
        // clears `contract` -- seen in the ISA of the consumer, not in the -ffp-contract=off unit of the producer: 1 ulp of D apart)
        asm volatile("" : "+v"(c[i]));
        D += c[i];
    }
    const float r = __builtin_amdgcn_rcpf(D);
#pragma unroll
    for (int i = 0; i < ATT_MAX_SPLITS; i++) c[i] *= r;
}
static __device__ inline __attribute__((always_inline)) float attn_merge_value(const _Float16 (&o)[ATT_MAX_SPLITS], const float (&c)[ATT_MAX_SPLITS]) {
#pragma clang fp reassociate(off) contract(off)
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_MAX_SPLITS; i++) v = __builtin_fmaf(c[i], (float)o[i], v);
    return v;
}
// fp32 -> fp16 behind a compiler fence: with fast-math hipcc fuses the multiply above and this conversion into ONE v_fma_mixlo_f16 (a single
// rounding), without it they are v_mul_f32 + v_cvt_f16_f32 (two): 1 fp16 ulp apart once in a few thousand values.  Both sides round twice.
static __device__ inline __attribute__((always_inline)) _Float16 attn_round_f16(float v) {
    asm volatile("" : "+v"(v));
    return (_Float16)v;
}
#endif

}  // namespace gptq
