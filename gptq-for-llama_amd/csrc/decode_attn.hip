// decode_attn.hip -- the two non-GEMV pieces of a batch-1 decode step that the reference does with
// torch ops between its Triton kernels (quant/fused_attn.py:126-155): RoPE on q,k + KV-cache
// append (triton_rotate_half_ :126, torch.cat :142-143) and the single-query attention
// (F.scaled_dot_product_attention :155).  Written as plain HIP so that the whole decode step can
// be captured in ONE hipGraph: the current position is read from device memory, the KV cache is
// a preallocated [t_max, heads*head_dim] buffer (no torch.cat, no shape change per token).
//
// HBM-bound elementwise / reduction work (K,V rows are read once per token): no MFMA.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "attn_split.h"
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {


// Cross-lane reductions on the VALU (DPP row operations + gfx950 permlane swaps) instead of __shfl_xor, which hipcc lowers to
// ds_bpermute_b32: ~120 cycles of LDS round trip per dependent step -- the per-wave stamps (tools/timeline_attn.py) showed 3500
// cycles in the 32 dependent shuffles of the score loop alone, 6500 of the launch's 10 000.  Same operand pairs as the xor
// butterfly: bit-identical results.
template <int CTRL>
GPTQ_DEV float att_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
GPTQ_DEV float att_sum16(float v) {   // every lane of a 16-lane row ends with the row's sum (xor 1, 2, 4, 8)
    v += att_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v += att_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v += att_dpp<0x141>(v);  // row_half_mirror
    v += att_dpp<0x140>(v);  // row_mirror
    return v;
}
GPTQ_DEV float att_max16(float v) {
    v = fmaxf(v, att_dpp<0xB1>(v));
    v = fmaxf(v, att_dpp<0x4E>(v));
    v = fmaxf(v, att_dpp<0x141>(v));
    v = fmaxf(v, att_dpp<0x140>(v));
    return v;
}
GPTQ_DEV float att_rows_sum(float v) {   // sum over the four 16-lane rows of the wave (xor 16, 32)
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s16 = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    const uint32_t u2 = __builtin_bit_cast(uint32_t, s16);
    auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
}
GPTQ_DEV float att_rows_max(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m16 = fmaxf(__builtin_bit_cast(float, (uint32_t)a[0]), __builtin_bit_cast(float, (uint32_t)a[1]));
    const uint32_t u2 = __builtin_bit_cast(uint32_t, m16);
    auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)b[0]), __builtin_bit_cast(float, (uint32_t)b[1]));
}
GPTQ_DEV float att_wave_sum(float v) { return att_rows_sum(att_sum16(v)); }
GPTQ_DEV float att_wave_max(float v) { return att_rows_max(att_max16(v)); }

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
        dot = att_sum16(dot);
        if (d8 == 0) sc[tl] = (tl < nact) ? dot * scale : -INFINITY;
    }
    __syncthreads();

    // ---- softmax statistics of the chunk ------------------------------------------------------
    float sv = (tid < ATT_TS) ? sc[tid] : -INFINITY;
    float m = sv;
    m = att_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float p = (tid < nact) ? __expf(sv - m) : 0.f;
    float l = p;
    l = att_wave_sum(l);
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


// ---------------------------------------------------------------------------------------
// One launch per layer: RoPE(q, k) + KV append + single-query attention (fused_attn.py:126-155 as ONE kernel).
//
// Round 6: a STREAMING kernel.  grid = (heads, rows of the decode batch, S splits); the number of ACTIVE splits of a row and their ranges follow
// from the row's length at run time (attn_split.h: tokens-per-split target `tps`, at most S) -- rounds 2-5 cut fixed 128-step splits, so every
// context above 128 tokens paid the cross-workgroup merge (records published write-through, ticket, last arriver re-reads: 5.5 us per layer,
// 925 -> 795 tok/s between 0 and 500 tokens of context).  Now a split walks its range in tiles of 32 NW timesteps:
//   * K / V rows by buffer loads (16 bytes per lane: 16 lanes = one 256-byte head row, four rows per wave instruction); the descriptor ends at the
//     split's last old row, so lanes past it read zeros without touching memory -- no clamps, no branches in the load phase; the NEXT tile is
//     requested before the current one is used (two register sets, counted vmcnt across the loop);
//   * every WAVE keeps its own online-softmax state {M, l, acc} for the timesteps it owns (scores summed over 16 lanes with DPP row operations, the
//     tile maximum over the wave's four rows with permlane swaps): no LDS traffic and no barrier inside the loop; q . k and p . v are
//     v_fma_mix_f32 (fp16 operand, fp32 accumulate: one VALU per multiply-add, no conversions); exponentials in the log2 domain (v_exp_f32);
//   * the new token (k rotated here, v) is taken from LDS by wave 0 of the split that owns position `pos` and appended to the cache at the end;
//   * the NW waves meet once, through LDS: {M, den, num[128]} of the split.
// What happens to that record (template REC / run time):
//   one active split, !REC  -> out = fp16(num / den), done (every context up to `tps` tokens; every decode batch of three or more rows);
//   REC                     -> the record is stored (plain stores) for the NEXT launch to merge: o_proj's decode kernel stages x from the records
//                              of up to ATT_MAX_SPLITS splits (stripe_kernel.inc, ATT instances) -- the kernel boundary is the hand-off;
//   several splits, !REC    -> records go out with system-scope stores, an arrival ticket per (row, head) elects the last split, which merges
//                              them (callers without a merging consumer; act-order o_proj with the producer-side permutation).
// ws = [batch][S][heads * 128] fp16 partial outputs | [batch][S][heads] fp32 {M, den} | [batch][heads] uint32 tickets (zero between launches).
// A negative position marks an idle row of the batch: nothing is read or written for it.
// ---------------------------------------------------------------------------------------
struct AttnArgs {
    const half_t *qkv;
    const int64_t *pos;
    half_t *kc, *vc, *out;
    half_t *o16;                 // records: [batch][S][heads * 128] fp16 partial outputs
    float *md;                   //          [batch][S][heads] {M, den}
    unsigned *tickets;
    const float2 *rope_tab;
    const int32_t *out_perm;
    int heads, t_max, ldq, ldo, tps;
    float inv_base, scale2;      // scale2 = softmax scale x log2(e): scores live in the log2 domain
};

GPTQ_DEV u32x4 att_load16(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

template <int NW, bool REC>
__global__ void __launch_bounds__(NW * 64) attn_stream_kernel(const AttnArgs a) {
    constexpr int TILE = NW * 32, RP = NW * 4;   // timesteps per tile / per pass of the workgroup (16 lanes per timestep)
    constexpr int NP = 8;                        // passes per tile = 16-byte K (and V) loads per lane and tile
    __shared__ float qs[ATT_HD];
    __shared__ __attribute__((aligned(16))) half_t knew[ATT_HD];
    __shared__ __attribute__((aligned(16))) half_t vnew[ATT_HD];
    __shared__ __attribute__((aligned(16))) float accs[NW][ATT_HD];
    __shared__ float mw[NW], lw[NW];
    __shared__ int last_flag;
    const int h = blockIdx.x, b = blockIdx.y, s = blockIdx.z, S = gridDim.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hd = a.heads * ATT_HD;
    // out_perm (round 5): o_proj is an act-order layer whose image holds group-sorted rows -- element k of the attention output goes where its sorted
    // order wants it, so o_proj runs the trivial kernel.  Requested first: nothing depends on it until the store.
    const int ocol = (!REC && tid < ATT_HD) ? (a.out_perm ? a.out_perm[h * ATT_HD + tid] : h * ATT_HD + tid) : 0;
    // this row's q / k / v halves of the head: they depend on nothing but the grid, so they are requested BEFORE the position is known (the
    // row was written by the previous launch on other XCDs: a fabric round trip that now runs under the scalar load of `pos`)
    const half_t *qrow = a.qkv + (size_t)b * a.ldq;
    half_t q0 = (half_t)0, q1 = (half_t)0, k0 = (half_t)0, k1 = (half_t)0, nv0 = (half_t)0, nv1 = (half_t)0;
    if (tid < ATT_HD / 2) {
        const half_t *q = qrow + (size_t)h * ATT_HD + tid;
        q0 = q[0]; q1 = q[ATT_HD / 2];
        k0 = q[hd]; k1 = q[hd + ATT_HD / 2];
        nv0 = q[2 * hd]; nv1 = q[2 * hd + ATT_HD / 2];
    }
    // (through the scalar cache: constant address space -- inside a by-value struct the pointer carries no noalias, and hipcc made it a vector load)
    const int64_t pos64 = ((const __attribute__((address_space(4))) int64_t *)a.pos)[b];
    if (pos64 < 0 || pos64 >= a.t_max) return;
    const int pos = (int)pos64, len = pos + 1;
    const AttnSplit sp = attn_split(len, S, a.tps);
    if (s >= sp.nsp) return;
    const int t0 = s * sp.chunk;
    const int t_end = min(t0 + sp.chunk, len);   // this split attends to rows [t0, t_end)
    const bool owner = t_end == len;             // ... the last of which is the new token: this split rotates k, appends k / v, takes that row from LDS
    const int n_old = min(t_end, pos);           // cache rows of this split: [t0, n_old)
    const int ntiles = (t_end - t0 + TILE - 1) / TILE;   // >= 1
    const int d8 = tid & 15, tsub = tid >> 4;

    // descriptors over rows [0, n_old) of this (row, head): anything past the split's last cache row reads as zero, without a memory request
    const size_t slice = ((size_t)b * a.t_max) * hd + (size_t)h * ATT_HD;
    const int nrec = n_old > 0 ? ((n_old - 1) * hd + ATT_HD) * 2 : 0;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(a.kc + slice), 0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(a.vc + slice), 0, nrec, 0x00020000);
    const int hd2 = hd * 2;
    u32x4 KA[NP], VA[NP], KB[NP], VB[NP];
    auto load = [&](u32x4 (&K)[NP], u32x4 (&V)[NP], int tb) {
        const int vo = (tb + tsub) * hd2 + d8 * 16;
#pragma unroll
        for (int it = 0; it < NP; it++) K[it] = att_load16(rk, vo + it * RP * hd2);
#pragma unroll
        for (int it = 0; it < NP; it++) V[it] = att_load16(rv, vo + it * RP * hd2);
    };
    // a context of up to two passes (8 NW timesteps: the first tokens of a sequence) requests and computes two passes, not eight
    const bool tiny = t_end - t0 <= 2 * RP;
    if (tiny) {
        const int vo = (t0 + tsub) * hd2 + d8 * 16;
        KA[0] = att_load16(rk, vo);
        KA[1] = att_load16(rk, vo + RP * hd2);
        VA[0] = att_load16(rv, vo);
        VA[1] = att_load16(rv, vo + RP * hd2);
    } else {
        load(KA, VA, t0);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- RoPE of q (every split) and of the new k (owner) under the latency of the first tile ----
    half_t nk0 = (half_t)0, nk1 = (half_t)0;
    if (tid < ATT_HD / 2) {
        const int c = tid;
        float cs, sn;
        if (a.rope_tab) {   // {cos, sin} of (pos, c) from the table rope_table_kernel filled with the SAME instructions
            const float2 e = a.rope_tab[(size_t)pos * (ATT_HD / 2) + c];
            cs = e.x;
            sn = e.y;
        } else {
            const float freq = expf((float)c * a.inv_base) * (float)pos;
            cs = cosf(freq);
            sn = sinf(freq);
        }
        const float qx = (float)q0, qy = (float)q1;
        qs[c] = (float)(half_t)(qx * cs - qy * sn);       // rounded to fp16 like the in-place reference RoPE (fused_attn.py:43-57)
        qs[c + ATT_HD / 2] = (float)(half_t)(qx * sn + qy * cs);
        if (owner) {
            const float kx = (float)k0, ky = (float)k1;
            nk0 = (half_t)(kx * cs - ky * sn);
            nk1 = (half_t)(kx * sn + ky * cs);
            knew[c] = nk0;
            knew[c + ATT_HD / 2] = nk1;
            vnew[c] = nv0;
            vnew[c + ATT_HD / 2] = nv1;
        }
    }
    __syncthreads();
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) qf[j] = qs[d8 * 8 + j];

    // ---- the wave's running state: M (log2 domain), l and acc[8 dims of this lane] relative to M ----
    float M = ATT_M_FLOOR, l = 0.f, av[8];
#pragma unroll
    for (int j = 0; j < 8; j++) av[j] = 0.f;
    auto scores8 = [&](const u32x4 k) {
        const half8_t k8 = __builtin_bit_cast(half8_t, k);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) dot = __builtin_fmaf(qf[j], (float)k8[j], dot);
        return att_sum16(dot) * a.scale2;
    };
    auto tile_full = [&](const u32x4 (&K)[NP], const u32x4 (&V)[NP]) {
        float sc[NP];
#pragma unroll
        for (int it = 0; it < NP; it++) sc[it] = scores8(K[it]);
        float mt = sc[0];
#pragma unroll
        for (int it = 1; it < NP; it++) mt = fmaxf(mt, sc[it]);
        mt = att_rows_max(mt);                                   // the wave's four rows: M stays wave-uniform
        const float Mn = fmaxf(M, mt), alpha = __builtin_amdgcn_exp2f(M - Mn);
        float p[NP], ps = 0.f;
#pragma unroll
        for (int it = 0; it < NP; it++) {
            p[it] = __builtin_amdgcn_exp2f(sc[it] - Mn);
            ps += p[it];
        }
        l = __builtin_fmaf(l, alpha, ps);
#pragma unroll
        for (int j = 0; j < 8; j++) av[j] *= alpha;
#pragma unroll
        for (int it = 0; it < NP; it++) {
            const half8_t v8 = __builtin_bit_cast(half8_t, V[it]);
#pragma unroll
            for (int j = 0; j < 8; j++) av[j] = __builtin_fmaf(p[it], (float)v8[j], av[j]);
        }
        M = Mn;
    };
    // the split's last tile (PASSES = NP), or a tiny context (PASSES = 2): rows past t_end are masked, and the lanes that hold position `pos`
    // take that row's k / v from LDS (the new token is not in the cache yet).  Straight-line code: masks and selects, no branches.
    auto tile_last = [&](const u32x4 (&K)[NP], const u32x4 (&V)[NP], int tb, auto passes) {
        constexpr int PASSES = decltype(passes)::value;
        u32x4 kn = u32x4{0u, 0u, 0u, 0u}, vn = u32x4{0u, 0u, 0u, 0u};
        if (owner) {
            kn = *(const u32x4 *)(knew + d8 * 8);
            vn = *(const u32x4 *)(vnew + d8 * 8);
        }
        float sc[PASSES];
        float mt = -INFINITY;
#pragma unroll
        for (int it = 0; it < PASSES; it++) {
            const int t = tb + it * RP + tsub;
            const bool isnew = owner && t == pos;
            const float v = scores8(isnew ? kn : K[it]);
            sc[it] = t < t_end ? v : -INFINITY;
            mt = fmaxf(mt, sc[it]);
        }
        mt = att_rows_max(mt);
        const float Mn = fmaxf(M, mt), alpha = __builtin_amdgcn_exp2f(M - Mn);
        float p[PASSES], ps = 0.f;
#pragma unroll
        for (int it = 0; it < PASSES; it++) {
            p[it] = __builtin_amdgcn_exp2f(sc[it] - Mn);         // masked slots hold -inf: 0
            ps += p[it];
        }
        l = __builtin_fmaf(l, alpha, ps);
#pragma unroll
        for (int j = 0; j < 8; j++) av[j] *= alpha;
#pragma unroll
        for (int it = 0; it < PASSES; it++) {
            const bool isnew = owner && tb + it * RP + tsub == pos;
            const half8_t v8 = __builtin_bit_cast(half8_t, isnew ? vn : V[it]);
#pragma unroll
            for (int j = 0; j < 8; j++) av[j] = __builtin_fmaf(p[it], (float)v8[j], av[j]);
        }
        M = Mn;
    };
    auto tile = [&](const u32x4 (&K)[NP], const u32x4 (&V)[NP], int tb) {
        if (tb + TILE <= n_old) tile_full(K, V);                 // all of its rows are cache rows of this split
        else tile_last(K, V, tb, std::integral_constant<int, NP>());
    };
    if (tiny) {
        tile_last(KA, VA, t0, std::integral_constant<int, 2>());
    } else if (ntiles == 1) {    // every context up to a tile: one register set, no loop
        tile(KA, VA, t0);
    } else {
        for (int i = 0; i < ntiles; i += 2) {
            load(KB, VB, t0 + (i + 1) * TILE);                   // (past the split's rows: zeros, no traffic)
            __builtin_amdgcn_sched_barrier(0);
            tile(KA, VA, t0 + i * TILE);
            __builtin_amdgcn_sched_barrier(0);
            load(KA, VA, t0 + (i + 2) * TILE);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < ntiles) tile(KB, VB, t0 + (i + 1) * TILE);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (owner && tid < ATT_HD / 2) {   // append the new token's row to the cache (nothing in this launch reads it back: it came from LDS), after
        half_t *kd = a.kc + slice + (size_t)pos * hd + tid;   // the last cache load of this workgroup has been consumed
        half_t *vd = a.vc + slice + (size_t)pos * hd + tid;
        kd[0] = nk0;
        kd[ATT_HD / 2] = nk1;
        vd[0] = nv0;
        vd[ATT_HD / 2] = nv1;
    }

    // ---- the wave's four rows -> one, the NW waves -> one (LDS): {Mx, den, num[tid]} of the split ----
#pragma unroll
    for (int j = 0; j < 8; j++) av[j] = att_rows_sum(av[j]);
    l = att_rows_sum(l);
    if (lane < 16) {
        *(float4_t *)(&accs[wave][lane * 8]) = float4_t{av[0], av[1], av[2], av[3]};
        *(float4_t *)(&accs[wave][lane * 8 + 4]) = float4_t{av[4], av[5], av[6], av[7]};
    }
    if (lane == 0) {
        mw[wave] = M;
        lw[wave] = l;
    }
    __syncthreads();
    float Mx = mw[0];
#pragma unroll
    for (int w = 1; w < NW; w++) Mx = fmaxf(Mx, mw[w]);
    float num = 0.f, den = 0.f;
    if (tid < ATT_HD) {
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const float e = __builtin_amdgcn_exp2f(mw[w] - Mx);
            num = __builtin_fmaf(e, accs[w][tid], num);
            den = __builtin_fmaf(e, lw[w], den);
        }
    }
    const half_t o = attn_round_f16(num * __builtin_amdgcn_rcpf(den));   // the split's normalised output (one split: the row itself)
    if constexpr (REC) {     // the consumer merges (also a single split: it always reads records)
        if (tid < ATT_HD) a.o16[((size_t)b * S + s) * hd + h * ATT_HD + tid] = o;
        if (tid == 0) *(float2 *)(a.md + (((size_t)b * S + s) * a.heads + h) * 2) = float2{Mx, den};
        return;
    }
    if (sp.nsp == 1) {       // this workgroup is the whole head
        if (tid < ATT_HD) a.out[(size_t)b * a.ldo + ocol] = o;
        return;
    }
    // ---- several splits and no merging consumer: system-scope records, arrival ticket, the last split merges ----
    half_t *ro = a.o16 + ((size_t)b * S) * hd + h * ATT_HD;
    float *rmd = a.md + (((size_t)b * S) * a.heads + h) * 2;
    if (tid < ATT_HD) __hip_atomic_store((uint16_t *)(ro + (size_t)s * hd + tid), __builtin_bit_cast(uint16_t, o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid == 0) {
        __hip_atomic_store(rmd + (size_t)s * a.heads * 2, Mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(rmd + (size_t)s * a.heads * 2 + 1, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned *ticket = a.tickets + (size_t)b * a.heads + h;
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(sp.nsp - 1));
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    if (tid < ATT_HD) {      // the records of all splits requested together (one memory latency), merged in split order
        float mi[ATT_MAX_SPLITS], di[ATT_MAX_SPLITS];
        half_t oi[ATT_MAX_SPLITS];
#pragma unroll
        for (int i = 0; i < ATT_MAX_SPLITS; i++) {
            const int ii = min(i, sp.nsp - 1);
            mi[i] = __hip_atomic_load(rmd + (size_t)ii * a.heads * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            di[i] = __hip_atomic_load(rmd + (size_t)ii * a.heads * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint16_t u = __hip_atomic_load((const uint16_t *)(ro + (size_t)ii * hd + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            oi[i] = __builtin_bit_cast(half_t, u);
        }
        float c[ATT_MAX_SPLITS];
        attn_merge_coeffs(mi, di, sp.nsp, c);
        a.out[(size_t)b * a.ldo + ocol] = attn_round_f16(attn_merge_value(oi, c));
    }
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

// S of a launch's grid: enough workgroups to stream K / V at the chip's rate once a context is long (about 128 of them: a CU keeps ~36 KB of loads
// in flight, i.e. ~80 GB/s against a quiet HBM), never more than ATT_MAX_SPLITS (what a merging consumer reads) -- heads x rows x S ~ 128.
// GPTQ_ATTN_SPLITS pins it (A/B).
int decode_attn_grid_splits(int heads, int t_max, int batch) {
    static const int pin = [] { const char *e = getenv("GPTQ_ATTN_SPLITS"); return e ? atoi(e) : 0; }();
    static const int wgs = [] { const char *e = getenv("GPTQ_ATTN_WGS"); return e ? atoi(e) : 256; }();
    int S = pin > 0 ? pin : wgs / std::max(1, heads * batch);
    S = std::min(std::max(S, 1), ATT_MAX_SPLITS);
    return std::min(S, (t_max + ATT_TILE - 1) / ATT_TILE);
}

// tokens a split owns at least.  With a merging consumer a further split costs one more record per x piece of o_proj's staging (~0.1 us): cut as
// soon as a tile is full.  With the in-kernel merge it costs the ticket's dependent round trips (~4-5 us): one workgroup per head up to
// ~768 tokens.  GPTQ_ATTN_TPS_REC / GPTQ_ATTN_TPS override (A/B).
int decode_attn_tps(bool rec) {
    static const int t_rec = [] { const char *e = getenv("GPTQ_ATTN_TPS_REC"); return e ? atoi(e) : 128; }();
    static const int t_own = [] { const char *e = getenv("GPTQ_ATTN_TPS"); return e ? atoi(e) : 768; }();
    return std::max(1, rec ? t_rec : t_own);
}

// rec: leave the records to the next launch (out is not written); tps <= 0: the default of the mode
int decode_attn_fused_launch(const half_t *qkv, const int64_t *pos, half_t *kc, half_t *vc, half_t *out, float *ws, int heads, int t_max,
                             float base, float scale, const float *rope_table, hipStream_t s, int batch, int64_t ldq, int64_t ldo,
                             const int32_t *out_perm, bool rec, int tps) {
    const int S = decode_attn_grid_splits(heads, t_max, batch);
    AttnArgs a{};
    a.qkv = qkv; a.pos = pos; a.kc = kc; a.vc = vc; a.out = out;
    a.o16 = (half_t *)ws;
    a.md = (float *)((half_t *)ws + (size_t)batch * S * heads * ATT_HD);
    a.tickets = (unsigned *)(a.md + (size_t)batch * S * heads * 2);
    a.rope_tab = (const float2 *)rope_table;
    a.out_perm = out_perm;
    a.heads = heads; a.t_max = t_max; a.ldq = (int)ldq; a.ldo = (int)ldo;
    a.tps = tps > 0 ? tps : decode_attn_tps(rec);
    a.inv_base = -2.0f * logf(base) / (float)ATT_HD;   // reference fused_attn.py:91
    a.scale2 = scale * 1.44269504088896340736f;
    const dim3 grid(heads, batch, S);
    // (four waves: eight -- two per SIMD, 256-step tiles -- measured the same within noise at every depth, gpurun_out r6a / r6b, and spill)
    if (rec) hipLaunchKernelGGL((attn_stream_kernel<4, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_stream_kernel<4, false>), grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// {cos, sin}(pos * base^(-2c / head_dim)) for pos < t_max, c < head_dim / 2: the arithmetic of the in-kernel RoPE above, once
__global__ void __launch_bounds__(64) rope_table_kernel(float2 *__restrict__ tab, int half, float inv_base) {
    const int pos = blockIdx.x, c = threadIdx.x;
    if (c >= half) return;
    const float freq = expf((float)c * inv_base) * (float)pos;
    tab[(size_t)pos * half + c] = float2{cosf(freq), sinf(freq)};
}

int rope_table_launch(float *table, int t_max, int head_dim, float base, hipStream_t s) {
    const float inv_base = -2.0f * logf(base) / (float)head_dim;
    hipLaunchKernelGGL(rope_table_kernel, dim3(t_max), dim3(64), 0, s, (float2 *)table, head_dim / 2, inv_base);
    return (int)hipGetLastError();
}

// records of the streaming kernel [batch][S][heads * 128] fp16 + [batch][S][heads][2] fp32 + tickets [batch][heads]; never less than the records of the two-launch
// path above (batch 1: [heads][t_max / 128][130])
size_t decode_attn_ws_bytes(int heads, int t_max, int batch) {
    const int S = decode_attn_grid_splits(heads, t_max, batch);
    const size_t stream = (size_t)batch * ((size_t)S * heads * (ATT_HD * sizeof(half_t) + 2 * sizeof(float)) + (size_t)heads * sizeof(unsigned));
    const size_t two = batch == 1 ? (size_t)heads * ((t_max + ATT_TS - 1) / ATT_TS) * ATT_REC * sizeof(float) : 0;
    return std::max(stream, two);
}

}  // namespace gptq
