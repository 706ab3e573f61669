// decode_attn.hip -- the two non-GEMV pieces of a batch-1 decode step that the reference does with
// torch ops between its Triton kernels (quant/fused_attn.py:126-155): RoPE on q,k + KV-cache
// append (triton_rotate_half_ :126, torch.cat :142-143) and the single-query attention
// (F.scaled_dot_product_attention :155).  Written as plain HIP so that the whole decode step can
// be captured in ONE hipGraph: the current position is read from device memory, the KV cache is
// a preallocated [t_max, heads*head_dim] buffer (no torch.cat, no shape change per token).
//
// HBM-bound elementwise / reduction work (K,V rows are read once per token): no MFMA.
#include <cstdlib>

#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

int decode_attn_ts_grid(int t_max, int batch);

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
constexpr int ATT_LONG = 1024; // contexts above this many tokens run 64-step splits when the grid was launched for them (batch 1)
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
// One launch per layer: RoPE(q, k) + KV append + single-query attention + split merge.
// Every active split rotates q itself (128 values); the split that owns row `pos` also rotates k,
// appends k,v to the cache and takes that row from LDS (no read-after-write through memory).
// With more than one active split the partial records are published with system-scope
// (write-through) stores, drained, and a per-head arrival ticket elects the last split to merge
// them (MI355X_MICROARCH.md, "Valid forms": sc0 sc1 stores AND loads, flag behind vmcnt(0)).
// ws = [batch][heads][nsplit][ATT_REC] floats followed by [batch][heads] uint32 tickets (zero between launches).
// Round 5: blockIdx.y = row of a decode BATCH (blockIdx.z = split: see the note on the dispatch order below) -- every row has its own position (sequences of different lengths; a left-padded prompt is
// stored without its pads, see quant/engine_hook.py), its own [t_max][heads * 128] slice of the K / V cache, its own qkv row (stride ldq)
// and output row (stride ldo).  A negative position marks an idle row: nothing is read or written for it.
// ---------------------------------------------------------------------------------------
// Dispatch order (round 5): workgroups are issued x fastest, then y, then z.  With grid (heads, splits, rows) the ONE active split of a short
// context sat between 15 idle ones per row -- a batch of 16 rows at t_max = 2048 issues 8 192 workgroups of which 7 680 load `pos` and leave, and row 15's
// active workgroups came after 7 680 others (four rounds of resident workgroups, each a memory round trip: 8.1 us per launch at 16 rows against 5.1
// at one).  Grid (heads, rows, splits): every row's FIRST split is issued first, the idle ones drain behind the work.
__global__ void __launch_bounds__(256) attn_decode_fused_kernel(const half_t *__restrict__ qkv, const int64_t *__restrict__ pos_ptr,
                                                                half_t *__restrict__ kc, half_t *__restrict__ vc,
                                                                half_t *__restrict__ out, float *__restrict__ ws, int heads, int t_max,
                                                                float inv_base, float scale, const float2 *__restrict__ rope_tab,
                                                                u64_t *__restrict__ dbg, int ldq, int ldo, int ts_grid, const int32_t *__restrict__ out_perm) {
    {   // this workgroup's row of the batch
        const int b = blockIdx.y;
        const size_t hdz = (size_t)heads * ATT_HD;
        pos_ptr += b;
        qkv += (size_t)b * ldq;
        out += (size_t)b * ldo;
        kc += (size_t)b * t_max * hdz;
        vc += (size_t)b * t_max * hdz;
        if (b) dbg = nullptr;   // (development stamps: row 0 only)
    }
    float *const ws_tickets = ws + (size_t)gridDim.y * heads * gridDim.z * ATT_REC;
    ws += (size_t)blockIdx.y * heads * gridDim.z * ATT_REC;
    u64_t st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64_t sx_[4] = {0, 0, 0, 0};   // development stamps (gptq_set_debug_buffer, tools/timeline_attn.py)
    if (dbg) { st_[0] = stamp_realtime(); st_[1] = stamp_cycles(0); }
    __shared__ float qs[ATT_HD];
    __shared__ __attribute__((aligned(16))) half_t knew[ATT_HD];
    __shared__ __attribute__((aligned(16))) half_t vnew[ATT_HD];
    __shared__ float sc[ATT_TS];
    __shared__ float accs[4][ATT_HD];
    __shared__ int last_flag;
    const int h = blockIdx.x, nsplit = gridDim.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // out_perm (round 5): o_proj is an act-order layer whose image holds group-sorted rows -- element k of the attention output goes where its sorted
    // order wants it, so o_proj runs the trivial kernel.  Requested first: nothing depends on it until the store.
    const int ocol = (tid < ATT_HD) ? (out_perm ? out_perm[h * ATT_HD + tid] : h * ATT_HD + tid) : 0;
    const int64_t pos = pos_ptr[0];
    if (dbg) st_[2] = stamp_cycles((uint32_t)pos);
    if (pos < 0 || pos >= t_max) return;
    const int len = (int)pos + 1;
    // Round 5: the split length can be chosen at RUN time.  The grid is fixed when the step is captured (ts_grid = 128, or -- an A/B knob, see
    // decode_attn_ts_grid: measured, slower, off by default -- 64 for a batch-1 launch: twice the workgroups); with the 64-step grid a launch whose
    // context is at most ATT_LONG tokens folds two grid splits into one 128-step split (odd grid splits leave at once), a longer one keeps
    // 64-step splits: 24 workgroups per head stream K / V at 1500 tokens instead of 12.
    int s = blockIdx.z, ts = ATT_TS;
    if (ts_grid == ATT_TS / 2) {
        if (len > ATT_LONG) ts = ATT_TS / 2;
        else if (s & 1) return;
        else s >>= 1;
    }
    const int t0 = s * ts;
    if (t0 >= len) return;
    const int nact = (len - t0) < ts ? (len - t0) : ts;
    const int nsp = (len + ts - 1) / ts;                 // active splits of this head
    const bool own_new = (int)pos >= t0 && (int)pos < t0 + ts;
    const int hd = heads * ATT_HD;
    const int tnew = own_new ? (int)pos - t0 : -1;
    const int d8 = tid & 15, tsub = tid >> 4;

    // every cache row this thread will need is requested NOW (they depend on nothing but pos), so the
    // RoPE trig below and the two softmax barriers run under the memory latency instead of after it
    half8_t kpre[ATT_TS / 16];
#pragma unroll
    for (int it = 0; it < ATT_TS / 16; it++) {
        const int tl = it * 16 + tsub;
        kpre[it] = (tl < nact && tl != tnew) ? *(const half8_t *)(kc + (size_t)(t0 + tl) * hd + (size_t)h * ATT_HD + d8 * 8)
                                              : (half8_t)(half_t)0;
    }
    half8_t vpre[ATT_TS / 16];   // same (timestep, 8-dim slice) map as K: 16-byte loads, 4 rows per wave instruction
#pragma unroll
    for (int it = 0; it < ATT_TS / 16; it++) {
        const int tl = it * 16 + tsub;
        vpre[it] = (tl < nact && tl != tnew) ? *(const half8_t *)(vc + (size_t)(t0 + tl) * hd + (size_t)h * ATT_HD + d8 * 8)
                                              : (half8_t)(half_t)0;
    }

    half_t nk0 = (half_t)0, nk1 = (half_t)0, nv0 = (half_t)0, nv1 = (half_t)0;   // the new token's K / V halves of this thread
    if (tid < ATT_HD / 2) {
        const int c = tid;
        float cs, sn;
        if (rope_tab) {   // {cos, sin} of (pos, c) from the table rope_table_kernel filled with the SAME instructions: one load under
            const float2 e = rope_tab[(size_t)pos * (ATT_HD / 2) + c];   // the K/V latency instead of ~1 us of accurate-libm range reduction
            cs = e.x;
            sn = e.y;
        } else {
            const float freq = expf((float)c * inv_base) * (float)pos;
            cs = cosf(freq);
            sn = sinf(freq);
        }
        const half_t *q = qkv + (size_t)h * ATT_HD + c;
        const float qx = (float)q[0], qy = (float)q[ATT_HD / 2];
        qs[c] = (float)(half_t)(qx * cs - qy * sn);       // rounded to fp16 like the in-place reference RoPE
        qs[c + ATT_HD / 2] = (float)(half_t)(qx * sn + qy * cs);
        if (own_new) {
            const half_t *k = qkv + hd + (size_t)h * ATT_HD + c;
            const half_t *v = qkv + 2 * hd + (size_t)h * ATT_HD + c;
            const float kx = (float)k[0], ky = (float)k[ATT_HD / 2];
            const half_t k0 = (half_t)(kx * cs - ky * sn), k1 = (half_t)(kx * sn + ky * cs);
            nk0 = k0; nk1 = k1; nv0 = v[0]; nv1 = v[ATT_HD / 2];
            knew[c] = nk0;
            knew[c + ATT_HD / 2] = nk1;
            vnew[c] = nv0;
            vnew[c + ATT_HD / 2] = nv1;
            // the cache row itself is written AFTER the last load of this launch has been consumed (below): hipcc waits vmcnt(0)
            // before every use of the prefetched rows (they sit behind branches), and a store in flight made each of those waits
            // a full write latency -- 1500 of the launch's 8000 cycles (tools/timeline_attn.py)
        }
    }
    __syncthreads();
    if (dbg) st_[3] = stamp_cycles(0);
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; j++) qf[j] = qs[d8 * 8 + j];
    if (dbg) sx_[0] = stamp_cycles(__builtin_bit_cast(uint32_t, qf[7]));
    const int nit = (nact + 15) / 16;   // 16-timestep groups that hold anything: a short context does not pay for 128 rows
#pragma unroll
    for (int it = 0; it < ATT_TS / 16; it++) {
        const int tl = it * 16 + tsub;
        if (it >= nit) {                // wave-uniform
            if (d8 == 0) sc[tl] = -INFINITY;
            continue;
        }
        float dot = 0.f;
        if (tl < nact) {
            half8_t k8 = kpre[it];
            if (tl == tnew) k8 = *(const half8_t *)(knew + d8 * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) dot += qf[j] * (float)k8[j];
        }
        dot = att_sum16(dot);
        if (d8 == 0) sc[tl] = (tl < nact) ? dot * scale : -INFINITY;
    }
    if (dbg) sx_[1] = stamp_cycles(0);
    __syncthreads();
    if (dbg) st_[4] = stamp_cycles(0);
    // softmax statistics WITHOUT another barrier (round 4; there were three more here: max through red[], the probabilities written back over
    // sc[], their sum through red[]): every wave reads all ATT_TS scores (two per lane), reduces max and sum of exp on its own with DPP /
    // permlane steps -- four redundant copies of 128 values cost less than one LDS round trip + barrier -- and the P V loop below turns the
    // scores it needs into probabilities itself (eight v_exp per thread).  Empty slots hold -inf: exp -> 0.
    const float s0 = sc[lane], s1 = sc[lane + 64];
    const float m = att_wave_max(fmaxf(s0, s1));
    const float l = att_wave_sum(__expf(s0 - m) + __expf(s1 - m));
    if (dbg) st_[5] = stamp_cycles(__builtin_bit_cast(uint32_t, l));

    float av[8];
#pragma unroll
    for (int j = 0; j < 8; j++) av[j] = 0.f;
#pragma unroll
    for (int it = 0; it < ATT_TS / 16; it++) {
        const int tl = it * 16 + tsub;
        if (it < nit && tl < nact) {
            half8_t v8 = vpre[it];
            if (tl == tnew) v8 = *(const half8_t *)(vnew + d8 * 8);
            const float pt = __expf(sc[tl] - m);
#pragma unroll
            for (int j = 0; j < 8; j++) av[j] += pt * (float)v8[j];
        }
    }
    if (dbg) sx_[2] = stamp_cycles(__builtin_bit_cast(uint32_t, av[0]));
    if (own_new && tid < ATT_HD / 2) {   // append the new token's row to the cache (nothing in this launch reads it back: it came from LDS)
        half_t *kd = kc + (size_t)pos * hd + (size_t)h * ATT_HD + tid;
        half_t *vd = vc + (size_t)pos * hd + (size_t)h * ATT_HD + tid;
        kd[0] = nk0;
        kd[ATT_HD / 2] = nk1;
        vd[0] = nv0;
        vd[ATT_HD / 2] = nv1;
    }
    // the 4 timestep groups of a wave (lane >> 4), then the 4 waves through LDS
#pragma unroll
    for (int j = 0; j < 8; j++) {
        av[j] = att_rows_sum(av[j]);
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 8; j++) accs[wave][lane * 8 + j] = av[j];
    }
    __syncthreads();
    float acc = 0.f;
    if (tid < ATT_HD) acc = accs[0][tid] + accs[1][tid] + accs[2][tid] + accs[3][tid];

    if (nsp == 1) {  // short context: this workgroup is the whole head
        if (tid < ATT_HD) out[ocol] = (half_t)(acc / l);
        if (dbg && lane == 0) {
            st_[6] = stamp_cycles(__builtin_bit_cast(uint32_t, acc));
            const u64_t te = stamp_realtime();
            u64_t *d = dbg + ((size_t)h * 4 + wave) * 10;
#pragma unroll
            for (int i = 0; i < 7; i++) d[i] = st_[i];
            d[8] = te;
            d[7] = sx_[0] - st_[1]; d[9] = ((sx_[1] - st_[1]) << 32) | (uint32_t)(sx_[2] - st_[1]);
        }
        return;
    }
    float *rec = ws + ((size_t)h * nsplit + s) * ATT_REC;
    if (tid < ATT_HD) __hip_atomic_store(rec + 2 + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid == 0) {
        __hip_atomic_store(rec + 0, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(rec + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned *ticket = (unsigned *)ws_tickets + (size_t)blockIdx.y * heads + h;
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(nsp - 1));
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    if (tid < ATT_HD) {
        // the records of up to 16 splits are requested TOGETHER (48 independent system-scope loads, one memory latency): fetched one
        // after the other, as the first version did, the merge cost 2 nsp dependent round trips -- 13 us of the 18.6 us this launch
        // took at 1500 tokens of context.  Running {max, num, den} rescaled between batches of 16 (t_max > 2048).
        const float *base = ws + (size_t)h * nsplit * ATT_REC;
        float Mx = -INFINITY, num = 0.f, den = 0.f;
        for (int i0 = 0; i0 < nsp; i0 += 16) {
            float mi[16], li[16], ai[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float *r = base + (size_t)min(i0 + i, nsp - 1) * ATT_REC;
                mi[i] = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                li[i] = __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ai[i] = __hip_atomic_load(r + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            float mb = Mx;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (i0 + i < nsp) mb = fmaxf(mb, mi[i]);
            const float resc = __expf(Mx - mb);   // 0 for the first batch (Mx = -inf), then the usual running-softmax rescale
            num *= resc;
            den *= resc;
            Mx = mb;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (i0 + i < nsp) {
                    const float w = __expf(mi[i] - Mx);
                    num += w * ai[i];
                    den += w * li[i];
                }
        }
        out[ocol] = (half_t)(num / den);
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

int decode_attn_fused_launch(const half_t *qkv, const int64_t *pos, half_t *kc, half_t *vc, half_t *out, float *ws, int heads, int t_max,
                             float base, float scale, const float *rope_table, u64_t *dbg, hipStream_t s, int batch, int64_t ldq, int64_t ldo,
                             const int32_t *out_perm) {
    // batch 1 and a cache that can hold a long context: a grid of 64-step splits (the kernel folds pairs of them below ATT_LONG tokens)
    const int ts_grid = decode_attn_ts_grid(t_max, batch);
    const int nsplit = (t_max + ts_grid - 1) / ts_grid;
    const float inv_base = -2.0f * logf(base) / (float)ATT_HD;
    hipLaunchKernelGGL(attn_decode_fused_kernel, dim3(heads, batch, nsplit), dim3(256), 0, s, qkv, pos, kc, vc, out, ws, heads, t_max, inv_base,
                       scale, (const float2 *)rope_table, dbg, (int)ldq, (int)ldo, ts_grid, out_perm);
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

int decode_attn_ts_grid(int t_max, int batch) {
    // MEASURED AND OFF (gpurun_out r5f / profiles/r5*/engine_context.txt, tok/s of the 7B engine at 0 / 500 / 1000 / 1500 / 1900 tokens of context):
    // 64-step splits 922 / 784 / 746 / 699 / 671 against 925 / 795 / 763 / 749 / 720 for 128-step splits -- twice the workgroups per head mean twice
    // the records in the merge and twice the tickets, and the K / V stream of a head was not short of parallelism: the launch is bound by the
    // dependent round trips of the merge, not by the splits.  GPTQ_ATTN_LONG_SPLITS=1 switches the 64-step grid on for A/B runs.
    static const int long_splits = [] { const char *e = getenv("GPTQ_ATTN_LONG_SPLITS"); return e ? atoi(e) : 0; }();
    return (batch == 1 && t_max > ATT_LONG && long_splits) ? ATT_TS / 2 : ATT_TS;
}

size_t decode_attn_ws_bytes(int heads, int t_max, int batch) {
    const int ts = decode_attn_ts_grid(t_max, batch);    // (the two-launch path below uses 128-step splits: never more records than this)
    return (size_t)batch * ((size_t)heads * ((t_max + ts - 1) / ts) * ATT_REC * sizeof(float) + (size_t)heads * sizeof(unsigned));
}

}  // namespace gptq
