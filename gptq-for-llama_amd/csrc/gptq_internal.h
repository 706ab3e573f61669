// gptq_internal.h -- declarations shared by the kernel translation units and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/gptq_mi355x.h"
#include "attn_split.h"
#include "gptq_device.h"

#include <mutex>

namespace gptq {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the DEVICE's copy of the code object and must never shrink under a
// concurrent launch: one state per (kernel instantiation, device), raised monotonically under a mutex (round-2 advisor finding:
// a per-process "configured" word let the second device of a multi-GPU process launch > 48 KB of LDS without the opt-in).
constexpr int GPTQ_MAX_DEVICES = 32;
struct LdsOptIn {
    std::mutex mu;
    size_t bytes[GPTQ_MAX_DEVICES] = {};
    int ensure(const void *kern, size_t lds) {
        if (lds <= 48 * 1024) return 0;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
        std::lock_guard<std::mutex> lock(mu);
        const bool known = dev >= 0 && dev < GPTQ_MAX_DEVICES;
        if (!known || lds > bytes[dev]) {
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            if (known) bytes[dev] = lds;
        }
        return 0;
    }
};


constexpr int GEMV_MAX_M = 4;         // rows served by the wavefront-reduction GEMV
constexpr int SKINNY_MAX_M = 64;      // rows served by the weight-streaming MFMA kernel
constexpr int GEMV_NUM_VARIANTS = 3;   // packed rows in flight per wave: variant v -> U = 8 >> v
// split-K workspace: one 64-bit word per output element ([M][N]), all-zero between launches
constexpr size_t WS_BYTES = (size_t)SKINNY_MAX_M * 32768 * 8;
// workspace layout: [0, 512 KiB) one u64 word per output element of the rowwave GEMV combine ([M <= 4][N]) | 4 KiB of per-tile
// arrival tickets of the stream kernel (u32, zero between launches) | the stream kernel's partial tiles
constexpr size_t SPLITK_TICKET_OFFSET = 65536 * 8;   // 512 KiB of combine words: [M <= 4][N] for the small-batch rowwave
constexpr size_t SPLITK_PART_OFFSET = SPLITK_TICKET_OFFSET + 4096;

struct GemvParams {
    const half_t *x;
    int64_t ldx;
    const uint32_t *qw[2];
    const half_t *sc[2];
    const int32_t *qz[2];
    const int32_t *gi[2];
    const half_t *bias;
    half_t *y;
    int64_t ldy;
    int M, K, N, G, groupsize;
    int ntiles, split_k, nchunks, chunks_per_slice;
    int units_per_group, upg_shift;  // stream kernel: groupsize / unit_k and its log2 (or -1)
    u64_t *ws;
    const int32_t *xperm;  // non-NULL: x is read as x[xperm[k]] (act-order layer re-sorted at load; M == 1 rowwave only)
    const half_t *norm_w;  // non-NULL: RMS-normalise x on the fly with this weight (M == 1 rowwave only)
    float norm_eps;
    u64_t *dbg;  // optional timeline buffer [blocks][waves][8] (tools/timeline.py), else nullptr
    float *y32 = nullptr;  // generic kernel only (round 6): the fp32 sums leave unrounded, [M][ldy] -- the partial product of a ROW shard (no bias, single set)
};

int gemv_fast_dispatch(int bits, bool fused2, int u, const GemvParams &p, hipStream_t s);
int gemv_rowwave_mr_dispatch(bool fused2, int u, const GemvParams &p, hipStream_t s);   // 2 <= M <= 4, 4-bit, one launch
int gemv_rowwave_mfma_dispatch(bool fused2, int u, const GemvParams &p, hipStream_t s); // 2 <= M <= 8, 4-bit, MFMA 4x4x4
int gemv_generic_dispatch(int bits, bool fused2, int nl, const GemvParams &p, hipStream_t s);

// skinny MFMA (weight streaming, M <= 64) and tiled MFMA GEMM (prefill)
int skinny_dispatch(int bits, bool fused2, int stg, int waves, bool xlds, const GemvParams &p, hipStream_t s);
int gemm_dispatch(int bits, bool pair, const GemvParams &p, hipStream_t s);   // pair: fused gate/up in ONE launch, SiLU on the fp32 sums
int gemm_set_version(int v);   // 2 = ping-pong kernel (default), 3 = all-DMA packed-B kernel; returns the previous value
int transpose_dispatch(int bits, const half_t *dy, int64_t lddy, const uint32_t *qw, const half_t *sc,
                       const int32_t *qz, const int32_t *gi, half_t *dx, int64_t lddx, int M, int K, int N,
                       int G, int groupsize, hipStream_t s);

int rmsnorm_launch(const half_t *x, int64_t ldx, const half_t *w, half_t *y, int64_t ldy, int M, int N,
                   float eps, hipStream_t s);
int rope_launch(half_t *qk, int64_t row_stride, const int64_t *pos, int64_t pos_batch_stride, int bsz,
                int seq, int heads, int head_dim, float base, hipStream_t s);
int pack_launch(const float *weight, const float *scales, const float *zeros, const int32_t *g_idx, int K,
                int N, int G, int bits, int groupsize, int32_t *qweight, int32_t *qzeros, half_t *scales16,
                hipStream_t s);
int dequant_launch(const uint32_t *qw, const half_t *sc, const int32_t *qz, const int32_t *gi, int K, int N, int G, int groupsize,
                   int bits, half_t *out, int64_t ldo, hipStream_t s);
int dequant_t_launch(const uint32_t *qw, const half_t *sc, const int32_t *qz, const int32_t *gi, int K, int N, int G, int groupsize,
                     int bits, half_t *out, int64_t ldo, hipStream_t s);   // the same weight as [N][K] (k contiguous)
// gemm8.hip: the hand-written prefill GEMM: c = x . wt^T (+ bias) / silu(x . wt_gate^T) * (x . wt_up^T); GPTQ_E_VARIANT when not served
int gemm8_set_tile(int rows);   // 0 (per launch), 192, 256; returns the previous value
int gemm8_set_mfma(int shape);   // 16 (16x16x32) or 32 (32x32x16); returns the previous value
int gemm8_dense_f16(const half_t *x, int64_t ldx, const half_t *wt, int64_t ldw, const half_t *bias, half_t *c, int64_t ldc, int M, int K, int N,
                    bool pair, hipStream_t s);
// dense_gemm.hip: y = x . W (+ bias) through hipBLASLt (dlopen'ed), plans cached per shape
bool dense_gemm_available();
int dense_gemm_plan_count();   // plans currently cached (bounded LRU)
int dense_gemm_set_enabled(int on);   // test hook: 0 = behave as if hipBLASLt were absent; returns the previous value
int dense_gemm_f16(const half_t *x, int64_t ldx, const half_t *W, int64_t ldw, const half_t *bias, void *y, int64_t ldy, int M, int K, int N,
                   void *ws, size_t ws_bytes, hipStream_t s, bool trans_w = false, bool out_f32 = false);
// dense_gemv.hip: y[N] = W[N][K] . x (one row of x, dense fp16 weight stored [out, in]; optional RMSNorm of x in front, optional bias)
int dense_gemv_launch(const half_t *x, const half_t *W, int64_t ldw, const half_t *bias, half_t *y, int N, int K, const half_t *norm_w, float eps,
                      hipStream_t s, int M = 1, int64_t ldx = 0, int64_t ldy = 0);   // M <= 16 rows of x share ONE pass over W
int silu_mul_launch(const half_t *g, int64_t ldg, const half_t *u, int64_t ldu, half_t *c, int64_t ldc, int M, int N, hipStream_t s);
int silu_mul_f32_launch(const float *g, int64_t ldg, const float *u, int64_t ldu, half_t *c, int64_t ldc, int M, int N, hipStream_t s);
int gather_cols_launch(const half_t *x, int64_t ldx, const int32_t *perm, half_t *xg, int64_t ldg, int M, int K, hipStream_t s);   // xg = x[:, perm]
int add_rows_launch(half_t *y, int64_t ldy, const half_t *r, int64_t ldr, int M, int N, hipStream_t s);   // y = fp16(y + r)
int slices_combine_norm_launch(const float *partials, int S, int M, int N, const half_t *add, int64_t ldb, half_t *y, int64_t ldy, const half_t *nw, float eps,
                               half_t *h, int64_t ldh, hipStream_t s);   // round 6: [S][16][N] fp32 rows -> y (+ residual / bias), h = rmsnorm(y) * nw
int dirty_lds_launch(uint32_t pattern, hipStream_t s);   // test support: every CU's LDS overwritten
int act_order_repack_launch(const uint32_t *qw, const int32_t *perm, int K, int N, int bits, uint32_t *out, hipStream_t s);
int gidx_trivial_launch(const int32_t *g_idx, int K, int groupsize, int32_t *out, hipStream_t s);

// ---- stripe16: no-split-K GEMV on a load-time repacked copy (stripe.hip) ----
struct StripeParams {
    const half_t *x;
    int64_t ldx, ldy;      // row strides of x / y (elements); M = 1: unused
    const uint32_t *R;     // [N/16][K/(16 KPW)][NS][64][4]
    const uint32_t *tab;   // half2 [N/16][NS][G][16] {scale, zero + 1}
    half_t *y;
    const half_t *bias;
    int64_t ldb;           // 0: bias[N] added to every row; != 0: `bias` is a residual [M][ldb] added row by row (decode kernel, 16-row tiles)
    const half_t *norm_w;  // non-NULL: RMS-normalise x while it is staged (decode kernel, every row its own rstd)
    float norm_eps;
    const uint16_t *xperm; // non-NULL: x (and norm_w) gathered through this permutation (M == 1); uint16: K <= 24576 on this path
    float *y32;            // non-NULL: store the fp32 sums [M][NS][N] here instead of fp16 y (no bias): partial of a K-sharded layer (M <= 4)
    int M, K, N, G, NS, gq_shift, bits;
    uint32_t *progress;    // non-NULL: the decode kernel adds 1 here when it starts (debug hook gptq_set_progress_counter)
    const int32_t *yperm;  // non-NULL: column n of y is stored at yperm[n] (the consumer's sorted order: decode kernel only)
    AttnMerge att;         // att.o16 non-NULL (round 6): x is the decode attention's split records, merged while x is staged (M == 1, plain launch)
    // round 6 (16-row tiles with K slices, <= 16 rows, one set): the combine launch also writes h[M][ldh] = rmsnorm(y) * next_norm_w and sets
    // *next_norm_done = 1; any other route ignores these fields (the caller then runs the norm itself)
    const half_t *next_norm_w;
    float next_norm_eps;
    half_t *h;
    int64_t ldh;
    int *next_norm_done;
};
int stripe_gq_shift(int K, int N, int bits, int groupsize);            // log2(groupsize / (4 KPW)), -1 one group, -2 ineligible
size_t stripe_tab_offset(int K, int N, int bits, int nsets);
size_t stripe_total_bytes(int K, int N, int bits, int groupsize, int nsets);
// perm != NULL (bits 2 / 4 / 8): the image of the GROUP-SORTED rows of a regular act-order layer, gathered straight from the checkpoint layout
int stripe_repack_launch(const uint32_t *qw0, const half_t *sc0, const int32_t *qz0, const uint32_t *qw1, const half_t *sc1,
                         const int32_t *qz1, void *out, int K, int N, int bits, int groupsize, hipStream_t s, const int32_t *perm = nullptr);
int stripe_gemv_dispatch(const StripeParams &p, hipStream_t s);
uint32_t *stripe_progress_counter();   // capi.hip: gptq_set_progress_counter (NULL by default)
// inverse of stripe_repack_launch for ONE set (bits 2 / 4 / 8): qweight [K/32*bits][N], scales [G][N], qzeros [G][N/32*bits]
// (invperm != NULL: the image holds group-sorted rows; the ORIGINAL row order comes back)
int stripe_unpack_launch(const void *image, int K, int N, int bits, int groupsize, int nsets, int set, uint32_t *qw, half_t *sc, int32_t *qz,
                         hipStream_t s, const int32_t *invperm = nullptr);

// round 5: Wt[set * N + n][k] fp16 (k contiguous: the operand of gemm8) straight from an image with trivial g_idx; bit-identical to dequant_t_launch
int stripe_dequant_t_launch(const void *image, int K, int N, int bits, int groupsize, int nsets, half_t *out, int64_t ldo, hipStream_t s);

int gptq_block_launch(const float *W, int64_t ldw, const float *Hinv, int64_t ldh, int rows, int i1, int count, int groupsize, int maxq,
                      const float *scale, const float *zero, int64_t ldg, float *Q, int64_t ldq, float *Err, int64_t lde, float *loss_rows,
                      hipStream_t s);

int decode_rope_kv_launch(half_t *qkv, const int64_t *pos, half_t *kc, half_t *vc, int heads, int head_dim, int t_max, float base,
                          hipStream_t s);
int decode_attn_launch(const half_t *q, const half_t *kc, const half_t *vc, const int64_t *pos, half_t *out, float *ws, int heads,
                       int t_max, float scale, hipStream_t s);
size_t decode_attn_ws_bytes(int heads, int t_max, int batch = 1);
// batch rows: pos[batch], qkv rows ldq apart, out rows ldo apart, kc / vc [batch][t_max][heads * 128], ws of decode_attn_ws_bytes(heads, t_max, batch)
// rec: the records {M, den, num[128]} of every active split are left in ws for the next launch to merge (out is not written); tps <= 0: default
int decode_attn_fused_launch(const half_t *qkv, const int64_t *pos, half_t *kc, half_t *vc, half_t *out, float *ws, int heads, int t_max,
                             float base, float scale, const float *rope_table, hipStream_t s, int batch = 1, int64_t ldq = 0,
                             int64_t ldo = 0, const int32_t *out_perm = nullptr, bool rec = false, int tps = 0);
int decode_attn_grid_splits(int heads, int t_max, int batch);   // S of the launch grid (and of the workspace layout)
int decode_attn_tps(bool rec);                                  // default tokens per split of the mode
int rope_table_launch(float *table, int t_max, int head_dim, float base, hipStream_t s);

}  // namespace gptq
