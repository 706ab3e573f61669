/*
 * gptq_mi355x.h -- C ABI of libgptq_mi355x.so: the MI355X (gfx950) native hot path behind
 * GPTQ-for-LLaMa's `quant.QuantLinear` operator family.
 *
 * Every entry point is what a binding of the reference's operator interface for this path
 * would call; the reference symbol each one stands in for is cited as
 * `file:line` into qwopqwop200/GPTQ-for-LLaMa (triton branch).  There are no torch types in
 * the signatures: plain device pointers, sizes, a HIP stream.  The caller owns every buffer;
 * the library allocates nothing, never synchronises and never throws -- all launches go to
 * `stream` and are hipGraph-capturable.
 *
 * Buffer conventions (identical to the reference checkpoint format, quant_linear.py:316-321):
 *   x        fp16 [M, K]      row stride ldx (elements), last dim contiguous
 *   qweight  int32 [K/32*bits, N]  row-major; bits in {2,4,8}: word r holds k = r*f .. r*f+f-1
 *            (f = 32/bits) of one column, field j at bit bits*j (quant_linear.py:103,109,127).
 *            bits == 3 (EXTENSION, the reference raises NotImplementedError,
 *            quant_linear.py:308-309): 32 consecutive k of a column are a dense little-endian
 *            96-bit stream over 3 consecutive rows.
 *   qzeros   int32 [G, N/32*bits]   same packing along n; stored value is zero-1 and the
 *            kernel adds 1 WITHOUT re-masking (quant_linear.py:120-121)
 *   scales   fp16 [G, N]
 *   g_idx    int32 [K] group of each k, or NULL meaning the trivial map k / groupsize
 *   bias     fp16 [N] or NULL; added after the fp16 rounding of the product
 *            (separate torch add in the reference, quant_linear.py:376)
 *   y        fp16 [M, N]      row stride ldy
 *   G = ceil(K / groupsize); groupsize == K for the reference's "-1".
 *
 * Return value: 0 on success; negative = GPTQ_E_* (bad argument, nothing launched);
 * positive = hipError_t from a launch.
 */
#ifndef GPTQ_MI355X_H
#define GPTQ_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *gptq_stream_t; /* hipStream_t */

enum {
    GPTQ_OK = 0,
    GPTQ_E_BITS = -1,      /* bits not in {2,3,4,8}             (quant_linear.py:308-309)  */
    GPTQ_E_SHAPE = -2,     /* K or N not a multiple of 32, M < 0, groupsize <= 0            */
    GPTQ_E_ALIGN = -3,     /* pointer / leading dimension alignment                          */
    GPTQ_E_NULL = -4,      /* required pointer is NULL                                       */
    GPTQ_E_WORKSPACE = -5, /* workspace too small for the requested variant                  */
    GPTQ_E_VARIANT = -6,   /* unknown / inapplicable kernel variant                          */
    GPTQ_E_NORM_WIDTH = -7,/* RMSNorm row wider than 64 KiB (triton_norm.py:59-60)           */
    GPTQ_E_LIBRARY = -8    /* prefill route: hipBLASLt not loadable, or it refused the product */
};

/* gptq_query(what) */
enum {
    GPTQ_Q_ABI_VERSION = 0,
    GPTQ_Q_GEMV_MAX_M = 1,        /* largest M of the rowwave GEMV family (4-bit: M <= 4 rows share one launch) */
    GPTQ_Q_SKINNY_MAX_M = 2,      /* largest M served by the weight-streaming MFMA kernel    */
    GPTQ_Q_WORKSPACE_BYTES = 3,   /* bytes of zero-initialised workspace split-K needs       */
    GPTQ_Q_NUM_GEMV_VARIANTS = 4,
    GPTQ_Q_STRIPE_MM_WORKSPACE_BYTES = 5   /* bytes of the (separate, scratch) workspace of gptq_stripe_matmul_f16 */
};

int gptq_query(int what);
const char *gptq_strerror(int code);

/* Dispatch override used by the tests and the sweeps: variant < 0 restores the built-in choice.
 * For the rowwave GEMV variant v selects U = 8 >> v packed rows in flight per wave (v in 0..2; 3-bit:
 * 2 >> v 32-k blocks, v in 0..1) and is refused (GPTQ_E_VARIANT) when U does not divide the group;
 * for the weight-streaming MFMA kernel the values 2, 4, 8 select the waves per workgroup.
 * gptq_set_split_k forces the number of K slices (>= 1).  Both return the previous value. */
int gptq_set_gemv_variant(int variant);
int gptq_set_split_k(int split_k);
/* Prefill GEMM kernel selection (tests / A-B measurements): 2 = ping-pong kernel (default), 3 = all-LDS-DMA
 * kernel with packed B in LDS (4-bit, groupsize % 64 == 0; measured 2-4 % slower).  Returns the previous value. */
int gptq_set_gemm_kernel(int version);
/* Prefill route behind gptq_prefill_matmul_f16 / _fused_mlp_f16 / _transpose_matmul248_f16 (tests / A-B measurements):
 * 1 (default) = the hand-written LDS-DMA + MFMA tile GEMM of csrc/gemm8.hip on the dequantised weight wherever it can run
 * (K % 128 == 0: every LLaMA shape) -- since round 4 the default route never reaches hipBLASLt for such shapes; 2 = the same (kept for
 * callers of earlier rounds); 0 = hipBLASLt only (the reported ceiling).  Returns the previous value. */
int gptq_set_prefill_route(int route);
/* Rows per pass of the 16-row MFMA tiles on the stripe16 image (gptq_stripe_matmul_f16, the 5..128-row route of gptq_layer_forward):
 * 128 (default: 65..128 rows in ONE pass over the weights) or 64 (round 2's schedule; A-B runs).  Returns the previous value. */
int gptq_set_stripe_mm_pass_rows(int rows);
/* Batches of 129 .. `rows` rows of a layer with a stripe16 image run the fused-dequantise tile GEMM on the image (csrc/stripe_mm.inc,
 * stripe_gemm_kernel: weights stay packed, no per-call dequantise pass) instead of the dense route; 0 = never (tests / A-B runs).
 * Round 5: gptq_layer_forward keeps a trivial-g_idx layer on the image only while its 128 x 128 tiles fit the chip at once
 * (ceil(M / 128) * N * nsets / 128 <= 512) or M <= 640 (gate | up pair: 1152); above, dequantise + the tile GEMM is the faster own kernel
 * (gptq_layer_route_for_shape says which; gptq_stripe_matmul_f16 itself serves every M up to `rows`).  Returns the previous value. */
int gptq_set_stripe_gemm_max_rows(int rows);
/* which engine a dense product of this shape takes under the current switch: 1 = tile GEMM, 0 = hipBLASLt (host logic only;
 * nsets = 2: gate/up pair; trans = 1: the backward product) */
int gptq_prefill_route_for(int M, int K, int N, int nsets, int trans);
/* test hooks: make every library call answer GPTQ_E_LIBRARY as if hipBLASLt were not installed (returns the previous setting);
 * number of hipBLASLt plans currently cached (bounded LRU of 64) */
int gptq_set_library_enabled(int on);
/* MFMA shape of the tile GEMM (tests / A-B runs): 16 = v_mfma_f32_16x16x32_f16, 32 = v_mfma_f32_32x32x16_f16; returns the previous value */
int gptq_set_gemm8_mfma(int shape);
/* Rows of the tile GEMM's workgroup tile: 0 = chosen per launch (the tile whose rounds of 256 workgroups cost less), 192, 256.  Returns the
 * previous value, GPTQ_E_VARIANT for anything else.  Results do not depend on it (same K order per accumulator).  Test / A-B hook. */
int gptq_set_gemm8_tile(int rows);
int gptq_prefill_plan_count(void);
/* Development aid: when non-NULL, the decode kernels write per-wave s_memtime checkpoints
 * ([block][wave][8] uint64) into this device buffer.  Returns the previous pointer. */
void *gptq_set_debug_buffer(void *device_buffer);
/* Test support: one launch that overwrites all 160 KB of LDS on every CU with (pattern ^ word index) -- the soak test interleaves it with the decode
 * launches: no kernel of this library may read LDS it has not written (tests/test_gpu_soak.py). */
int gptq_debug_dirty_lds(uint32_t pattern, gptq_stream_t stream);
/* debug hook: device uint32 that every stripe16 decode launch (M = 1) increments (one relaxed device-scope add) when it starts; NULL
 * (default) = no tick.  Used by tools/warmlab.hip to pace a run-ahead prefetcher on a second stream (measured, loses: DESIGN 3.6). */
int gptq_set_progress_counter(void *device_u32);

/*
 * y = x . deq(B) (+ bias)  -- reference matmul248() + matmul_248_kernel + the bias add in
 * QuantLinear.forward (quant/quant_linear.py:263-269, :72-137, :373-377).
 * Chooses the rowwave GEMV (M = 1; 2 <= M <= 4 at 4 bits: all rows in one launch), the weight-streaming MFMA
 * kernel (M <= 64) or the tiled MFMA GEMM (prefill).  workspace: >= gptq_query(GPTQ_Q_WORKSPACE_BYTES) bytes, zero on first
 * use, one per device and per stream of execution (its first 516 KiB -- combine words and arrival tickets --
 * are restored to zero by every launch, the rest is scratch); may be NULL, which disables the K split.
 */
int gptq_matmul248_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales,
                       const int32_t *qzeros, const int32_t *g_idx, const void *bias, void *y,
                       int64_t ldy, int M, int K, int N, int bits, int groupsize, void *workspace,
                       size_t workspace_bytes, gptq_stream_t stream);

/* Round 6: the fp32 partial product of a ROW SHARD of a layer with ANY g_idx -- what a tensor-parallel rank computes for o_proj / down_proj of an
 * act-order checkpoint (the reference has no tensor parallelism: llama.py:328-382 places whole layers; north_star: K-shards, fp32 partials, ONE
 * all-reduce and ONE fp16 rounding per linear).  The shard's k range is fixed by the rank's heads / gate-up columns, so its rows point into all
 * groups of the layer: qweight = the shard's K / 32 * bits packed rows, scales [n_groups][N] / qzeros [n_groups][N / 32 * bits] = the WHOLE
 * layer's tables, g_idx [K] = the group of each of the shard's rows (required).  y32[M][ldy] = x[M][K] . dequant(W shard), sums left in fp32
 * (weights dequantised as matmul_248_kernel does, quant_linear.py:128; fp32 accumulate).  Replaces the fp16-output generic launch that cost one
 * extra rounding per rank. */
int gptq_matmul248_partial_f32(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *g_idx,
                               float *y32, int64_t ldy, int M, int K, int N, int bits, int n_groups, gptq_stream_t stream);

/* Same contract, forcing one kernel family (tests and benchmarks). */
int gptq_gemv_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales,
                  const int32_t *qzeros, const int32_t *g_idx, const void *bias, void *y,
                  int64_t ldy, int M, int K, int N, int bits, int groupsize, void *workspace,
                  size_t workspace_bytes, gptq_stream_t stream);
int gptq_skinny_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales,
                    const int32_t *qzeros, const int32_t *g_idx, const void *bias, void *y,
                    int64_t ldy, int M, int K, int N, int bits, int groupsize, void *workspace,
                    size_t workspace_bytes, gptq_stream_t stream);
int gptq_gemm_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales,
                  const int32_t *qzeros, const int32_t *g_idx, const void *bias, void *y,
                  int64_t ldy, int M, int K, int N, int bits, int groupsize, gptq_stream_t stream);

/*
 * c = silu(x . deq(B_gate)) * (x . deq(B_up))  -- reference QuantLlamaMLP.triton_llama_mlp +
 * fusedmatmul_248_kernel (quant/fused_mlp.py:206-218, :84-168).  Both weight sets share
 * K, N, bits, groupsize; each has its own scales / qzeros / g_idx.
 */
int gptq_fused_mlp_f16(const void *x, int64_t ldx, const int32_t *qweight_gate,
                       const void *scales_gate, const int32_t *qzeros_gate,
                       const int32_t *g_idx_gate, const int32_t *qweight_up, const void *scales_up,
                       const int32_t *qzeros_up, const int32_t *g_idx_up, void *c, int64_t ldc,
                       int M, int K, int N, int bits, int groupsize, void *workspace,
                       size_t workspace_bytes, gptq_stream_t stream);

/*
 * dx[M,K] = dy[M,N] . deq(B)^T -- reference transpose_matmul248() +
 * transpose_matmul_248_kernel (quant/quant_linear.py:272-279, :191-258); the backward of
 * QuantLinearFunction (:294-301).
 */
int gptq_transpose_matmul248_f16(const void *dy, int64_t lddy, const int32_t *qweight,
                                 const void *scales, const int32_t *qzeros, const int32_t *g_idx,
                                 void *dx, int64_t lddx, int M, int K, int N, int bits,
                                 int groupsize, gptq_stream_t stream);

/*
 * y = x * rsqrt(mean(x^2) + eps) * w, fp32 math, one rounding to fp16 -- reference
 * TritonLlamaRMSNorm.forward + rms_norm_fwd_fused (quant/triton_norm.py:50-67, :7-39).
 * Rows wider than 64 KiB are rejected like the reference (:59-60).
 */
int gptq_rmsnorm_f16(const void *x, int64_t ldx, const void *weight, void *y, int64_t ldy, int M,
                     int N, float eps, gptq_stream_t stream);

/* y[N] = W[N][K] . x (+ bias) for ONE row of x and a dense fp16 weight stored [out, in] (row stride ldw): the LM head of a decode
 * step -- the ordinary nn.Linear the reference's model ends in (llama_inference.py:119-127 -> HF generate), 262 MB per token for LLaMA-7B.
 * HBM-bound, hand-written (csrc/dense_gemv.hip): rows streamed non-temporally, x staged once per workgroup, fp32 accumulation.
 * norm_weight != NULL: x is RMS-normalised first (rms_norm_fwd_fused, quant/triton_norm.py:22-39, rounded to fp16 like the stand-alone
 * launch) -- the model's final norm folded into the same launch.  K % 8 == 0, K <= 65536. */
int gptq_dense_matvec_f16(const void *x, const void *weight, int64_t ldw, const void *bias, void *y, int N, int K, const void *norm_weight,
                          float norm_eps, gptq_stream_t stream);
/*
 * In-place rotate-half RoPE on the q and k slices of a fused qkv activation -- reference
 * triton_rotate_half_ + rotate_half_kernel (quant/fused_attn.py:61-93, :8-58).
 * qk points at element [0,0,0,0,0] of a [bsz, seq, 2, heads, head_dim] fp16 view whose
 * (bsz*seq) rows are row_stride elements apart; position_ids int64 [bsz, seq] with batch
 * stride pos_batch_stride.  cos/sin are computed on the fly in fp32 with theta = base.
 */
int gptq_rope_f16(void *qk, int64_t row_stride, const int64_t *position_ids,
                  int64_t pos_batch_stride, int bsz, int seq, int heads, int head_dim, float base,
                  gptq_stream_t stream);

/*
 * GPU packer -- reference QuantLinear.pack (quant/quant_linear.py:325-371), which runs on the
 * CPU upstream ("TODO: perform packing on GPU", llama.py:264).  weight fp32 [N, K] (nn.Linear
 * layout, already grid-valued), scales/zeros fp32 [N, G] as produced by gptq.py:226-228.
 * Bit-exact with the reference, including the unmasked OR and the zeros-1 wrap.
 */
int gptq_pack_f32(const float *weight, const float *scales, const float *zeros,
                  const int32_t *g_idx, int K, int N, int bits, int groupsize, int32_t *qweight,
                  int32_t *qzeros, void *scales_f16, gptq_stream_t stream);

/* 1 iff g_idx[k] == k / groupsize for all k (device-side check, writes one int32 to `out`). */
int gptq_g_idx_is_trivial(const int32_t *g_idx, int K, int groupsize, int32_t *out,
                          gptq_stream_t stream);

/*
 * Batch-1 decode-step helpers (extension over the reference, which does this part with torch ops
 * between its Triton kernels: triton_rotate_half_ + torch.cat of the KV cache + torch SDPA,
 * quant/fused_attn.py:126-155).  They read the current position from DEVICE memory and use a
 * preallocated cache, so a whole decode step can be captured in one hipGraph.
 *   qkv       fp16 [3, heads, head_dim] -- the fused qkv activation of ONE token; q is rotated in place
 *   position  int64 [1] on the device   -- index of the token being decoded (0-based)
 *   k_cache / v_cache  fp16 [t_max, heads*head_dim]; row `position` is written by rope_kv, rows
 *             0..position are read by attn.  head_dim must be 128 for gptq_decode_attn_f16.
 *   out       fp16 [heads*head_dim] = softmax(q.K^T * scale) V, fp32 math
 */
/*
 * W[K, N] fp16 = the dequantised weight exactly as the reference's kernel forms it on the fly:
 * fp16(q - z) * fp16 scale, one rounding (quant/quant_linear.py:114-128).  Any bits / g_idx.
 * Used by the Python layer for every batch above the weight-streaming kernels (prefill): the product is a plain
 * dense GEMM there, and dequantise-per-call (10-20 us for a LLaMA-7B layer) + library GEMM measures 1.12-1.39x
 * the fused tile kernel behind gptq_matmul248_f16.  A C caller with a BLAS at hand can do the same.
 */
int gptq_dequant_f16(const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *g_idx,
                     void *w, int K, int N, int bits, int groupsize, gptq_stream_t stream);
/* the same with a leading dimension (ldw >= N halves): e.g. gate | up of an MLP side by side in ONE [K, 2N] matrix, so that the
 * pair is one library GEMM (quant/fused_mlp.py, prefill route). */
int gptq_dequant_ld_f16(const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *g_idx,
                        void *w, int64_t ldw, int K, int N, int bits, int groupsize, gptq_stream_t stream);
/*
 * c[m][n] = fp16(silu(gate[m][n]) * up[m][n]), fp32 math -- the epilogue of the reference's fused MLP kernel
 * (quant/fused_mlp.py:160-165) as a pass of its own, for gate / up products that come out of a library GEMM.  N % 8 == 0,
 * leading dimensions multiples of 8 halves, 16-byte aligned pointers; c may alias gate.
 */
int gptq_silu_mul_f16(const void *gate, int64_t ldg, const void *up, int64_t ldu, void *c, int64_t ldc, int M, int N,
                      gptq_stream_t stream);

/*
 * Prefill route (M above the weight-streaming kernels): the layer is dequantised ONCE PER CALL into the workspace (reference
 * numerics fp16(q - z) * fp16 scale, any width, any g_idx -- act-order needs no gather of x here) and the dense product runs
 * through the hand-written tile GEMM of csrc/gemm8.hip (fp16 operands by LDS-DMA, v_mfma_f32_16x16x32_f16, fp32 accumulation,
 * one rounding, bias in the epilogue): same interface and results as gptq_matmul248_f16 / gptq_fused_mlp_f16 (reference
 * matmul248, quant_linear.py:263-269; fused MLP, fused_mlp.py:84-168).  The fused variant stacks gate and up as one [2N, K]
 * operand; ONE launch forms both products per tile and applies SiLU to the FP32 accumulators like the reference's kernel
 * (fused_mlp.py:160-165): no intermediate, no extra rounding.  gptq_set_prefill_route(0) -- and shapes the tile GEMM does not
 * serve (K % 128 != 0) -- take hipBLASLt instead (dlopen'ed at first use, GPTQ_E_LIBRARY when absent, bounded LRU of plans; the
 * gate | up product then leaves the library in FP32 in chunks of <= 8192 rows and SiLU * mul is a pass of its own).
 * workspace: gptq_prefill_workspace_bytes(M, K, N, nsets) bytes (nsets = 1 matmul, 2 fused MLP), 256-byte aligned;
 * GPTQ_E_WORKSPACE when smaller.  K % 32 == 0, N % 32 == 0 like everywhere; any M >= 0.
 */
size_t gptq_prefill_workspace_bytes(int M, int K, int N, int nsets);
int gptq_prefill_matmul_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros,
                            const int32_t *g_idx, const void *bias, void *y, int64_t ldy, int M, int K, int N, int bits,
                            int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream);
/* the backward product on the same route: dx[M, K] = dy[M, N] . deq(W)^T (reference transpose_matmul248,
 * quant_linear.py:272-279, kernel :191-258); interface of gptq_transpose_matmul248_f16 + the workspace (nsets = 1). */
int gptq_prefill_transpose_matmul248_f16(const void *dy, int64_t lddy, const int32_t *qweight, const void *scales,
                                         const int32_t *qzeros, const int32_t *g_idx, void *dx, int64_t lddx, int M, int K, int N,
                                         int bits, int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream);
int gptq_prefill_fused_mlp_f16(const void *x, int64_t ldx, const int32_t *qweight_gate, const void *scales_gate,
                               const int32_t *qzeros_gate, const int32_t *g_idx_gate, const int32_t *qweight_up,
                               const void *scales_up, const int32_t *qzeros_up, const int32_t *g_idx_up, void *c, int64_t ldc,
                               int M, int K, int N, int bits, int groupsize, void *workspace, size_t workspace_bytes,
                               gptq_stream_t stream);

/*
 * Act-order fast path (extension; the reference re-gathers g_idx, scales and zeros for every k row
 * in the kernel, quant/quant_linear.py:114-118).  With perm = stable argsort(g_idx), the re-sorted
 * qweight (gptq_act_order_repack, one-off at load; same shape as qweight) is a trivial-g_idx layer
 * applied to x[perm]: gptq_matmul248_sorted_f16 gathers x through perm on the scalar path and runs
 * the rowwave GEMV (one launch per row of x; M == 1 is the intended use).  scales / qzeros are the
 * checkpoint's own.  Valid when every group has exactly `groupsize` members (what gptq.py:210-216
 * produces) and groupsize % (32/bits) == 0; bits == 4 in this release (else GPTQ_E_VARIANT).
 */
int gptq_act_order_repack(const int32_t *qweight, const int32_t *perm, int K, int N, int bits,
                          int32_t *qweight_sorted, gptq_stream_t stream);
int gptq_matmul248_sorted_f16(const void *x, int64_t ldx, const int32_t *perm, const int32_t *qweight_sorted,
                              const void *scales, const int32_t *qzeros, const void *bias, void *y,
                              int64_t ldy, int M, int K, int N, int bits, int groupsize, void *workspace,
                              size_t workspace_bytes, gptq_stream_t stream);
/* The fused gate/up + SiLU of an act-order MLP: gate_proj and up_proj see the same input, hence the same Hessian
 * diagonal and the same act-order permutation (as q/k/v do, fused_attn.py:177-188) -- one perm, two re-sorted weight
 * sets, c = silu(x[perm].Wg') * (x[perm].Wu').  Same validity rules as gptq_matmul248_sorted_f16. */
int gptq_fused_mlp_sorted_f16(const void *x, int64_t ldx, const int32_t *perm, const int32_t *qweight_gate_sorted,
                              const void *scales_gate, const int32_t *qzeros_gate, const int32_t *qweight_up_sorted,
                              const void *scales_up, const int32_t *qzeros_up, void *c, int64_t ldc, int M, int K, int N,
                              int bits, int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream);
/* [RMSNorm -> act-order QuantLinear] (qweight_up_sorted == NULL) or [RMSNorm -> act-order gate/up + SiLU] of one decode
 * token in ONE launch: the kernel gathers x and the norm weight through perm and normalises what it gathered
 * (sum(x^2) does not depend on the order).  M == 1, 4-bit, groups of >= 64. */
int gptq_rmsnorm_sorted_f16(const void *x, const void *norm_weight, float eps, const int32_t *perm, const int32_t *qweight_sorted,
                            const void *scales, const int32_t *qzeros, const int32_t *qweight_up_sorted, const void *scales_up,
                            const int32_t *qzeros_up, const void *bias, void *y, int K, int N, int bits, int groupsize,
                            void *workspace, size_t workspace_bytes, gptq_stream_t stream);

/*
 * [RMSNorm -> QuantLinear] and [RMSNorm -> fused gate/up] of a decoder layer as ONE launch, M == 1
 * (TritonLlamaRMSNorm.forward followed by QuantLinear.forward / triton_llama_mlp; reference
 * quant/triton_norm.py:50-67 + quant/quant_linear.py:373-377 / quant/fused_mlp.py:206-218).
 * x fp16 [K] is normalised on the fly with the arithmetic of rms_norm_fwd_fused (fp32, one fp16
 * rounding) and never written back.  Returns GPTQ_E_VARIANT when the shape needs the generic
 * kernels (act-order, 3-bit, bits != 4, odd groups): call gptq_rmsnorm_f16 + gptq_matmul248_f16.
 */
int gptq_rmsnorm_matmul248_f16(const void *x, const void *norm_weight, float eps, const int32_t *qweight,
                               const void *scales, const int32_t *qzeros, const int32_t *g_idx,
                               const void *bias, void *y, int K, int N, int bits, int groupsize,
                               void *workspace, size_t workspace_bytes, gptq_stream_t stream);
int gptq_rmsnorm_fused_mlp_f16(const void *x, const void *norm_weight, float eps, const int32_t *qweight_gate,
                               const void *scales_gate, const int32_t *qzeros_gate,
                               const int32_t *g_idx_gate, const int32_t *qweight_up, const void *scales_up,
                               const int32_t *qzeros_up, const int32_t *g_idx_up, void *c, int K, int N,
                               int bits, int groupsize, void *workspace, size_t workspace_bytes,
                               gptq_stream_t stream);

int gptq_decode_rope_kv_f16(void *qkv, const int64_t *position, void *k_cache, void *v_cache, int heads,
                            int head_dim, int t_max, float base, gptq_stream_t stream);
size_t gptq_decode_attn_workspace_bytes(int heads, int head_dim, int t_max);
int gptq_decode_attn_f16(const void *q, const void *k_cache, const void *v_cache, const int64_t *position,
                         void *out, void *workspace, size_t workspace_bytes, int heads, int head_dim,
                         int t_max, float scale, gptq_stream_t stream);

/* gptq_decode_rope_kv_f16 + gptq_decode_attn_f16 as ONE launch (q is rotated internally, the qkv
 * buffer is left untouched); the workspace (same size query; 16-byte aligned) must be zero on first use -- its
 * trailing [heads] uint32 arrival tickets are restored to zero by the kernel.  Round 6: a streaming kernel -- one
 * workgroup per head walks the whole history up to ~768 tokens (no merge at all), at most gptq_decode_attn_splits()
 * splits with an arrival ticket beyond (see "round 6" below). */
int gptq_decode_attn_fused_f16(const void *qkv, const int64_t *position, void *k_cache, void *v_cache, void *out,
                               void *workspace, size_t workspace_bytes, int heads, int head_dim, int t_max,
                               float base, float scale, gptq_stream_t stream);
/* The same launch with {cos, sin} read from a table gptq_rope_table_f32 filled once ([t_max][head_dim / 2][2] fp32, the identical
 * instruction sequence: bit-identical results): the accurate-libm range reduction leaves the per-token critical path. */
int gptq_rope_table_f32(float *table, int t_max, int head_dim, float base, gptq_stream_t stream);
int gptq_decode_attn_fused_table_f16(const void *qkv, const int64_t *position, void *k_cache, void *v_cache, void *out, void *workspace,
                                     size_t workspace_bytes, int heads, int head_dim, int t_max, float base, float scale,
                                     const float *rope_table, gptq_stream_t stream);

/* ---- stripe16: the batch-1 decode matvec WITHOUT a K split, on a load-time repacked copy (csrc/stripe.hip) --------
 * Replaces, for M == 1, the launch of matmul_248_kernel (quant/quant_linear.py:263-269) / fusedmatmul_248_kernel
 * (quant/fused_mlp.py:206-218) on a layout the library owns: the checkpoint buffers are repacked ONCE at load time
 * (the reference has no counterpart; its load path ends at load_state_dict, llama_inference.py:57-60) into
 *   R   uint32 [N/16][K/128][nsets][64][4]  (4-bit) every workgroup's 16 columns contiguous, 1 KiB per wave load, fields
 *                                            re-ordered for a one-shift unpack (see csrc/stripe.hip)
 *   tab half2  [N/16][nsets][G][16]          {scale, zero + 1} per (group, column)
 * stored back to back in ONE buffer of gptq_stripe_bytes() bytes (0 = shape not eligible: bits in {2, 3, 4, 8}; K a multiple of
 * the row block = 256 / 128 / 128 / 64 k and at most 24576 / 24576 / 24576 / 22528; groupsize a power-of-two multiple of a quarter
 * of the row block that divides K, or >= K).  For 8 and 2 bits the same geometry holds with 32 / bits k per word, for 3 bits a lane
 * holds its 32 k in three words ([N/16][K/128][nsets][64][3]); see csrc/stripe.hip.  The
 * checkpoint buffers are not modified and stay the owner of the state_dict.  nsets == 2 packs gate and up together and the
* matvec returns silu(x Wg) * (x Wu).  1 <= M <= 4 rows of x (row strides ldx / ldy) cost the same weight stream as one: the
 * MFMA computes four rows anyway; 5 <= M <= 8 / 16 run two / four MFMA row groups on the same unpacked words while M rows of x
 * fit in LDS (K <= ~9200 / ~4600, else GPTQ_E_VARIANT: the caller takes the weight-streaming MFMA kernel).  M == 1 only: norm_weight != NULL fuses the RMSNorm of x (rms_norm_fwd_fused,
 * quant/triton_norm.py:22-39) in front; perm != NULL (uint16 [K]: the same permutation gptq_act_order_repack takes, narrowed -- every
 * workgroup reads all of it, so its width is load traffic) reads x through a permutation (an act-order layer whose qweight rows were
 * sorted by group with gptq_act_order_repack BEFORE gptq_stripe_repack).  No workspace, no atomics: results are bit-identical
 * run to run. */
size_t gptq_stripe_bytes(int K, int N, int bits, int groupsize, int nsets);
int gptq_stripe_repack(const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *qweight_up, const void *scales_up,
                       const int32_t *qzeros_up, void *stripes, size_t stripes_bytes, int K, int N, int bits, int groupsize,
                       gptq_stream_t stream);
int gptq_stripe_matvec_f16(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, const void *bias, void *y, int64_t ldy, int M,
                           int K, int N, int bits, int groupsize, int nsets, const void *norm_weight, float norm_eps, const uint16_t *perm,
                           gptq_stream_t stream);
/* The same matvec with the fp32 sums stored unrounded: the per-rank PARTIAL of a row-(K-)sharded layer (BASELINE config 5,
 * quant/tensor_parallel.py) -- the shards are summed by ONE all-reduce and rounded to fp16 once, like the unsharded layer.
 * y_partial is fp32 [nsets][N]: with nsets == 2 the gate and the up sums are stored separately (no SiLU: it needs the
 * complete sums). */
int gptq_stripe_matvec_partial_f32(const void *x, const void *stripes, size_t stripes_bytes, float *y_partial, int K, int N, int bits,
                                   int groupsize, int nsets, const uint16_t *perm, gptq_stream_t stream);
/* The same for 1 <= M <= 4 rows of x (row stride ldx): y_partial is fp32 [M][nsets][N] -- the row groups of the decode kernel cost the
 * weight stream of one row, so a K-shard of a small decode batch keeps the one-rounding-after-the-reduce order of M = 1 (trivial g_idx
 * only: gather x[:, perm] first for a group-sorted act-order image). */
int gptq_stripe_matmul_partial_f32(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, float *y_partial, int M, int K, int N,
                                   int bits, int groupsize, int nsets, gptq_stream_t stream);
/* Batches on the same image (csrc/stripe_mm.inc).  1 <= M <= 128 (to 256 in passes of 128): 16-row MFMA tiles (v_mfma_f32_16x16x32_f16) on exactly
 * dequantised q - z, fp32 group scales; either one launch (a stripe x whole K per workgroup, x streamed through LDS) or K slices
 * (128 columns x one slice per workgroup) that meet through fp32 partial tiles in `workspace` and a reduce kernel (summed in
 * slice order: bit-reproducible; no atomics).  Every layer with a stripe image (bits 2 / 3 / 4 / 8): groups of at least a row block
 * keep q - z exact and scale the fp32 accumulator; smaller groups multiply by the lane's fp16 scale before the MFMA (one rounding, the
 * reference's own dequantisation, quant_linear.py:128); GPTQ_E_VARIANT when there is no image (callers fall back to gptq_matmul248_f16).  The workspace
 * (gptq_query(GPTQ_Q_STRIPE_MM_WORKSPACE_BYTES), 256-byte aligned) is pure scratch for the partial tiles: no initialisation, no state
 * between launches; do not share it between launches that may overlap, nor with the zero-invariant split-K workspace of the
 * rowwave kernels.  129 <= M <= gptq_set_stripe_gemm_max_rows() (default 2048; groups of at least a row block, bits 3 / 4 / 8): ONE launch of
 * the 2-D tiled fused-dequantise GEMM (stripe_gemm_kernel: 128 x 128 tiles, weights stay packed, x through LDS) -- no per-call
 * dequantise pass; round 5: while the tiles cover at most half the chip (N = 4096: up to 512 rows) K is sliced over the row tiles too
 * (partial tiles in `workspace` + the reduce kernel, as below 129 rows).  Round 6: 17 <= M <= 128 of a 4-bit layer whose stripes need two or three
 * rounds of workgroups (N = 8192 .. 12288; K <= 8192) and of the gate | up pair of such a shape run the loader / consumer kernel (stripe_mmr_kernel:
 * two waves per workgroup stream x into an LDS ring by LDS-DMA, six only read A fragments out of it and run the MFMAs on weights unpacked in their
 * registers; the pair's consumers split by set) -- one launch, no workspace; a long K on one round of stripes (N <= 4096, K > 6144) runs its K-sliced
 * form: fp32 rows per slice in `workspace` + a combine launch (slice order, one rounding).  Same arithmetic and result conventions as the tiles above.
 * Reference semantics: quant/quant_linear.py:103-137, :415-419; nsets = 2: quant/fused_mlp.py:128-168 (no bias). */
int gptq_stripe_matmul_f16(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, const void *bias, void *y, int64_t ldy, int M,
                           int K, int N, int bits, int groupsize, int nsets, void *workspace, size_t workspace_bytes, gptq_stream_t stream);

/* ---- one-shot all-reduce for the row-sharded layout (BASELINE config 5; csrc/p2p.hip) ------------------------------
 * New functionality (the reference has no collective: llama.py:328-382 is layer placement).  The fp32 partials of a
 * K-sharded QuantLinear are 32-176 KB at batch 1 -- latency-bound -- and MI355X's xGMI is a full mesh, so every rank
 * WRITES its partial into all peers (memory mapped through HIP IPC), flags, and sums the P slots locally in rank order:
 * one hop instead of the 2 (P - 1) of a ring, bit-identical results on every rank, fp16 rounding (+ bias) fused.
 *   gptq_p2p_create   allocates this rank's exchange buffer (gptq_p2p_buffer_bytes: two slot sets [world][n_max] fp32 +
 *                     flags + per-workgroup epoch counters, zeroed) and exports its 64-byte IPC handle; the caller ships
 *                     the handle to the other ranks (any channel, e.g. torch.distributed.all_gather_object)
 *   gptq_p2p_open     maps a peer's buffer into this process (hipIpcOpenMemHandle; needs HSA_ENABLE_IPC_MODE_LEGACY=0)
 *   gptq_p2p_allreduce_f32  peer_buffers = HOST array of `world` pointers, [rank] = own buffer; n % 4 == 0, n <= n_max;
 *                     writes y_f16 = fp16(sum) (+ bias) or, when y_f32 != NULL, the fp32 sum.  All ranks must issue the same
 *                     sequence of calls.  hipGraph-capturable (the epoch lives in device memory).
 *   gptq_p2p_status   0, or 1 + the rank a workgroup gave up waiting for (every spin is bounded). */
size_t gptq_p2p_buffer_bytes(int world, int n_max);
int gptq_p2p_create(int world, int n_max, void **buffer, void *ipc_handle_64);
int gptq_p2p_open(const void *ipc_handle_64, void **buffer);
int gptq_p2p_close(void *buffer, int opened);
int gptq_p2p_status(void *own_buffer, int world, int n_max, gptq_stream_t stream);
int gptq_p2p_allreduce_f32(const float *partial, void *const *peer_buffers, int rank, int world, int n, int n_max, void *y_f16, float *y_f32,
                           const void *bias, gptq_stream_t stream);
/* partial = [2][n_half] fp32, the gate | up partials of a K-sharded fused MLP (gptq_stripe_matvec_partial_f32 with nsets = 2):
 * y_f16[n_half] = fp16(silu(sum gate) * sum up) -- the epilogue of fusedmatmul_248_kernel (quant/fused_mlp.py:160-166), after the reduce. */
int gptq_p2p_allreduce_silu_mul_f32(const float *partial, void *const *peer_buffers, int rank, int world, int n_half, int n_max, void *y_f16,
                                    gptq_stream_t stream);

/* ---- Prepared layers: the ONE call site of the product ------------------------------------------------------------------
 * The reference reaches its kernels through a single call, matmul248(input, qweight, scales, qzeros, g_idx, bits, maxq)
 * (quant/quant_linear.py:263-269) behind QuantLinear.forward (:373-377) -- and fusedmatmul_248 behind QuantLlamaMLP
 * (quant/fused_mlp.py:203-218) -- with an autotuner that picks a kernel per M (custom_autotune.py).  Here a layer is PREPARED once
 * (load time): g_idx is inspected on the host (0 trivial / 1 regular act-order / 2 irregular), the stripe16 image -- of the
 * group-sorted rows plus the permutation for a regular act-order layer -- is built into the caller's image buffer, and
 * gptq_layer_forward() contains the whole M -> kernel table: decode matvec (M = 1, the headline path), row groups (M <= 8),
 * 16-row MFMA tiles (M <= 128), the prefill tile GEMM / library route above, the checkpoint-layout kernels for whatever has no
 * image.  A non-Python consumer binds exactly these entries and gets the product's speed.
 *
 * Memory stays the caller's: `image` (gptq_layer_image_bytes, 256-byte aligned; NULL = no derived copies, checkpoint-layout
 * kernels only), `workspace` (gptq_layer_workspace_bytes(); its first gptq_query(GPTQ_Q_WORKSPACE_BYTES) bytes must be ZERO on
 * first use and are left zero by every call -- one per stream that may run concurrently), `scratch` (gptq_layer_scratch_bytes(layer,
 * M): transient, may be NULL -- forward then takes a slower route that needs none).  The checkpoint buffers are borrowed and must
 * outlive the handle; the handle is a small host object.  gptq_layer_prepare / _inspect synchronise the stream (load time);
 * gptq_layer_forward only enqueues and is hipGraph-capturable.  nsets = 2 (qweight_up != NULL): y = silu(x Wg) * (x Wu).
 */
typedef struct gptq_layer gptq_layer_t;
int gptq_layer_inspect(const int32_t *g_idx, int K, int groupsize, gptq_stream_t stream);   /* 0 / 1 / 2, or GPTQ_E_* */
size_t gptq_layer_image_bytes(int K, int N, int bits, int groupsize, int nsets, int kind);
int gptq_layer_prepare(gptq_layer_t **layer, const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *g_idx,
                       const void *bias, const int32_t *qweight_up, const void *scales_up, const int32_t *qzeros_up,
                       const int32_t *g_idx_up, int K, int N, int bits, int groupsize, void *image, size_t image_bytes,
                       gptq_stream_t stream);
void gptq_layer_destroy(gptq_layer_t *layer);
int gptq_layer_kind(const gptq_layer_t *layer);
int gptq_layer_stripe_image(const gptq_layer_t *layer, const void **stripe, size_t *stripe_bytes, const uint16_t **perm16);
/* Memory mode (reference README.md:23-29 quotes 4891 MiB for 7B 4-bit g128: ONE copy of the packed weights): after this call the
 * caller may free qweight / scales / qzeros -- the stripe16 image is a bijection of them.  Trivial or (round 4) regular act-order g_idx --
 * its image holds the group-sorted rows, the permutation and its inverse; g_idx itself (K ints) stays borrowed --, any width, and an
 * image (GPTQ_E_VARIANT otherwise, nothing changes).  Routes that read the checkpoint layout (prefill, fall-backs) then rebuild it
 * from the image into `scratch` per call; gptq_layer_unpack_checkpoint reproduces one weight set bit-exactly (state_dict()). */
int gptq_layer_release_checkpoint(gptq_layer_t *layer);
int gptq_layer_unpack_checkpoint(const gptq_layer_t *layer, int set, int32_t *qweight, void *scales, int32_t *qzeros, gptq_stream_t stream);
size_t gptq_layer_workspace_bytes(void);
size_t gptq_layer_scratch_bytes(const gptq_layer_t *layer, int M);
/* every fall-back's needs at once: the scratch to retry with when forward() answers GPTQ_E_WORKSPACE (a kernel of the fast route declined at launch) */
size_t gptq_layer_fallback_scratch_bytes(const gptq_layer_t *layer, int M);
int gptq_layer_forward(const gptq_layer_t *layer, const void *x, int64_t ldx, void *y, int64_t ldy, int M, void *workspace,
                       size_t workspace_bytes, void *scratch, size_t scratch_bytes, gptq_stream_t stream);
/* The M -> kernel table of gptq_layer_forward as a host-only query (no launch, no GPU needed): the kernel family a batch of M rows takes,
 * assuming the caller passes gptq_layer_scratch_bytes() of scratch.  It replaces what the reference's Autotuner decides at run time
 * (quant/custom_autotune.py:76-102) by something a caller can read.  A kernel may still decline at launch (LDS limits) and hand the
 * batch to the next rung of the ladder.  kind: the value of gptq_layer_inspect(); has_image: an image was given to gptq_layer_prepare. */
enum {
    GPTQ_ROUTE_STRIPE_DECODE = 1,        /* stripe16 decode kernel (M = 1) / its row groups (2 .. 8 rows); wide layers: C stripes per workgroup (round 5) */
    GPTQ_ROUTE_STRIPE_TILES = 2,         /* 16-row MFMA tiles on the image, up to 128 rows (csrc/stripe_mm.inc) */
    GPTQ_ROUTE_STRIPE_GEMM = 3,          /* fused-dequantise tile GEMM on the image, 129 .. gptq_set_stripe_gemm_max_rows() rows (see there) */
    GPTQ_ROUTE_DENSE_TILE_GEMM = 4,      /* dequantise per call + the tile GEMM of csrc/gemm8.hip */
    GPTQ_ROUTE_DENSE_LIBRARY = 5,        /* dequantise per call + hipBLASLt */
    GPTQ_ROUTE_CHECKPOINT_KERNELS = 6    /* rowwave / stream / generic kernels on the checkpoint layout */
};
int gptq_layer_route_for_shape(int M, int K, int N, int bits, int groupsize, int nsets, int kind, int has_image);
int gptq_layer_route_for(const gptq_layer_t *layer, int M);

/* ---- Batched decode (round 5) --------------------------------------------------------------------------------------------
 * The reference serves every batch with the one kernel (matmul248 masks M only, quant/quant_linear.py:263-269, :373-377) and HF
 * `generate` drives it with [B, 1] steps, left-padded prompts and per-row position_ids (llama_inference.py:119-127).  The entries
 * below are what a graph-captured decode step of B <= 16 sequences is made of (quant/decode.py DecodeEngine(batch = B)):
 *
 * gptq_layer_decode_f16      y[M][N] = residual[M][..] + layer(rmsnorm(x[M][K]))   (norm_weight / residual may be NULL; 1 <= M <= 128)
 *     one decoder block = four of these (input norm -> qkv, o_proj + residual, post-attention norm -> gate/up with SiLU, down_proj
 *     + residual: quant/fused_attn.py:117-161, quant/fused_mlp.py:203-218 between HF's norms and adds).  The decode kernel takes norm
 *     and residual into its own launch up to 4 rows (8 on shapes one round of workgroups covers); the 16-row MFMA tiles take the
 *     residual into their epilogue, the norm then is one launch into `scratch` (gptq_layer_decode_scratch_bytes).  The residual is
 *     added to the ROUNDED product (fp16(fp16(acc) + r)), as the module chain does it (one fp16 tensor add).  y must not alias x.
 * gptq_decode_attn_batch_f16  the fused RoPE + KV append + single-query attention launch (gptq_decode_attn_fused_table_f16) for B rows:
 *     positions[b] (negative = idle row), qkv row b at qkv + b ldq, out row b at out + b ldo, cache slice b at k_cache + b t_max heads 128;
 *     out_perm != NULL: element k of an output row is stored at out_perm[k] (see "Producer-side permutation" below).
 * gptq_dense_matmat_f16       y[M][N] = rmsnorm(x)[M][K] . W[N][K]^T, M <= 16, ONE pass over the dense fp16 weight (the LM head).
 * gptq_add_rows_f16           y = fp16(y + r), row by row.
 */
size_t gptq_layer_decode_scratch_bytes(const gptq_layer_t *layer, int M);
int gptq_layer_decode_f16(const gptq_layer_t *layer, const void *x, int64_t ldx, void *y, int64_t ldy, int M, const void *norm_weight,
                          float norm_eps, const void *residual, int64_t ldr, void *workspace, size_t workspace_bytes, void *scratch,
                          size_t scratch_bytes, gptq_stream_t stream);
/* Round 6: gptq_layer_decode_f16 that leaves the NEXT RMSNorm's rows behind when that is free.  A decode batch of 9 .. 16 rows runs a long-K
 * layer (LLaMA's down_proj) as K slices of the 16-row tiles + a combine launch; that launch owns whole rows of y, so it also writes
 * h[M][ldh] = rmsnorm(y) * next_norm_weight -- bit for bit what gptq_rmsnorm_f16 writes (triton_norm.py:22-39 between down_proj and the next block's
 * qkv_proj: llama modeling's input_layernorm) -- and sets *h_written = 1: the consumer runs without its norm, one launch less per decoder block
 * (at 16 rows the norm fused into the consumer's launch costs more than a launch: every workgroup would normalise all rows itself).  Every
 * other route behaves exactly like gptq_layer_decode_f16, leaves h untouched and sets *h_written = 0.  1 <= M <= 16 for the fusion. */
int gptq_layer_decode_next_norm_f16(const gptq_layer_t *layer, const void *x, int64_t ldx, void *y, int64_t ldy, int M, const void *norm_weight,
                                    float norm_eps, const void *residual, int64_t ldr, const void *next_norm_weight, float next_norm_eps, void *h,
                                    int64_t ldh, int *h_written, void *workspace, size_t workspace_bytes, void *scratch, size_t scratch_bytes,
                                    gptq_stream_t stream);
size_t gptq_decode_attn_batch_workspace_bytes(int batch, int heads, int head_dim, int t_max);
int gptq_decode_attn_batch_f16(const void *qkv, int64_t ldq, const int64_t *positions, void *k_cache, void *v_cache, void *out, int64_t ldo,
                               void *workspace, size_t workspace_bytes, int batch, int heads, int head_dim, int t_max, float base, float scale,
                               const float *rope_table, const int32_t *out_perm, gptq_stream_t stream);
/* Producer-side permutation (round 5): the reference gathers g_idx / scales / zeros per k row in its kernel (quant_linear.py:114-118); here a regular
 * act-order layer runs on the image of its group-sorted rows, which needs x in sorted order.  At M = 1 the decode kernel gathers x itself; where the
 * PRODUCER of x is one of our launches it can write x sorted instead: out_perm above (attention -> o_proj) and gptq_stripe_matvec_perm_out_f16
 * (gate/up + SiLU -> down_proj) store element n at perm[n], perm = gptq_layer_inverse_perm() of the consuming layer, which then runs the trivial
 * kernel (perm = NULL).  NULL = natural order. */
int gptq_layer_inverse_perm(const gptq_layer_t *layer, const int32_t **invperm32);
int gptq_stripe_matvec_perm_out_f16(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, const void *bias, void *y, int64_t ldy, int M,
                                    int K, int N, int bits, int groupsize, int nsets, const void *norm_weight, float norm_eps, const uint16_t *perm,
                                    const int32_t *y_perm, gptq_stream_t stream);
int gptq_dense_matmat_f16(const void *x, int64_t ldx, const void *weight, int64_t ldw, const void *bias, void *y, int64_t ldy, int M, int N,
                          int K, const void *norm_weight, float norm_eps, gptq_stream_t stream);
int gptq_add_rows_f16(void *y, int64_t ldy, const void *r, int64_t ldr, int M, int N, gptq_stream_t stream);

/* ---- round 6: the decode attention as a STREAMING launch whose splits are merged by the NEXT launch ------------------------------------------
 * Replaces F.scaled_dot_product_attention over the grown cache (quant/fused_attn.py:142-155) for one new token per row.  The launch behind every
 * gptq_decode_attn_* entry cuts a row's history into at most gptq_decode_attn_splits() ranges chosen from the row's length at run time; a split is
 * one workgroup that walks its range with an online softmax and ends with a record per head: {M, den} in fp32 and its normalised partial
 * output o[128] in fp16 (one split: o IS the output row).  The entries above merge
 * the records themselves (one split: nothing to merge; several: arrival ticket, last split merges).  The pair below leaves them to the consumer:
 *
 * gptq_decode_attn_split_f16   RoPE + KV append + attention like gptq_decode_attn_batch_f16, but NO output row: the records stay in `workspace`
 *                              ([batch][S][heads * 128] fp16 partial outputs | [batch][S][heads] fp32 {M, den}; S = gptq_decode_attn_splits()).
 * gptq_layer_decode_attn_f16   y = residual + layer(x), where x is the merge of those records, formed while the decode kernel stages x (the
 *                              kernel boundary is the hand-off between the splits' workgroups).  One row, a layer with a trivial g_idx and a
 *                              stripe16 image, K = heads * 128 (o_proj of a batch-1 decode step); GPTQ_E_VARIANT otherwise --
 *                              gptq_layer_decode_attn_supported() answers 1 / 0 ahead of time.  Bit-identical to
 *                              gptq_decode_attn_batch_f16 + gptq_layer_decode_f16 at the same tokens_per_split.
 * tokens_per_split <= 0: the library's default for the mode; both calls of a pair must pass the same value (and the same t_max / batch). */
int gptq_decode_attn_splits(int batch, int heads, int head_dim, int t_max);
int gptq_decode_attn_split_f16(const void *qkv, int64_t ldq, const int64_t *positions, void *k_cache, void *v_cache, void *workspace,
                               size_t workspace_bytes, int batch, int heads, int head_dim, int t_max, float base, float scale, const float *rope_table,
                               int tokens_per_split, gptq_stream_t stream);
int gptq_layer_decode_attn_supported(const gptq_layer_t *layer, int batch, int heads, int head_dim);
int gptq_layer_decode_attn_f16(const gptq_layer_t *layer, const void *attn_workspace, size_t attn_workspace_bytes, const int64_t *positions, int batch,
                               int heads, int head_dim, int t_max, int tokens_per_split, void *y, int64_t ldy, const void *residual, int64_t ldr,
                               gptq_stream_t stream);

/* ---- GPTQ solver (the caller that PRODUCES the weights; reference gptq.py:128-228) -------------------------------
 * One column block [i1, i1 + count), count <= 128, of the sequential quantise / error-feedback loop (gptq.py:177-199)
 * for all rows at once, in the reference's own fp32 arithmetic (IEEE division, round-half-even, no contraction).
 *   W [rows, cols] (ldw)    current weights, read only (the block's columns; as the reference, the in-block updates
 *                           live in registers -- its W1 clone -- and W itself changes only through the trailing update)
 *   Hinv [cols, cols] (ldh) upper Cholesky factor of the inverse damped Hessian (gptq.py:160-163)
 *   scale, zero [rows, ldg] grid of row r, group g = column / groupsize (pass groupsize = cols for "no groups");
 *                           the host fits them from W before the block (gptq.py:181-183 -> quantizer.py:32-76)
 *   Q [rows, cols] (ldq)    out: quantised (de-quantised, fp32) weights of the block's columns
 *   Err [rows, count] (lde) out: Err1, for the trailing update W[:, i2:] -= Err1 . Hinv[i1:i2, i2:] (gptq.py:204)
 *   loss_rows [rows]        in/out: += sum over the block's columns of (w - q)^2 / d^2 / 2 (gptq.py:194,202) */
int gptq_solver_block_f32(const float *W, int64_t ldw, const float *Hinv, int64_t ldh, int rows, int cols, int i1, int count,
                          int groupsize, int maxq, const float *scale, const float *zero, int64_t ldg, float *Q, int64_t ldq,
                          float *Err, int64_t lde, float *loss_rows, gptq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GPTQ_MI355X_H */
