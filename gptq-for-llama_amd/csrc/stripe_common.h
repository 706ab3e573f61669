// stripe_common.h -- shared by stripe.hip (repack, dispatch) and the per-bit-width kernel translation units.
#pragma once
#include "gptq_internal.h"

namespace gptq {

constexpr int STRIPE_NW = 8;   // waves per workgroup
// row blocks per wave (fully unrolled): 4-bit K <= 24576, 8-bit K <= 22528, 2-bit K <= 24576
__host__ __device__ constexpr int stripe_max_nu(int bits) { return bits == 8 ? 44 : ((bits == 4 || bits == 3) ? 24 : 12); }
// k per lane and row block (the lane's dwordx4; 3-bit: dwordx3 = 32 k) and per row block
__host__ __device__ constexpr int stripe_lk(int bits) { return bits == 3 ? 32 : 4 * (32 / bits); }
// field position p (bits bits*p ..) of a stripe word holds k = stripe_k_of_pos(p, F) of the packed row, F = 32 / bits fields:
// even k in the low half-word, odd k in the high half-word
__host__ __device__ constexpr int stripe_k_of_pos(int p, int F) { return p < F / 2 ? 2 * p : 2 * (p - F / 2) + 1; }

int stripe_gemv_dispatch_b2(const StripeParams &p, hipStream_t s);
int stripe_gemv_dispatch_b3(const StripeParams &p, hipStream_t s);
int stripe_gemv_dispatch_b4(const StripeParams &p, hipStream_t s);
int stripe_gemv_dispatch_b8(const StripeParams &p, hipStream_t s);
// small decode batches through v_mfma_f32_16x16x32_f16 (stripe_mm.inc): 1 <= M <= 64, GPTQ_E_VARIANT when not eligible
int stripe_mm_dispatch_b2(const StripeParams &p, void *ws, size_t ws_bytes, int forced_slices, hipStream_t s);
int stripe_mm_dispatch_b3(const StripeParams &p, void *ws, size_t ws_bytes, int forced_slices, hipStream_t s);
int stripe_mm_dispatch_b4(const StripeParams &p, void *ws, size_t ws_bytes, int forced_slices, hipStream_t s);
int stripe_mm_dispatch_b8(const StripeParams &p, void *ws, size_t ws_bytes, int forced_slices, hipStream_t s);
// batches above 128 rows: 2-D tiled fused-dequantise MFMA GEMM on the stripe16 image (stripe_mm.inc, stripe_gemm_kernel)
int stripe_gemm_dispatch_b3(const StripeParams &p, void *ws, size_t ws_bytes, hipStream_t s);
int stripe_gemm_dispatch_b4(const StripeParams &p, void *ws, size_t ws_bytes, hipStream_t s);
int stripe_gemm_dispatch_b8(const StripeParams &p, void *ws, size_t ws_bytes, hipStream_t s);
int stripe_gemm_dispatch_b2(const StripeParams &p, void *ws, size_t ws_bytes, hipStream_t s);
constexpr size_t STRIPE_MM_WS_BYTES = (size_t)64 << 20;   // counters + partial tiles of the largest supported launch

}  // namespace gptq
