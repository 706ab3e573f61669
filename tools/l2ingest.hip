// tools/l2ingest.hip -- how fast can ONE CU pull bytes that sit in L2?  (the bound DESIGN 3.3a names for 16 < M <= 128: every
// workgroup of the small-batch kernels reads the same x, M K 2 bytes, next to its own weights.)
// 256 workgroups x 512 threads (one per CU), all reading the SAME buffer of `bytes` (L2 / Infinity-Cache resident after the first
// touch), each workgroup starting at a different offset (phase) so that the 32 workgroups of an XCD do not ask for one line at once.
// Variants: plain dwordx4 loads into registers with U loads in flight per wave; LDS-DMA (global_load_lds_dwordx4) into a ring with
// D instructions in flight per wave; 1 or 2 workgroups per CU; buffer private per workgroup (control: no sharing, HBM/MALL stream).
// Build: hipcc -O3 --offload-arch=gfx950 -o l2ingest l2ingest.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// plain loads: every thread reads pieces tid, tid + 512, ... of the buffer (16 B each), U in flight, xor-accumulate
template <int U>
__global__ void __launch_bounds__(512) k_plain(const u32x4 *__restrict__ buf, size_t pieces, size_t wg_stride_pieces, int phase_pieces, uint32_t *out) {
    const u32x4 *b = buf + (size_t)blockIdx.x * wg_stride_pieces;
    const size_t start = ((size_t)blockIdx.x * phase_pieces) % pieces;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i0 = 0; i0 < pieces; i0 += 512 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t p = start + i0 + (size_t)u * 512 + threadIdx.x;
            if (p >= pieces) p -= pieces;
            v[u] = b[p];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = acc[0];
}

// LDS-DMA: each wave keeps D instructions (1 KiB each) in flight into its private ring; consumption = one ds_read per landed slot
template <int D>
__global__ void __launch_bounds__(512) k_dma(const u32x4 *__restrict__ buf, size_t pieces, size_t wg_stride_pieces, int phase_pieces, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const u32x4 *b = buf + (size_t)blockIdx.x * wg_stride_pieces;
    const size_t start = ((size_t)blockIdx.x * phase_pieces) % pieces;
    unsigned char *ring = smem + wave * D * 1024;
    const size_t per_wave = pieces / 8;            // pieces of this wave's share (contiguous 1/8 of the buffer, rotated by the phase)
    const size_t w0 = start + wave * per_wave;
    u32x4 acc = {0, 0, 0, 0};
    const size_t n = per_wave / 64;                // instructions of this wave
    auto issue = [&](size_t i) {
        size_t p = w0 + i * 64 + lane;
        while (p >= pieces) p -= pieces;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b + p),
                                         (__attribute__((address_space(3))) void *)(ring + (i % D) * 1024), 16, 0, 0);
    };
    for (size_t i = 0; i < (size_t)D && i < n; i++) issue(i);
    for (size_t i = 0; i < n; i++) {
        // wait until instruction i has landed: at most D - 1 younger ones may still be in flight
        if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (D == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (D == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (D == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        acc ^= *(const u32x4 *)(ring + (i % D) * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + D < n) issue(i + D);
        else asm volatile("s_nop 0" :::);   // tail: fewer in flight than the wait constant assumes -> the waits above only get stricter
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = acc[0];
}


// A-fragment pattern of v_mfma_f32_16x16x32_f16 straight from global memory: x [rows][K] fp16; lane l reads the 16 bytes of row
// 16 t + l % 16 at k = k0 + 8 (l / 16): a wave instruction touches 16 rows x 64 contiguous bytes.  Wave w walks its K / 8 share of k for
// all row tiles, U instructions in flight.  (What a small-batch kernel would do to skip the LDS staging of x.)
template <int U>
__global__ void __launch_bounds__(512) k_frag(const uint16_t *__restrict__ x, int rows, int K, int phase_k, uint32_t *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles = rows / 16, kw = K / 8, steps = kw / 32;
    const int kstart = (int)(((size_t)blockIdx.x * phase_k) % K);
    u32x4 acc = {0, 0, 0, 0};
    const int n = steps * tiles;                     // instructions of this wave: step-major, tiles inside
    for (int i0 = 0; i0 < n; i0 += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(i0 + u, n - 1), st = i / tiles, t = i % tiles;
            int k = kstart + wave * kw + st * 32 + 8 * (lane >> 4);
            if (k >= K) k -= K;
            v[u] = *(const u32x4 *)(x + (size_t)(16 * t + (lane & 15)) * K + k);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = acc[0];
}


// plain loads -> ds_write_b128 into a wave-private LDS slot -> ds_read_b128 back (what a kernel that stages x through LDS without DMA pays):
// U instructions (1 KiB each) per round and wave, next round's loads in flight while this round's bytes go through LDS
template <int U, int THREADS>
__global__ void __launch_bounds__(THREADS) k_plain_lds(const u32x4 *__restrict__ buf, size_t pieces, int phase_pieces, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWV = THREADS / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 *slot = (u32x4 *)(smem + wave * U * 1024);
    const size_t start = ((size_t)blockIdx.x * phase_pieces) % pieces;
    const size_t per_wave = pieces / NWV, w0 = start + wave * per_wave, n = per_wave / 64;
    auto addr = [&](size_t i) { size_t p = w0 + i * 64 + lane; while (p >= pieces) p -= pieces; return buf + p; };
    u32x4 v[U], acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = *addr(u);
    for (size_t i0 = 0; i0 < n; i0 += U) {
#pragma unroll
        for (int u = 0; u < U; u++) slot[u * 64 + lane] = v[u];
        if (i0 + U < n) {
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = *addr(i0 + U + u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= slot[u * 64 + (lane ^ 1)];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = acc[0];
}

template <typename F>
static float time_us(F launch, hipStream_t s, int reps = 20) {
    launch(); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) launch();
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms * 1e3f / reps);
    }
    return best;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t max_bytes = 1 << 20;
    u32x4 *shared_buf, *priv_buf; uint32_t *out;
    CK(hipMalloc(&shared_buf, max_bytes));
    CK(hipMalloc(&priv_buf, max_bytes * 512));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(shared_buf, 1, max_bytes)); CK(hipMemset(priv_buf, 1, max_bytes * 512));
    CK(hipFuncSetAttribute((const void *)k_dma<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16 * 1024));
    CK(hipFuncSetAttribute((const void *)k_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8 * 1024));
    CK(hipFuncSetAttribute((const void *)k_plain_lds<8, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8 * 1024));
    CK(hipFuncSetAttribute((const void *)k_plain_lds<4, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 4 * 1024));
    const float empty = time_us([&] { hipLaunchKernelGGL(k_plain<1>, dim3(256), dim3(512), 0, s, shared_buf, (size_t)0, (size_t)0, 0, out); }, s);
    printf("empty launch (256 x 512 threads, back to back): %.2f us\n", empty);
    for (int wgs : {256, 512}) {
        for (size_t bytes : {(size_t)128 << 10, (size_t)512 << 10, (size_t)1 << 20}) {
            const size_t pieces = bytes / 16;
            printf("== %d workgroups, every one reads the same %zu KB (L2 hits after the first touch); us per launch / GB/s per CU (launch %.2f us subtracted)\n", wgs,
                   bytes >> 10, empty);
            for (int phase : {0, 1}) {
                const int ph = phase ? (int)(pieces / 32) : 0;   // workgroup b starts b / 32 of the buffer in: the 32 workgroups of an XCD are spread over it
                auto rep = [&](const char *name, float us) {
                    const double per_cu = (double)bytes * wgs / 256.0;
                    printf("   %-34s phase %d: %7.2f us  %6.1f GB/s per CU  (%5.1f TB/s chip)\n", name, phase, us, per_cu / (us - empty) / 1e3, per_cu * 256 / (us - empty) / 1e6);
                };
                rep("plain loads, 4 in flight / wave", time_us([&] { hipLaunchKernelGGL(k_plain<4>, dim3(wgs), dim3(512), 0, s, shared_buf, pieces, (size_t)0, ph, out); }, s));
                rep("plain loads, 8 in flight / wave", time_us([&] { hipLaunchKernelGGL(k_plain<8>, dim3(wgs), dim3(512), 0, s, shared_buf, pieces, (size_t)0, ph, out); }, s));
                rep("plain loads, 16 in flight / wave", time_us([&] { hipLaunchKernelGGL(k_plain<16>, dim3(wgs), dim3(512), 0, s, shared_buf, pieces, (size_t)0, ph, out); }, s));
                rep("LDS-DMA, 4 in flight / wave", time_us([&] { hipLaunchKernelGGL(k_dma<4>, dim3(wgs), dim3(512), 8 * 4 * 1024, s, shared_buf, pieces, (size_t)0, ph, out); }, s));
                rep("LDS-DMA, 8 in flight / wave", time_us([&] { hipLaunchKernelGGL(k_dma<8>, dim3(wgs), dim3(512), 8 * 8 * 1024, s, shared_buf, pieces, (size_t)0, ph, out); }, s));
                rep("LDS-DMA, 16 in flight / wave", time_us([&] { hipLaunchKernelGGL(k_dma<16>, dim3(wgs), dim3(512), 8 * 16 * 1024, s, shared_buf, pieces, (size_t)0, ph, out); }, s));
            }
            for (int phase : {0, 1}) {
                const int ph = phase ? (int)(pieces / 32) : 0;
                const double per_cu = (double)bytes * wgs / 256.0;
                float us = time_us([&] { hipLaunchKernelGGL((k_plain_lds<8, 512>), dim3(wgs), dim3(512), 8 * 8 * 1024, s, shared_buf, pieces, ph, out); }, s);
                printf("   %-34s phase %d: %7.2f us  %6.1f GB/s per CU  (%5.1f TB/s chip)\n", "plain -> ds_write -> ds_read, 8 x 8 waves", phase, us, per_cu / (us - empty) / 1e3, per_cu * 256 / (us - empty) / 1e6);
                us = time_us([&] { hipLaunchKernelGGL((k_plain_lds<4, 512>), dim3(wgs), dim3(512), 8 * 4 * 1024, s, shared_buf, pieces, ph, out); }, s);
                printf("   %-34s phase %d: %7.2f us  %6.1f GB/s per CU  (%5.1f TB/s chip)\n", "plain -> ds_write -> ds_read, 4 x 8 waves", phase, us, per_cu / (us - empty) / 1e3, per_cu * 256 / (us - empty) / 1e6);
                if (wgs == 256) {
                    us = time_us([&] { hipLaunchKernelGGL((k_plain_lds<4, 1024>), dim3(wgs), dim3(1024), 16 * 4 * 1024, s, shared_buf, pieces, ph, out); }, s);
                    printf("   %-34s phase %d: %7.2f us  %6.1f GB/s per CU  (%5.1f TB/s chip)\n", "plain -> ds_write -> ds_read, 4 x 16 waves", phase, us, per_cu / (us - empty) / 1e3, per_cu * 256 / (us - empty) / 1e6);
                }
            }
            {   // the same bytes as an [M][4096] fp16 matrix read in MFMA A-fragment order
                const int K = 4096, rows = (int)(bytes / (K * 2));
                for (int phase : {0, 1}) {
                    const int phk = phase ? K / 32 : 0;
                    const double per_cu = (double)bytes * wgs / 256.0;
                    float us = time_us([&] { hipLaunchKernelGGL(k_frag<8>, dim3(wgs), dim3(512), 0, s, (const uint16_t *)shared_buf, rows, K, phk, out); }, s);
                    printf("   %-34s phase %d: %7.2f us  %6.1f GB/s per CU  (%5.1f TB/s chip)\n", "A-fragment loads (16 rows x 64 B), 8", phase, us, per_cu / (us - empty) / 1e3, per_cu * 256 / (us - empty) / 1e6);
                    us = time_us([&] { hipLaunchKernelGGL(k_frag<16>, dim3(wgs), dim3(512), 0, s, (const uint16_t *)shared_buf, rows, K, phk, out); }, s);
                    printf("   %-34s phase %d: %7.2f us  %6.1f GB/s per CU  (%5.1f TB/s chip)\n", "A-fragment loads (16 rows x 64 B), 16", phase, us, per_cu / (us - empty) / 1e3, per_cu * 256 / (us - empty) / 1e6);
                }
            }
            // control: every workgroup its own buffer (no sharing: the stream comes from HBM / Infinity Cache)
            const float us = time_us([&] { hipLaunchKernelGGL(k_plain<8>, dim3(wgs), dim3(512), 0, s, priv_buf, pieces, pieces, 0, out); }, s);
            printf("   %-34s         : %7.2f us  %6.1f GB/s per CU  (%5.1f TB/s chip)\n", "control: private buffers, plain 8", us, (double)bytes * wgs / 256.0 / (us - empty) / 1e3,
                   (double)bytes * wgs / (us - empty) / 1e6);
        }
    }
    return 0;
}
