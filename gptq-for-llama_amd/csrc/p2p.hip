// p2p.hip -- one-shot all-reduce of the fp32 partials of a row-(K-)sharded QuantLinear (BASELINE config 5; the reference has
// no collective at all: its multi-GPU mode is layer placement, llama.py:328-382).
//
// The messages are 32-176 KB (fp32 [N] partials at batch 1), i.e. latency-bound: a ring all-reduce pays 2 (P - 1) hops.
// MI355X's xGMI is a full mesh (7 links per GPU), so every rank can WRITE its partial straight into all P peers: one hop.
//   rank r, workgroup w (the same chunk of the vector on every rank):
//     1. copy chunk w of the local partial into slot[parity][r] of every peer p (system-scope, write-through stores into memory
//        mapped with hipIpcOpenMemHandle; p == r is the local copy),
//     2. drain (vmcnt(0) in EVERY storing wave), barrier, one lane per peer stores flag[parity][r][w] = epoch there with a
//        system-scope RELEASE (buffer_wbl2 sc0 sc1 + vmcnt(0) ahead of the store: the data is ordered before the flag in the
//        HIP memory model too, not only by the write-through property of the data stores),
//     3. poll the P local flags [parity][q][w] (one lane each, RELAXED system-scope loads, bounded -- an acquire per poll would
//        invalidate the caches on every iteration), then ONE system-scope ACQUIRE fence after the match, barrier,
//     4. sum the P slots in rank order (identical on every rank: bit-identical results), round the sum to fp16 once; a bias is
//        added to the ROUNDED sum and rounded again -- fp16(fp16(sum) + bias), exactly what the unsharded layer does (the
//        reference adds the bias to the kernel's fp16 output, quant_linear.py:376).
// A workgroup that gives up on a peer (bounded spin) sets the status word AND stores NaN instead of its sums: a stalled or
// dead rank shows in the output, it cannot silently corrupt the replicated hidden state.
// No grid-wide sync: a workgroup only needs the flags of its own chunk.  The epoch lives in DEVICE memory (one word per
// workgroup, bumped by the kernel): a kernel argument would be frozen under hipGraph replay.  Two slot sets alternate by epoch
// parity: a rank can run at most one all-reduce ahead of a peer (it needs that peer's contribution to finish the next one),
// so the set a slow peer is still summing is never overwritten.  Every spin is bounded and reports through a status word.
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

namespace {

constexpr int P2P_MAX_WORLD = 16;
constexpr int P2P_WGS = 64;          // chunks of the vector = workgroups
constexpr int P2P_THREADS = 256;

struct P2PPeers {
    float *slots[P2P_MAX_WORLD];      // peer p's slot area  [2][world][n_max]
    uint32_t *flags[P2P_MAX_WORLD];   // peer p's flag area  [2][world][P2P_WGS]
};

GPTQ_DEV void store_sys_f4(float *p, float4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// y16 = fp16(sum) (+ bias), or y32 = sum (fp32).  PAIR: the vector is [2][n/2] = the gate and up partials of a K-sharded
// fused MLP; a workgroup owns the same columns of both halves and y16[n/2] = fp16(silu(sum gate) * sum up), the epilogue of
// fusedmatmul_248_kernel (reference quant/fused_mlp.py:160-166) applied AFTER the reduce.
template <bool PAIR>
__global__ void __launch_bounds__(P2P_THREADS) p2p_allreduce_kernel(const float *__restrict__ part, const P2PPeers peers, int rank, int world, int n,
                                                                    int n_max, uint32_t *__restrict__ epochs, uint32_t *__restrict__ status,
                                                                    half_t *__restrict__ y16, float *__restrict__ y32,
                                                                    const half_t *__restrict__ bias) {
    __shared__ uint32_t s_epoch, s_fail;
    const int wg = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        const uint32_t e = epochs[wg] + 1u;   // this workgroup's call counter: the same sequence on every rank
        epochs[wg] = e;
        s_epoch = e;
        s_fail = 0u;
    }
    __syncthreads();
    const uint32_t epoch = s_epoch;
    const int par = (int)(epoch & 1u);
    // chunk of this workgroup, in float4 units
    const int nh = PAIR ? n / 2 : n;                 // columns a workgroup chunk is cut from
    const int nv = nh / 4, per = (nv + P2P_WGS - 1) / P2P_WGS;
    const int v0 = wg * per, v1 = min(nv, v0 + per);

    // 1. push my chunk to every peer
    for (int v = v0 + tid; v < v1; v += P2P_THREADS) {
#pragma unroll
        for (int h = 0; h < (PAIR ? 2 : 1); h++) {
            const size_t o = (size_t)h * nh + (size_t)v * 4;
            const float4_t val = *(const float4_t *)(part + o);
            for (int p = 0; p < world; p++) store_sys_f4(peers.slots[p] + ((size_t)par * world + rank) * n_max + o, val);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // 2. publish: one flag per (source rank, chunk) at every peer
    if (tid < world)
        __hip_atomic_store(peers.flags[tid] + ((size_t)par * world + rank) * P2P_WGS + wg, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 3. wait for the same chunk of every rank
    if (tid < world) {
        const uint32_t *f = peers.flags[rank] + ((size_t)par * world + tid) * P2P_WGS + wg;
        uint32_t seen = 0;
        int spin = 0;
        for (; spin < (1 << 24); spin++) {
            seen = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (seen == epoch) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (seen != epoch) {   // gave up: peer tid never arrived
            __hip_atomic_store(status, 1u + (uint32_t)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s_fail = 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope, once, after the match: the slot reads below happen-after the peers' releases
    }
    __syncthreads();
    const bool failed = s_fail != 0u;
    // 4. sum in rank order (system-scope loads: the slots were written by other devices), epilogue
    const float *mine = peers.slots[rank] + (size_t)par * world * n_max;
    for (int v = v0 + tid; v < v1; v += P2P_THREADS) {
        // eight system-scope loads in flight, one wait (gptq_device.h); ranks past `world` re-read slot 0 and are skipped
        float4_t acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < (PAIR ? 2 : 1); h++) {
            float4_t a = {0.f, 0.f, 0.f, 0.f};
            for (int q0 = 0; q0 < world; q0 += 8) {
                const float *ptrs[8];
#pragma unroll
                for (int i = 0; i < 8; i++) ptrs[i] = mine + (size_t)(q0 + i < world ? q0 + i : 0) * n_max + (size_t)h * nh + (size_t)v * 4;
                float4_t t[8];
                load_sys16_x8(t, ptrs);
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (q0 + i < world) a += t[i];
            }
            if (h == 0) acc = a;
            else acc2 = a;
        }
        if (failed) {   // a peer's contribution is missing: make it visible
            const float qnan = __builtin_nanf("");
            acc = float4_t{qnan, qnan, qnan, qnan};
            acc2 = acc;
        }
        if (PAIR) {
            half4_t h;
#pragma unroll
            for (int i = 0; i < 4; i++) h[i] = (half_t)(acc[i] / (1.f + __expf(-acc[i])) * acc2[i]);
            *(half4_t *)(y16 + (size_t)v * 4) = h;
        } else if (y32) {
            *(float4_t *)(y32 + (size_t)v * 4) = acc;
        } else {
            half4_t h = {(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
            if (bias) {
                const half4_t b = *(const half4_t *)(bias + (size_t)v * 4);
#pragma unroll
                for (int i = 0; i < 4; i++) h[i] = (half_t)((float)h[i] + (float)b[i]);
            }
            *(half4_t *)(y16 + (size_t)v * 4) = h;
        }
    }
}

}  // namespace

size_t p2p_buffer_bytes(int world, int n_max) {
    return (size_t)2 * world * n_max * 4 + (size_t)2 * world * P2P_WGS * 4 + (size_t)P2P_WGS * 4 + 256;
}

int p2p_allreduce_launch(const float *part, void *const *peer_bufs, int rank, int world, int n, int n_max, half_t *y16, float *y32, const half_t *bias,
                         bool pair, hipStream_t s) {
    P2PPeers pp{};
    const size_t slot_bytes = (size_t)2 * world * n_max * 4, flag_bytes = (size_t)2 * world * P2P_WGS * 4;
    for (int p = 0; p < world; p++) {
        pp.slots[p] = (float *)peer_bufs[p];
        pp.flags[p] = (uint32_t *)((char *)peer_bufs[p] + slot_bytes);
    }
    uint32_t *epochs = (uint32_t *)((char *)peer_bufs[rank] + slot_bytes + flag_bytes);
    uint32_t *status = epochs + P2P_WGS;
    if (pair) hipLaunchKernelGGL(p2p_allreduce_kernel<true>, dim3(P2P_WGS), dim3(P2P_THREADS), 0, s, part, pp, rank, world, n, n_max, epochs, status, y16, y32, bias);
    else hipLaunchKernelGGL(p2p_allreduce_kernel<false>, dim3(P2P_WGS), dim3(P2P_THREADS), 0, s, part, pp, rank, world, n, n_max, epochs, status, y16, y32, bias);
    return (int)hipGetLastError();
}

}  // namespace gptq

using namespace gptq;

extern "C" {

size_t gptq_p2p_buffer_bytes(int world, int n_max) {
    if (world < 1 || world > P2P_MAX_WORLD || n_max <= 0 || n_max % 4 != 0) return 0;
    return p2p_buffer_bytes(world, n_max);
}

// allocate this rank's exchange buffer (uncached device memory, zeroed) and export its IPC handle (64 bytes)
int gptq_p2p_create(int world, int n_max, void **buffer, void *ipc_handle_64) {
    if (!buffer || !ipc_handle_64) return GPTQ_E_NULL;
    const size_t bytes = gptq_p2p_buffer_bytes(world, n_max);
    if (!bytes) return GPTQ_E_SHAPE;
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return (int)e;
    }
    e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) return (int)e;
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
    __builtin_memcpy(ipc_handle_64, &h, 64);
    *buffer = p;
    return 0;
}

int gptq_p2p_open(const void *ipc_handle_64, void **buffer) {
    if (!ipc_handle_64 || !buffer) return GPTQ_E_NULL;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, ipc_handle_64, 64);
    const hipError_t e = hipIpcOpenMemHandle(buffer, h, hipIpcMemLazyEnablePeerAccess);
    return (int)e;
}

int gptq_p2p_close(void *buffer, int opened) {
    if (!buffer) return GPTQ_E_NULL;
    return (int)(opened ? hipIpcCloseMemHandle(buffer) : hipFree(buffer));
}

// status word of this rank's buffer: 0, or 1 + the rank a workgroup gave up waiting for
int gptq_p2p_status(void *own_buffer, int world, int n_max, gptq_stream_t stream) {
    if (!own_buffer) return GPTQ_E_NULL;
    uint32_t v = 0;
    const size_t off = (size_t)2 * world * n_max * 4 + (size_t)2 * world * 64 * 4 + (size_t)64 * 4;
    hipError_t e = hipMemcpyAsync(&v, (char *)own_buffer + off, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? (int)v : -(int)e - 1000;
}

int gptq_p2p_allreduce_f32(const float *partial, void *const *peer_buffers, int rank, int world, int n, int n_max, void *y_f16, float *y_f32,
                           const void *bias, gptq_stream_t stream) {
    if (!partial || !peer_buffers || (!y_f16 && !y_f32)) return GPTQ_E_NULL;
    if (world < 1 || world > 16 || rank < 0 || rank >= world || n <= 0 || n % 4 != 0 || n > n_max || n_max % 4 != 0) return GPTQ_E_SHAPE;
    for (int p = 0; p < world; p++)
        if (!peer_buffers[p]) return GPTQ_E_NULL;
    if (((uintptr_t)partial % 16) || ((uintptr_t)y_f16 % 8) || ((uintptr_t)y_f32 % 16) || (bias && ((uintptr_t)bias % 8))) return GPTQ_E_ALIGN;
    return p2p_allreduce_launch(partial, peer_buffers, rank, world, n, n_max, (half_t *)y_f16, y_f32, (const half_t *)bias, false, (hipStream_t)stream);
}

// partial = [2][n_half] fp32 (gate | up partials of a K-sharded fused MLP): y_f16[n_half] = fp16(silu(sum gate) * sum up)
int gptq_p2p_allreduce_silu_mul_f32(const float *partial, void *const *peer_buffers, int rank, int world, int n_half, int n_max, void *y_f16,
                                    gptq_stream_t stream) {
    if (!partial || !peer_buffers || !y_f16) return GPTQ_E_NULL;
    if (world < 1 || world > 16 || rank < 0 || rank >= world || n_half <= 0 || n_half % 4 != 0 || 2 * (int64_t)n_half > n_max || n_max % 4 != 0)
        return GPTQ_E_SHAPE;
    for (int p = 0; p < world; p++)
        if (!peer_buffers[p]) return GPTQ_E_NULL;
    if (((uintptr_t)partial % 16) || ((uintptr_t)y_f16 % 8)) return GPTQ_E_ALIGN;
    return p2p_allreduce_launch(partial, peer_buffers, rank, world, 2 * n_half, n_max, (half_t *)y_f16, nullptr, nullptr, true, (hipStream_t)stream);
}

}  // extern "C"
