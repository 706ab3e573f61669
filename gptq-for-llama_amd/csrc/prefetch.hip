// prefetch.hip -- run-ahead weight prefetcher for the batch-1 decode chain (round 4).
//
// The decode pass is a chain of dependent launches, each of which streams 8.7 .. 47 MB of weights nobody has touched since the last
// token.  Between two launches the HBM pipe is idle: ~1.3 us of dependent-launch boundary + the first-byte latency of the next kernel
// + the reduce / store tail of the previous one (0.51 of the HBM peak over the LLaMA-7B pass; two independent chains
// on one GPU reach 0.65).  Weights, unlike activations, do not depend on the chain -- so a SECOND, independent stream of work can pull
// them towards the CUs while the chain is busy with an earlier op: this kernel.  It reads (and discards) the head of every stripe of
// op j + lead while op j runs, paced by a counter the decode kernels tick when they start (StripeParams::progress), so that the lines
// sit in the 256 MiB Infinity Cache -- and, with XCD affinity, in the L2 of the XCD whose workgroups will ask for them (block b of a
// decode launch runs on XCD b % 8, MI355X_MICROARCH.md "Workgroup dispatch") -- when the consumer's loads arrive.
//
// No hand-off, no flag the chain waits on: the chain never reads anything this kernel writes; a late or absent prefetcher costs
// nothing but the missed hits.  Loads are LDS-DMA (global_load_lds_dwordx4) into one scratch slot per wave: no VGPR destinations, up to
// DEPTH KiB in flight per wave, nothing to consume.  Every spin is bounded: a prefetcher the runtime serialises behind (or in front
// of) the chain gives up and exits.
#include <atomic>
#include "gptq_device.h"
#include "gptq_internal.h"
#include "../../include/gptq_mi355x.h"

namespace gptq {
namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int PF_WAVES = 4;

// one 1-KiB wave load of `base + off` (clamped to the last whole KiB of the region), DEPTH - 1 older ones may stay in flight
template <int DEPTH>
GPTQ_DEV void touch_kib(const char *base, size_t off, size_t region_bytes, unsigned char *slot, int lane) {
    size_t o = off + (size_t)lane * 16;
    const size_t last = region_bytes - 16;
    if (o > last) o = last;
    __builtin_amdgcn_global_load_lds((gptr_t)(base + o), (lptr_t)slot, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
}

// grid = 8 * blocks_per_xcd workgroups of PF_WAVES waves.  Op j is touched once `*progress - base + lead >= j` (progress == NULL: no pacing:
// the whole plan, front to back).  head_kib = KiB per stripe to touch (0: the whole stripe).  affinity != 0: the workgroups that
// (by observation) run on XCD c take the stripes s with s % 8 == c.
template <int DEPTH>
__global__ void __launch_bounds__(PF_WAVES * 64) prefetch_kernel(const gptq_prefetch_op_t *__restrict__ ops, int first, int last, const uint32_t *progress,
                                                                 uint32_t base, int lead, uint32_t head_kib, int affinity, uint32_t spin_limit,
                                                                 uint32_t *status) {
    __shared__ __attribute__((aligned(1024))) unsigned char slots[PF_WAVES * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *slot = slots + wave * 1024;
    const int nx = affinity ? 8 : 1;
    const int xcd = affinity ? (int)(blockIdx.x % 8) : 0;
    const int gw = (affinity ? (int)(blockIdx.x / 8) : (int)blockIdx.x) * PF_WAVES + wave;   // this wave among the waves of its XCD (of the grid)
    const int nw = (affinity ? (int)(gridDim.x / 8) : (int)gridDim.x) * PF_WAVES;
    uint32_t seen = 0;   // ops the chain had started when this wave last looked (monotonic: a stale value only delays this wave)
    for (int j = first; j < last; j++) {
        if (progress != nullptr && (int)seen + lead < j) {
            // every wave polls for itself (one coalesced request; a workgroup barrier here would drain all four waves' loads per op)
            uint32_t spins = 0;
            for (;;) {
                seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base);
                if ((int)seen + lead >= j) break;
                if (++spins > spin_limit) {   // the chain is not moving (or this launch was serialised around it): give up, never block anybody
                    if (lane == 0 && status) atomicAdd(status, 1u);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    return;
                }
                __builtin_amdgcn_s_sleep(32);
            }
        }
        const gptq_prefetch_op_t op = ops[j];
        const char *R = (const char *)op.weights, *T = (const char *)op.table;
        const int ns = (int)op.nstripes / nx;                    // stripes of this XCD (nstripes % 8 == 0 for every eligible shape)
        const uint32_t tk = (op.table_stripe_bytes + 1023) / 1024;    // KiB of table per stripe (first thing a decode workgroup asks for)
        uint32_t hk = op.stripe_bytes / 1024;
        if (head_kib != 0 && head_kib < hk) hk = head_kib;
        const size_t rbytes = (size_t)op.nstripes * op.stripe_bytes, tbytes = (size_t)op.nstripes * op.table_stripe_bytes;
        // block-major: the first KiB of every stripe, then the second ... (the consumer's waves ask for row blocks in this order)
        const int ntab = ns * (int)tk, nblk = ns * (int)hk;
        for (int t = gw; t < ntab; t += nw) {
            const int s = (t % ns) * nx + xcd, b = t / ns;
            touch_kib<DEPTH>(T, (size_t)s * op.table_stripe_bytes + (size_t)b * 1024, tbytes, slot, lane);
        }
        for (int t = gw; t < nblk; t += nw) {
            const int s = (t % ns) * nx + xcd, b = t / ns;
            touch_kib<DEPTH>(R, (size_t)s * op.stripe_bytes + (size_t)b * 1024, rbytes, slot, lane);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

std::atomic<uint32_t *> g_progress{nullptr};

}  // namespace

uint32_t *stripe_progress_counter() { return g_progress.load(std::memory_order_relaxed); }

}  // namespace gptq

using namespace gptq;

extern "C" {

int gptq_set_progress_counter(void *device_u32) {
    g_progress.store((uint32_t *)device_u32);
    return 0;
}

int gptq_prefetch_describe(const void *stripes, int K, int N, int bits, int groupsize, int nsets, gptq_prefetch_op_t *op) {
    if (!stripes || !op) return GPTQ_E_NULL;
    const size_t total = stripe_total_bytes(K, N, bits, groupsize, nsets);
    if (total == 0 || (N / 16) % 8 != 0) return GPTQ_E_VARIANT;
    const size_t toff = stripe_tab_offset(K, N, bits, nsets);
    op->weights = stripes;
    op->table = (const char *)stripes + toff;
    op->nstripes = (uint32_t)(N / 16);
    op->stripe_bytes = (uint32_t)(toff / (size_t)(N / 16));
    op->table_stripe_bytes = (uint32_t)((total - toff) / (size_t)(N / 16));
    op->reserved = 0;
    return 0;
}

int gptq_prefetch_launch(const gptq_prefetch_op_t *plan_device, int first, int last, const void *progress_device, uint32_t progress_base, int lead,
                         int head_kib, int blocks_per_xcd, int depth, int affinity, uint32_t spin_limit, void *status_device,
                         gptq_stream_t stream) {
    if (!plan_device) return GPTQ_E_NULL;
    if (first < 0 || last < first || lead < 0 || head_kib < 0 || blocks_per_xcd < 1 || blocks_per_xcd > 256) return GPTQ_E_SHAPE;
    if (last == first) return 0;
    const dim3 grid(8 * blocks_per_xcd), block(PF_WAVES * 64);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, block, 0, (hipStream_t)stream, plan_device, first, last, (const uint32_t *)progress_device, progress_base, lead,
                           (uint32_t)head_kib, affinity, spin_limit, (uint32_t *)status_device);
        return (int)hipGetLastError();
    };
    switch (depth) {
        case 4: return go(prefetch_kernel<4>);
        case 8: return go(prefetch_kernel<8>);
        case 16: return go(prefetch_kernel<16>);
        case 32: return go(prefetch_kernel<32>);
        default: return GPTQ_E_VARIANT;
    }
}

}  // extern "C"
