// warmlab.hip -- round 4 lab (development tool, not shipped): how fast does the PRODUCT's batch-1 decode kernel read weights that are
// already on chip, and what does a run-ahead prefetcher (csrc/prefetch.hip) on a second stream buy the dependent decode chain?
//
// Corrects round 3's warm-weights measurement (tools/stripelab.hip time_graph divided the time of `reps` hipGraphLaunch calls over
// 1-3 kernel nodes, i.e. it measured the cadence of launching a tiny graph): every graph here holds >= 256 kernel nodes that cycle
// through the weight sets, and the figure is microseconds per NODE.
//
//   warmlab A            per shape: us per launch of gptq_stripe_matvec_f16 against the working set (1 .. n weight sets cycled inside
//                        one graph): <= 32 MiB = the L2s, <= 256 MiB = the Infinity Cache, beyond = HBM
//   warmlab B [opts]     the LLaMA-7B decode pass (32 x [qkv, o, gate/up, down], 3.37 GB of distinct images) as ONE graph on stream 1,
//                        alone and with the persistent prefetcher on stream 2 (paced by the progress tick of the decode kernels); sweep of
//                        lead / head KiB / workgroups / depth / XCD affinity
//   warmlab C [opts]     the same pass with per-op prefetch launches as forked graph nodes (edges only FROM the chain INTO the prefetch
//                        branch: op i done -> prefetch of op i + 2 may start; the chain itself never waits)
//   (A / B / C results of round 4: profiles/r4a_warm/)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/warmlab tools/warmlab.hip -Lgptq-for-llama_amd/lib -lgptq_mi355x -Wl,-rpath,'$ORIGIN/../gptq-for-llama_amd/lib'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/gptq_mi355x.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define RC(x) do { int r_ = (x); if (r_ != 0) { printf("gptq error %d (%s) at line %d\n", r_, gptq_strerror(r_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return a;
}
__global__ void fill_u32(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = hash32((uint32_t)i * 2654435761u + seed);
}
// table entry: half2 {scale in [0.001, 0.011], zero + 1 in [1, 16]}
__global__ void fill_tab(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t h = hash32((uint32_t)i + seed);
        const _Float16 s = (_Float16)(0.001f + 0.01f * (h >> 8) * (1.0f / 16777216.0f));
        const _Float16 z = (_Float16)(1.0f + (float)(h & 15u));
        uint16_t a, b;
        __builtin_memcpy(&a, &s, 2); __builtin_memcpy(&b, &z, 2);
        p[i] = (uint32_t)a | ((uint32_t)b << 16);
    }
}
__global__ void fill_x(_Float16 *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < 12; j++) s += (hash32((uint32_t)(i * 12 + j) + seed) >> 8) * (1.0f / 16777216.0f);
        p[i] = (_Float16)(s - 6.0f);
    }
}


// ---- the run-ahead prefetcher (lab copy: it lost, so the product does not carry it; round-4 commit 298b562 had it as csrc/prefetch.hip) ----
// Reads (and discards) the head of every stripe of op j + lead while op j runs, paced by a counter the decode kernels tick when they
// start (debug hook gptq_set_progress_counter).  LDS-DMA loads into one scratch slot per wave: no VGPR destinations, DEPTH KiB in flight
// per wave, nothing to consume.  Every spin is bounded.
struct gptq_prefetch_op_t {
    const void *weights;          /* R  [nstripes][stripe_bytes]        */
    const void *table;            /* tab [nstripes][table_stripe_bytes]  */
    uint32_t nstripes, stripe_bytes, table_stripe_bytes, reserved;
};
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int PF_WAVES = 4;

// one 1-KiB wave load of `base + off` (clamped to the last whole KiB of the region), DEPTH - 1 older ones may stay in flight
template <int DEPTH>
static __device__ __forceinline__ void touch_kib(const char *base, size_t off, size_t region_bytes, unsigned char *slot, int lane) {
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


// 4-bit g128 stripe16 image: R [N/16][K/128][nsets][64][4] uint32, then tab half2 [N/16][nsets][K/128][16]
static int gptq_prefetch_describe(const void *stripes, int K, int N, int bits, int groupsize, int nsets, gptq_prefetch_op_t *op) {
    if (bits != 4 || groupsize != 128 || (N / 16) % 8 != 0) return -1;
    op->weights = stripes;
    op->nstripes = (uint32_t)(N / 16);
    op->stripe_bytes = (uint32_t)(K / 128) * nsets * 1024u;
    op->table_stripe_bytes = (uint32_t)nsets * (K / 128) * 64u;
    op->table = (const char *)stripes + (size_t)op->nstripes * op->stripe_bytes;
    op->reserved = 0;
    return 0;
}
static int gptq_prefetch_launch(const gptq_prefetch_op_t *plan_device, int first, int last, const void *progress_device, uint32_t progress_base, int lead,
                                int head_kib, int blocks_per_xcd, int depth, int affinity, uint32_t spin_limit, void *status_device, hipStream_t stream) {
    if (last == first) return 0;
    const dim3 grid(8 * blocks_per_xcd), block(PF_WAVES * 64);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, block, 0, stream, plan_device, first, last, (const uint32_t *)progress_device, progress_base, lead,
                           (uint32_t)head_kib, affinity, spin_limit, (uint32_t *)status_device);
        return (int)hipGetLastError();
    };
    switch (depth) {
        case 4: return go(prefetch_kernel<4>);
        case 8: return go(prefetch_kernel<8>);
        case 16: return go(prefetch_kernel<16>);
        case 32: return go(prefetch_kernel<32>);
        default: return -2;
    }
}

struct Image { void *p; size_t bytes; int K, N, nsets; };

static Image make_image(int K, int N, int nsets, uint32_t seed, hipStream_t s) {
    Image im{nullptr, gptq_stripe_bytes(K, N, 4, 128, nsets), K, N, nsets};
    if (im.bytes == 0) { printf("no image for %dx%d\n", K, N); exit(1); }
    CK(hipMalloc(&im.p, im.bytes));
    gptq_prefetch_op_t op;
    RC(gptq_prefetch_describe(im.p, K, N, 4, 128, nsets, &op));
    const size_t rbytes = (size_t)op.nstripes * op.stripe_bytes, tbytes = (size_t)op.nstripes * op.table_stripe_bytes;
    if (rbytes + tbytes != im.bytes) { printf("layout mismatch %zu + %zu != %zu\n", rbytes, tbytes, im.bytes); exit(1); }
    hipLaunchKernelGGL(fill_u32, dim3(2048), dim3(256), 0, s, (uint32_t *)im.p, rbytes / 4, 1000u + seed);
    hipLaunchKernelGGL(fill_tab, dim3(256), dim3(256), 0, s, (uint32_t *)((char *)im.p + rbytes), tbytes / 4, 2000u + seed);
    return im;
}

static double alg_bytes(int K, int N, int nsets) {
    const double G = K / 128;
    return nsets * ((double)(K / 8) * N * 4 + G * (N / 8) * 4 + G * N * 2) + 2.0 * K + 2.0 * N;
}

static void matvec(const Image &im, const void *x, void *y, hipStream_t s) {
    RC(gptq_stripe_matvec_f16(x, im.K, im.p, im.bytes, nullptr, y, im.N, 1, im.K, im.N, 4, 128, im.nsets, nullptr, 0.f, nullptr, s));
}

static float time_exec(hipGraphExec_t ge, hipStream_t s, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

// ------------------------------------------------------------------------------------------------------------------ A
static void exp_a(hipStream_t s) {
    struct Shape { int K, N, nsets; const char *name; };
    const Shape shapes[] = {{4096, 4096, 1, "o 4096x4096"}, {4096, 12288, 1, "qkv 4096x12288"}, {4096, 11008, 2, "gate/up 2x4096x11008"},
                            {11008, 4096, 1, "down 11008x4096"}};
    printf("# A: us per launch of the product decode kernel (gptq_stripe_matvec_f16, M = 1) against the working set; >= 256 kernel nodes per graph\n");
    for (const Shape &sh : shapes) {
        const double bytes = alg_bytes(sh.K, sh.N, sh.nsets);
        const size_t ib = gptq_stripe_bytes(sh.K, sh.N, 4, 128, sh.nsets);
        const int nmax = (int)((600ull << 20) / ib) + 1;
        std::vector<Image> sets;
        for (int i = 0; i < nmax; i++) sets.push_back(make_image(sh.K, sh.N, sh.nsets, i, s));
        _Float16 *x, *y;
        CK(hipMalloc(&x, sh.K * 2)); CK(hipMalloc(&y, sh.N * 2));
        hipLaunchKernelGGL(fill_x, dim3(64), dim3(256), 0, s, x, (size_t)sh.K, 77u);
        CK(hipStreamSynchronize(s));
        printf("== %s: %.2f MB algorithmic, image %.2f MB, up to %d sets\n", sh.name, bytes / 1e6, ib / 1e6, nmax);
        std::vector<int> counts;
        for (int n = 1; n <= nmax; n = n < 4 ? n + 1 : (n * 3 + 1) / 2) counts.push_back(n);
        if (counts.back() != nmax) counts.push_back(nmax);
        for (int n : counts) {
            const int nodes = ((256 + n - 1) / n) * n;
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < nodes; i++) matvec(sets[i % n], x, y, s);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            float best = 1e9f;
            for (int r = 0; r < 3; r++) best = std::min(best, time_exec(ge, s, 4));
            const float us = best / nodes;
            printf("   sets %3d  working set %7.1f MB  %6.3f us/launch  %6.0f GB/s  (%.3f of 8 TB/s)\n", n, n * ib / 1e6, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        for (auto &im : sets) CK(hipFree(im.p));
        CK(hipFree(x)); CK(hipFree(y));
    }
}

// ------------------------------------------------------------------------------------------------------------------ B / C
struct Pass {
    std::vector<Image> ops;   // 128 images in chain order
    std::vector<const _Float16 *> xs;
    std::vector<_Float16 *> ys;
    gptq_prefetch_op_t *plan_dev = nullptr;
    double bytes = 0;
};

static Pass make_pass(int layers, hipStream_t s) {
    Pass P;
    _Float16 *xh, *xi, *yq, *yh, *yi;
    CK(hipMalloc(&xh, 4096 * 2)); CK(hipMalloc(&xi, 11008 * 2)); CK(hipMalloc(&yq, 12288 * 2)); CK(hipMalloc(&yh, 4096 * 2)); CK(hipMalloc(&yi, 11008 * 2));
    hipLaunchKernelGGL(fill_x, dim3(64), dim3(256), 0, s, xh, (size_t)4096, 77u);
    hipLaunchKernelGGL(fill_x, dim3(64), dim3(256), 0, s, xi, (size_t)11008, 78u);
    for (int l = 0; l < layers; l++) {
        P.ops.push_back(make_image(4096, 12288, 1, 4 * l, s)); P.xs.push_back(xh); P.ys.push_back(yq);
        P.ops.push_back(make_image(4096, 4096, 1, 4 * l + 1, s)); P.xs.push_back(xh); P.ys.push_back(yh);
        P.ops.push_back(make_image(4096, 11008, 2, 4 * l + 2, s)); P.xs.push_back(xh); P.ys.push_back(yi);
        P.ops.push_back(make_image(11008, 4096, 1, 4 * l + 3, s)); P.xs.push_back(xi); P.ys.push_back(yh);
    }
    std::vector<gptq_prefetch_op_t> plan(P.ops.size());
    for (size_t i = 0; i < P.ops.size(); i++) {
        RC(gptq_prefetch_describe(P.ops[i].p, P.ops[i].K, P.ops[i].N, 4, 128, P.ops[i].nsets, &plan[i]));
        P.bytes += alg_bytes(P.ops[i].K, P.ops[i].N, P.ops[i].nsets);
    }
    CK(hipMalloc(&P.plan_dev, plan.size() * sizeof(plan[0])));
    CK(hipMemcpy(P.plan_dev, plan.data(), plan.size() * sizeof(plan[0]), hipMemcpyHostToDevice));
    CK(hipStreamSynchronize(s));
    return P;
}

static hipGraphExec_t capture_chain(const Pass &P, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (size_t i = 0; i < P.ops.size(); i++) matvec(P.ops[i], P.xs[i], P.ys[i], s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return ge;
}

struct PfCfg { int lead, head_kib, bpx, depth, affinity; };

// `reps` passes back to back: per pass the prefetcher goes out on s2, the chain graph on s1; the timed region ends when both streams are done.
static float time_with_prefetcher(const Pass &P, hipGraphExec_t chain, hipStream_t s1, hipStream_t s2, uint32_t *progress, uint32_t *status,
                                  const PfCfg &c, int reps, uint32_t *gave_up) {
    const int nops = (int)P.ops.size();
    CK(hipMemsetAsync(progress, 0, 4, s1)); CK(hipMemsetAsync(status, 0, 4, s1));
    CK(hipStreamSynchronize(s1));
    hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    uint32_t base = 0;
    auto one = [&]() {
        RC(gptq_prefetch_launch(P.plan_dev, 0, nops, progress, base, c.lead, c.head_kib, c.bpx, c.depth, c.affinity, 6000u, status, s2));
        CK(hipGraphLaunch(chain, s1));
        base += (uint32_t)nops;
    };
    one(); one();
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    CK(hipEventRecord(e0, s1));
    CK(hipStreamWaitEvent(s2, e0, 0));   // the first timed prefetcher does not start before the clock
    for (int i = 0; i < reps; i++) one();
    CK(hipEventRecord(ej, s2));
    CK(hipStreamWaitEvent(s1, ej, 0));
    CK(hipEventRecord(e1, s1));
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(gave_up, status, 4, hipMemcpyDeviceToHost));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipEventDestroy(ej));
    return ms * 1e3f / reps;
}

static void exp_b(hipStream_t s1, int argc, char **argv) {
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int layers = getenv("LAB_LAYERS") ? atoi(getenv("LAB_LAYERS")) : 32;
    Pass P = make_pass(layers, s1);
    uint32_t *progress, *status;
    CK(hipMalloc(&progress, 256)); CK(hipMalloc(&status, 256));
    CK(hipMemset(progress, 0, 256)); CK(hipMemset(status, 0, 256));
    printf("# B: LLaMA-7B decode pass, %zu launches, %.3f GB algorithmic per pass\n", P.ops.size(), P.bytes / 1e9);
    // 1. the chain alone, no tick
    RC(gptq_set_progress_counter(nullptr));
    hipGraphExec_t plain = capture_chain(P, s1);
    float t0 = 1e9f;
    for (int r = 0; r < 3; r++) t0 = std::min(t0, time_exec(plain, s1, 10));
    printf("   chain alone (no tick)             %8.1f us/pass  %6.0f GB/s  %.4f of 8 TB/s\n", t0, P.bytes / t0 / 1e3, P.bytes / t0 / 1e3 / 8000.0);
    // 2. the chain with the tick, still alone
    RC(gptq_set_progress_counter(progress));
    hipGraphExec_t ticked = capture_chain(P, s1);
    RC(gptq_set_progress_counter(nullptr));
    float t1 = 1e9f;
    for (int r = 0; r < 3; r++) t1 = std::min(t1, time_exec(ticked, s1, 10));
    printf("   chain alone (progress tick)       %8.1f us/pass  %6.0f GB/s  %.4f\n", t1, P.bytes / t1 / 1e3, P.bytes / t1 / 1e3 / 8000.0);
    // 3. unpaced prefetcher alone: how fast can it pull the whole pass (= the HBM stream rate of this kernel)
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int bpx : {16, 32, 64}) for (int depth : {8, 16}) {
            RC(gptq_prefetch_launch(P.plan_dev, 0, (int)P.ops.size(), nullptr, 0, 0, 0, bpx, depth, 1, 0, nullptr, s1));
            CK(hipStreamSynchronize(s1));
            CK(hipEventRecord(e0, s1));
            for (int i = 0; i < 3; i++) RC(gptq_prefetch_launch(P.plan_dev, 0, (int)P.ops.size(), nullptr, 0, 0, 0, bpx, depth, 1, 0, nullptr, s1));
            CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const float us = ms * 1e3f / 3;
            printf("   prefetcher alone, unpaced: %3d wg/xcd depth %2d  %8.1f us/pass  %6.0f GB/s\n", bpx, depth, us, P.bytes / us / 1e3);
        }
    }
    // 4. sweep
    std::vector<PfCfg> cfgs;
    if (argc > 2) {
        for (int i = 2; i + 4 < argc; i += 5) cfgs.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), atoi(argv[i + 4])});
    } else {
        for (int lead : {1, 2, 3})
            for (int head : {4, 8, 16, 32, 0})
                cfgs.push_back({lead, head, 32, 8, 1});
        for (int bpx : {8, 16, 64}) cfgs.push_back({2, 16, bpx, 8, 1});
        for (int depth : {4, 16, 32}) cfgs.push_back({2, 16, 32, depth, 1});
        cfgs.push_back({2, 16, 32, 8, 0});
        cfgs.push_back({2, 0, 32, 8, 0});
        cfgs.push_back({2, 0, 64, 16, 1});
        cfgs.push_back({4, 0, 32, 8, 1});
    }
    printf("   lead head_KiB wg/xcd depth affinity |  us/pass    GB/s   of 8TB/s  vs alone  gave_up\n");
    for (const PfCfg &c : cfgs) {
        uint32_t gu = 0;
        float t = 1e9f;
        for (int r = 0; r < 2; r++) t = std::min(t, time_with_prefetcher(P, ticked, s1, s2, progress, status, c, 10, &gu));
        printf("   %4d %8d %6d %5d %8d | %8.1f  %6.0f   %.4f   %.3f   %u\n", c.lead, c.head_kib, c.bpx, c.depth, c.affinity, t, P.bytes / t / 1e3,
               P.bytes / t / 1e3 / 8000.0, t1 / t, gu);
        fflush(stdout);
    }
}

// C: prefetch launches as forked nodes of the SAME graph: after op i is enqueued, the side stream waits for an event recorded behind op i
// and launches the (unpaced) prefetch of op i + lead + 1; the chain never waits for the side stream except at the very end (capture join).
static void exp_c(hipStream_t s1) {
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    Pass P = make_pass(32, s1);
    const int nops = (int)P.ops.size();
    RC(gptq_set_progress_counter(nullptr));
    hipGraphExec_t plain = capture_chain(P, s1);
    float t0 = 1e9f;
    for (int r = 0; r < 3; r++) t0 = std::min(t0, time_exec(plain, s1, 10));
    printf("# C: forked prefetch nodes inside ONE graph\n   chain alone %8.1f us/pass  %6.0f GB/s\n", t0, P.bytes / t0 / 1e3);
    printf("   lead head_KiB wg/xcd depth |  us/pass    GB/s   of 8TB/s  vs alone\n");
    for (int lead : {1, 2})
        for (int head : {8, 32, 0})
            for (int bpx : {16, 32}) {
                std::vector<hipEvent_t> evs(nops);
                for (auto &e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
                CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0));
                RC(gptq_prefetch_launch(P.plan_dev, 0, std::min(nops, lead + 1), nullptr, 0, 0, head, bpx, 8, 1, 0, nullptr, s2));
                for (int i = 0; i < nops; i++) {
                    matvec(P.ops[i], P.xs[i], P.ys[i], s1);
                    const int j = i + lead + 1;
                    if (j < nops) {
                        CK(hipEventRecord(evs[i], s1));
                        CK(hipStreamWaitEvent(s2, evs[i], 0));
                        RC(gptq_prefetch_launch(P.plan_dev, j, j + 1, nullptr, 0, 0, head, bpx, 8, 1, 0, nullptr, s2));
                    }
                }
                CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s1, join, 0));
                CK(hipStreamEndCapture(s1, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                float t = 1e9f;
                for (int r = 0; r < 2; r++) t = std::min(t, time_exec(ge, s1, 6));
                printf("   %4d %8d %6d %5d | %8.1f  %6.0f   %.4f   %.3f\n", lead, head, bpx, 8, t, P.bytes / t / 1e3, P.bytes / t / 1e3 / 8000.0, t0 / t);
                fflush(stdout);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
                for (auto &e : evs) CK(hipEventDestroy(e));
            }
}

int main(int argc, char **argv) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const char *what = argc > 1 ? argv[1] : "A";
    if (!strcmp(what, "A")) exp_a(s);
    else if (!strcmp(what, "B")) exp_b(s, argc, argv);
    else if (!strcmp(what, "C")) exp_c(s);
    else { printf("usage: warmlab A | B [lead head_kib wg_per_xcd depth affinity]... | C\n"); return 2; }
    return 0;
}
