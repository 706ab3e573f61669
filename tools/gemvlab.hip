// gemvlab.hip -- standalone laboratory for the batch-1 4-bit g128 dequant-matvec on gfx950.
// Development tool (not shipped): candidate kernel structures are timed cold (rotating over
// > 256 MiB of distinct weight sets inside one hipGraph) and checked against a slow reference
// kernel, so that design decisions in csrc/gemv.hip are backed by a measurement.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemvlab tools/gemvlab.hip
// run  : ./tools/gemvlab [K N]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "../gptq-for-llama_amd/csrc/gptq_device.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash32(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return a;
}
__global__ void fill_u32(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = hash32((uint32_t)i * 2654435761u + seed);
}
__global__ void fill_scales(half_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (half_t)(0.001f + 0.01f * (hash32((uint32_t)i + seed) >> 8) * (1.0f / 16777216.0f));
}
__global__ void fill_x(half_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < 12; j++) s += (hash32((uint32_t)(i * 12 + j) + seed) >> 8) * (1.0f / 16777216.0f);
        p[i] = (half_t)(s - 6.0f);  // ~N(0,1)
    }
}

// slow, obviously-correct reference (reference quant_linear.py:103-130), double accumulation
__global__ void ref_kernel(const half_t *x, const uint32_t *qw, const half_t *sc, const uint32_t *qz, double *y, int K, int N) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double acc = 0;
    for (int k = 0; k < K; k++) {
        int g = k / 128;
        int q = (qw[(size_t)(k / 8) * N + n] >> (4 * (k % 8))) & 15;
        int z = ((qz[(size_t)g * (N / 8) + n / 8] >> (4 * (n % 8))) & 15) + 1;
        acc += (double)(float)x[k] * (double)(q - z) * (double)(float)sc[(size_t)g * N + n];
    }
    y[n] = acc;
}

struct P {
    const half_t *__restrict__ x;
    const uint32_t *__restrict__ qw;
    const half_t *__restrict__ sc;
    const uint32_t *__restrict__ qz;
    half_t *y;
    u64_t *ws;
    float *part;
    u64_t *dbg;
    int K, N, S, nchunk, ntile;
};


GPTQ_DEV u64_t stamp_dep(uint32_t dep) {
    u64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return t;
}
GPTQ_DEV u64_t stamp_real() {
    u64_t t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

// ------------------------------------------------------------------------------------------
// K1 "rowwave": a wave reads whole 1-KiB row segments (64 lanes x 16 B = 256 columns); every
// lane owns 4 columns, so x is wave-uniform and comes through the scalar cache; U rows in
// flight per wave; 4 waves = 32 rows per chunk; K is split over S workgroups per column tile.
// MODE 0: full (one-round-trip atomic combine); 1: partials stored, no combine (timing only);
// 2: loads only (xor), no math; 3: full math but no cross-workgroup output at all.
// ------------------------------------------------------------------------------------------
template <int U, int MODE>
__global__ void __launch_bounds__(256) k_rowwave(const P p) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x % p.ntile, slice = blockIdx.x / p.ntile;
    const int N = p.N;
    const int n0 = tile * 256 + lane * 4;
    float y[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t xo = 0;
    u64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (MODE == 4) { st[0] = stamp_real(); st[1] = stamp_dep(0); }
    const half2_t c1024 = {(half_t)1024.f, (half_t)1024.f}, c64 = {(half_t)64.f, (half_t)64.f}, ones = {(half_t)1.f, (half_t)1.f};
    const uint32_t K0 = sreg_const(0x000F000Fu), K1 = sreg_const(0x00F000F0u);
    const uint32_t M0 = vreg_const(0x64006400u), M1 = vreg_const(0x54005400u);

    for (int c = slice; c < p.nchunk; c += p.S) {
        const int row = c * (4 * U) + wave * U;  // packed row (8 k each), wave-uniform
        u32x4 w[U];
#pragma unroll
        for (int u = 0; u < U; u++) w[u] = __builtin_nontemporal_load((const u32x4 *)(p.qw + (size_t)(row + u) * N + n0));
        if constexpr (MODE == 2) {
#pragma unroll
            for (int u = 0; u < U; u++) xo ^= w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
            continue;
        }
        const int g = row / 16;
        const half4_t s4 = *(const half4_t *)(p.sc + (size_t)g * N + n0);
        const uint32_t zw = p.qz[(size_t)g * (N / 8) + n0 / 8];
        const u32x4 *xq = (const u32x4 *)p.x + row;
        __builtin_amdgcn_sched_barrier(0);  // every load of the chunk is issued before any math
        if constexpr (MODE == 4) st[2] = stamp_dep(0);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float J = 0.f, XS = 0.f;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32x4 xv = xq[u];
            if constexpr (MODE == 4) {
                if (u == 0) { st[3] = stamp_dep(xv[0]); st[4] = stamp_dep(w[0][0]); }
                if (u == U - 1) st[5] = stamp_dep(w[U - 1][0]);
            }
            const half2_t p0 = as_half2((xv[0] & 0xffffu) | (xv[2] << 16));
            const half2_t p1 = as_half2((xv[0] >> 16) | (xv[2] & 0xffff0000u));
            const half2_t p2 = as_half2((xv[1] & 0xffffu) | (xv[3] << 16));
            const half2_t p3 = as_half2((xv[1] >> 16) | (xv[3] & 0xffff0000u));
            J = __builtin_amdgcn_fdot2(p0, c1024, J, false);
            J = __builtin_amdgcn_fdot2(p1, c64, J, false);
            J = __builtin_amdgcn_fdot2(p2, c1024, J, false);
            J = __builtin_amdgcn_fdot2(p3, c64, J, false);
            XS = __builtin_amdgcn_fdot2(p0, ones, XS, false);
            XS = __builtin_amdgcn_fdot2(p1, ones, XS, false);
            XS = __builtin_amdgcn_fdot2(p2, ones, XS, false);
            XS = __builtin_amdgcn_fdot2(p3, ones, XS, false);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t v = w[u][j], v8 = v >> 8;
                acc[j] = __builtin_amdgcn_fdot2(as_half2((v & K0) | M0), p0, acc[j], false);
                acc[j] = __builtin_amdgcn_fdot2(as_half2((v & K1) | M1), p1, acc[j], false);
                acc[j] = __builtin_amdgcn_fdot2(as_half2((v8 & K0) | M0), p2, acc[j], false);
                acc[j] = __builtin_amdgcn_fdot2(as_half2((v8 & K1) | M1), p3, acc[j], false);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float zf = (float)(((zw >> (4 * ((n0 + j) & 7))) & 15u) + 1u);
            y[j] += (float)s4[j] * (acc[j] - J - zf * XS);
        }
    }
    if constexpr (MODE == 2) {
        if (xo == 0x9e3779b9u) p.part[blockIdx.x] = 1.f;
        return;
    }
    if constexpr (MODE == 4) st[6] = stamp_dep(__builtin_bit_cast(uint32_t, y[0]));
    *(float4_t *)&red[wave][4 * lane] = float4_t{y[0], y[1], y[2], y[3]};
    __syncthreads();
    const int t = threadIdx.x;
    float v = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    const int n = tile * 256 + t;
    if constexpr (MODE == 0 || MODE == 4) {
        float tot;
        if (p.S > 1) {
            if (splitk_add1(p.ws + n, v, p.S, tot)) p.y[n] = (half_t)tot;
        } else {
            p.y[n] = (half_t)v;
        }
        if constexpr (MODE == 4) {
            st[7] = stamp_dep(__builtin_bit_cast(uint32_t, tot));
            const u64_t te = stamp_real();
            if (lane == 0) {
                u64_t *d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 10;
#pragma unroll
                for (int i = 0; i < 8; i++) d[i] = st[i];
                d[8] = te;
                uint32_t xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                d[9] = xcc;
            }
        }
    } else if constexpr (MODE == 1) {
        p.part[(size_t)slice * N + n] = v;
    } else {
        if (v == 123.456f) p.part[n] = v;
    }
}


// ------------------------------------------------------------------------------------------
// K2 "stripe": one 1024-thread workgroup owns a 64-byte column stripe (16 columns) over ALL of
// K: no cross-workgroup combine.  lane = 4 column lanes x 16 row lanes per wave, 256 row lanes
// per workgroup, NI rows per lane, all loads in flight at once.  x is staged in LDS once per
// workgroup as {pair-permuted x[8], J, XS} per packed row.  REMAP puts the 8 stripes that share
// 128-byte lines / 512-byte DRAM bursts on the same XCD.  MODE 0 full, 2 loads only.
// ------------------------------------------------------------------------------------------
struct __attribute__((aligned(8))) XRow { uint32_t xp[4]; float J, XS; };

template <int NI, int MODE, bool REMAP, bool NT>
__global__ void __launch_bounds__(1024) k_stripe(const P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    XRow *xr = (XRow *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & 3, kl = tid >> 2;
    const int ntile = p.N / 16;
    int tile = blockIdx.x;
    if (REMAP) { const int per = ntile / 8; tile = (blockIdx.x % 8) * per + blockIdx.x / 8; }
    const int N = p.N, rows = p.K / 8;
    const int n0 = tile * 16 + cg * 4;
    u32x4 w[NI];
    half4_t s4[NI];
    uint32_t zw[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int r = kl + i * 256;
        if (r < rows) {
            const u32x4 *ptr = (const u32x4 *)(p.qw + (size_t)r * N + n0);
            w[i] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
        } else {
            w[i] = u32x4{0, 0, 0, 0};
        }
    }
    if constexpr (MODE == 2) {
        uint32_t xo = 0;
#pragma unroll
        for (int i = 0; i < NI; i++) xo ^= w[i][0] ^ w[i][1] ^ w[i][2] ^ w[i][3];
        if (xo == 0x9e3779b9u) p.part[blockIdx.x] = 1.f;
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int r = kl + i * 256;
        const int g = (r < rows ? r : 0) / 16;
        s4[i] = *(const half4_t *)(p.sc + (size_t)g * N + n0);
        zw[i] = p.qz[(size_t)g * (N / 8) + n0 / 8];
    }
    // stage x: one packed row (8 halves) per thread
    const half2_t c1024 = {(half_t)1024.f, (half_t)1024.f}, c64 = {(half_t)64.f, (half_t)64.f}, ones = {(half_t)1.f, (half_t)1.f};
    for (int r = tid; r < rows; r += 1024) {
        const u32x4 xv = ((const u32x4 *)p.x)[r];
        XRow o;
        o.xp[0] = (xv[0] & 0xffffu) | (xv[2] << 16);
        o.xp[1] = (xv[0] >> 16) | (xv[2] & 0xffff0000u);
        o.xp[2] = (xv[1] & 0xffffu) | (xv[3] << 16);
        o.xp[3] = (xv[1] >> 16) | (xv[3] & 0xffff0000u);
        float J = 0.f, XS = 0.f;
        J = __builtin_amdgcn_fdot2(as_half2(o.xp[0]), c1024, J, false);
        J = __builtin_amdgcn_fdot2(as_half2(o.xp[1]), c64, J, false);
        J = __builtin_amdgcn_fdot2(as_half2(o.xp[2]), c1024, J, false);
        J = __builtin_amdgcn_fdot2(as_half2(o.xp[3]), c64, J, false);
#pragma unroll
        for (int q = 0; q < 4; q++) XS = __builtin_amdgcn_fdot2(as_half2(o.xp[q]), ones, XS, false);
        o.J = J; o.XS = XS;
        xr[r] = o;
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    const uint32_t K0 = sreg_const(0x000F000Fu), K1 = sreg_const(0x00F000F0u);
    const uint32_t M0 = vreg_const(0x64006400u), M1 = vreg_const(0x54005400u);
    float y[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int r = kl + i * 256;
        if (r < rows) {
            const XRow xx = xr[r];
            const half2_t p0 = as_half2(xx.xp[0]), p1 = as_half2(xx.xp[1]), p2 = as_half2(xx.xp[2]), p3 = as_half2(xx.xp[3]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t v = w[i][j], v8 = v >> 8;
                float a = 0.f;
                a = __builtin_amdgcn_fdot2(as_half2((v & K0) | M0), p0, a, false);
                a = __builtin_amdgcn_fdot2(as_half2((v & K1) | M1), p1, a, false);
                a = __builtin_amdgcn_fdot2(as_half2((v8 & K0) | M0), p2, a, false);
                a = __builtin_amdgcn_fdot2(as_half2((v8 & K1) | M1), p3, a, false);
                const float zf = (float)(((zw[i] >> (4 * ((n0 + j) & 7))) & 15u) + 1u);
                y[j] += (float)s4[i][j] * (a - xx.J - zf * xx.XS);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) y[j] = wave_sum_xor(y[j], 4);
    __syncthreads();
    float *red = (float *)smem;  // [16 waves][16 cols]
    if (lane < 4) *(float4_t *)&red[wave * 16 + lane * 4] = float4_t{y[0], y[1], y[2], y[3]};
    __syncthreads();
    if (tid < 16) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) v += red[q * 16 + tid];
        p.y[tile * 16 + tid] = (half_t)v;
    }
}

template <int NI, int MODE, bool REMAP, bool NT>
static void launch_stripe(const P &p, hipStream_t s) {
    const size_t lds = (size_t)(p.K / 8) * sizeof(XRow) + 1024;
    hipLaunchKernelGGL((k_stripe<NI, MODE, REMAP, NT>), dim3(p.N / 16), dim3(1024), lds, s, p);
}

// touches one dword per `stride` bytes of a buffer from one workgroup per XCD (TLB warm-up probe)
__global__ void touch_kernel(const uint32_t *p, size_t bytes, size_t stride, float *out) {
    uint32_t acc = 0;
    for (size_t o = (size_t)threadIdx.x * stride; o < bytes; o += (size_t)blockDim.x * stride) acc ^= __builtin_nontemporal_load(p + o / 4);
    if (acc == 0x9e3779b9u) out[blockIdx.x] = 1.f;
}
__global__ void empty_kernel(float *out) { if (threadIdx.x == 1025) out[0] = 1; }

// ------------------------------------------------------------------------------------------
struct WSet { uint32_t *qw; half_t *sc; uint32_t *qz; };

typedef void (*launch_fn)(const P &, hipStream_t);

template <int U, int MODE>
static void launch_rowwave(const P &p, hipStream_t s) {
    hipLaunchKernelGGL((k_rowwave<U, MODE>), dim3(p.ntile * p.S), dim3(256), 0, s, p);
}

static float time_graph(launch_fn fn, P base, const std::vector<WSet> &sets, hipStream_t s, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (auto &w : sets) { P p = base; p.qw = w.qw; p.sc = w.sc; p.qz = w.qz; fn(p, s); }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (reps * sets.size());
}

// the same launches captured on `nch` forked streams (kernel j on chain j % nch, each chain with its
// own split-K workspace and output): do independent chains of a hipGraph overlap on this stack?
static float time_graph_chains(launch_fn fn, P base, const std::vector<WSet> &sets, hipStream_t s, int reps, int nch) {
    static hipStream_t side[8];
    static bool init = false;
    if (!init) { for (auto &x : side) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking)); init = true; }
    hipGraph_t g; hipGraphExec_t ge;
    hipEvent_t fork, join[8];
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (auto &e : join) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    CK(hipEventRecord(fork, s));
    for (int c = 1; c < nch; c++) CK(hipStreamWaitEvent(side[c], fork, 0));
    int j = 0;
    for (auto &w : sets) {
        const int c = j++ % nch;
        P p = base; p.qw = w.qw; p.sc = w.sc; p.qz = w.qz; p.ws = base.ws + (size_t)c * 16384; p.y = base.y + (size_t)c * 0;
        fn(p, c == 0 ? s : side[c]);
    }
    for (int c = 1; c < nch; c++) { CK(hipEventRecord(join[c], side[c])); CK(hipStreamWaitEvent(s, join[c], 0)); }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (reps * sets.size());
}

// ------------------------------------------------------------------------------------------
// L2 prefetch of the NEXT launch's weights by a concurrent kernel on a forked graph branch: same grid and the same
// block -> (tile, slice) -> address mapping as k_rowwave, so block b (XCD b % 8) pulls exactly the lines block b of
// the next GEMV will read into the L2 of that XCD.  Cached (not nt) loads, results discarded.  `limit` = chunks per
// slice to prefetch (L2 is 4 MiB per XCD).
template <int U>
__global__ void __launch_bounds__(256) k_prefetch(const P p, int limit) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x % p.ntile, slice = blockIdx.x / p.ntile;
    const int N = p.N;
    const int n0 = tile * 256 + lane * 4;
    uint32_t xo = 0;
    int done = 0;
    for (int c = slice; c < p.nchunk && done < limit; c += p.S, done++) {
        const int row = c * (4 * U) + wave * U;
        u32x4 w[U];
#pragma unroll
        for (int u = 0; u < U; u++) w[u] = *(const u32x4 *)(p.qw + (size_t)(row + u) * N + n0);
#pragma unroll
        for (int u = 0; u < U; u++) xo ^= w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
    }
    if (xo == 0x9e3779b9u && p.part) p.part[blockIdx.x] = 1.f;
}

// G_0 -> G_1 -> ... on the main stream; P_i (prefetch of set i+1) on a side stream, released together with G_i
template <int U>
static float time_graph_prefetch(launch_fn fn, P base, const std::vector<WSet> &sets, hipStream_t s, int reps, int limit) {
    static hipStream_t side = nullptr;
    if (!side) CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    std::vector<hipEvent_t> ev(sets.size());
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t join; CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (size_t i = 0; i < sets.size(); i++) {
        CK(hipEventRecord(ev[i], s));                      // G_{i-1} has finished
        CK(hipStreamWaitEvent(side, ev[i], 0));
        if (i + 1 < sets.size()) {
            P q = base; q.qw = sets[i + 1].qw;
            hipLaunchKernelGGL((k_prefetch<U>), dim3(q.ntile * q.S), dim3(256), 0, side, q, limit);
        }
        P p = base; p.qw = sets[i].qw; p.sc = sets[i].sc; p.qz = sets[i].qz;
        fn(p, s);
    }
    CK(hipEventRecord(join, side)); CK(hipStreamWaitEvent(s, join, 0));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (reps * sets.size());
}

// ------------------------------------------------------------------------------------------
// "column stripes on a repacked layout": what would the GEMV cost WITHOUT a K split (no combine atomics, no workspace)
// if every workgroup owned 16 whole output columns and its weight slice were contiguous in memory (a load-time
// repack: buf[wg][packed row][16 columns])?  One wave instruction = 16 packed rows x 16 columns (1 KiB, contiguous);
// lane l -> row l / 4, columns 4 (l % 4) .. +3; x comes from LDS (rows differ per lane); reduce over the 16 row
// lanes with shuffles, over the 4 waves through LDS, store y directly.  Group = 128 k = one row block.
__global__ void __launch_bounds__(256) k_colstripe(const P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_smem[];
    half_t *xl = (half_t *)cs_smem;                      // x[K]
    float *red = (float *)(cs_smem + (size_t)p.K * 2);   // [4][16]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x, N = p.N, rows = p.K / 8, nrb = rows / 16;
    const uint32_t *slice = p.qw + (size_t)wg * rows * 16;
    const int col0 = wg * 16 + 4 * (lane & 3);
    const half2_t ones = {(half_t)1.f, (half_t)1.f};
    const uint32_t MSK = sreg_const(0x00F000F0u), MAG = vreg_const(0x54005400u);
    constexpr int U = 8;
    float y[4] = {0.f, 0.f, 0.f, 0.f};
    bool staged = false;
    for (int rb0 = wave; rb0 < nrb; rb0 += 4 * U) {
        u32x4 w[U];
        half4_t s4[U];
        uint32_t zw[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int rb = rb0 + 4 * u;
            const int rbc = rb < nrb ? rb : nrb - 1;
            w[u] = __builtin_nontemporal_load((const u32x4 *)(slice + (size_t)rbc * 256 + lane * 4));
            s4[u] = *(const half4_t *)(p.sc + (size_t)rbc * N + col0);
            zw[u] = p.qz[(size_t)rbc * (N / 8) + col0 / 8];
        }
        if (!staged) {   // x -> LDS once, behind the first weight loads
            for (int i = threadIdx.x; i < p.K / 8; i += 256) *(u32x4 *)(xl + (size_t)i * 8) = *(const u32x4 *)(p.x + (size_t)i * 8);
            __syncthreads();
            staged = true;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int rb = rb0 + 4 * u;
            if (rb >= nrb) break;
            const int r = rb * 16 + (lane >> 2);
            const u32x4 v = *(const u32x4 *)(xl + (size_t)r * 8);
            half2_t X[4];
            X[0] = as_half2(__builtin_amdgcn_perm(v[2], v[0], 0x05040100u)); X[1] = as_half2(__builtin_amdgcn_perm(v[2], v[0], 0x07060302u));
            X[2] = as_half2(__builtin_amdgcn_perm(v[3], v[1], 0x05040100u)); X[3] = as_half2(__builtin_amdgcn_perm(v[3], v[1], 0x07060302u));
            float xs = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) xs = __builtin_amdgcn_fdot2(X[q], ones, xs, false);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t ww = w[u][j];
                half2_t t[4];
                t[0] = as_half2(((ww << 4) & MSK) | MAG); t[1] = as_half2((ww & MSK) | MAG);
                t[2] = as_half2(((ww >> 4) & MSK) | MAG); t[3] = as_half2(((ww >> 8) & MSK) | MAG);
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 4; q++) a = __builtin_amdgcn_fdot2(t[q], X[q], a, false);
                const float zf = (float)(((zw[u] >> (4 * ((col0 + j) & 7))) & 15u) + 1u) + 64.0f;
                y[j] += (float)s4[u][j] * (a - zf * xs);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) y[j] += __shfl_xor(y[j], off, 64);
    }
    if (lane < 4) {
#pragma unroll
        for (int j = 0; j < 4; j++) red[wave * 16 + 4 * lane + j] = y[j];
    }
    __syncthreads();
    if (threadIdx.x < 16) p.y[wg * 16 + threadIdx.x] = (half_t)(red[threadIdx.x] + red[16 + threadIdx.x] + red[32 + threadIdx.x] + red[48 + threadIdx.x]);
}
static void launch_colstripe(const P &p, hipStream_t s) {
    hipLaunchKernelGGL(k_colstripe, dim3(p.N / 16), dim3(256), (size_t)p.K * 2 + 256, s, p);
}
// logical [rows][N] view of a buffer that k_colstripe reads as [N/16][rows][16] (for the reference check only)
__global__ void unrepack_kernel(const uint32_t *buf, uint32_t *logical, int rows, int N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * N) return;
    const int r = (int)(i / N), n = (int)(i % N);
    logical[i] = buf[((size_t)(n / 16) * rows + r) * 16 + (n % 16)];
}

int main(int argc, char **argv) {
    hipStream_t s; CK(hipStreamCreate(&s));
    int shapes[][2] = {{4096, 4096}, {4096, 12288}, {11008, 4096}, {4096, 11008}};
    int nshape = 4;
    if (argc >= 3) { shapes[0][0] = atoi(argv[1]); shapes[0][1] = atoi(argv[2]); nshape = 1; }
    float *part; CK(hipMalloc(&part, 64 << 20));
    u64_t *ws; CK(hipMalloc(&ws, 1 << 20)); CK(hipMemset(ws, 0, 1 << 20));
    for (int si = 0; si < nshape; si++) {
        const int K = shapes[si][0], N = shapes[si][1], G = K / 128;
        const size_t qw_n = (size_t)(K / 8) * N, sc_n = (size_t)G * N, qz_n = (size_t)G * (N / 8);
        const size_t bytes = qw_n * 4 + sc_n * 2 + qz_n * 4 + 2 * K + 2 * N;
        int nbuf = (int)((400ull << 20) / bytes) + 1;
        if (getenv("LAB_NBUF")) nbuf = atoi(getenv("LAB_NBUF"));
        const bool big = getenv("LAB_BIGALLOC") != nullptr;
        std::vector<WSet> sets(nbuf);
        char *arena = nullptr;
        const size_t per = ((qw_n * 4 + sc_n * 2 + qz_n * 4) + 4095) & ~(size_t)4095;
        if (big) CK(hipMalloc(&arena, per * nbuf));
        for (int i = 0; i < nbuf; i++) {
            if (big) {
                sets[i].qw = (uint32_t *)(arena + per * i); sets[i].sc = (half_t *)(arena + per * i + qw_n * 4); sets[i].qz = (uint32_t *)(arena + per * i + qw_n * 4 + sc_n * 2);
            } else {
                CK(hipMalloc(&sets[i].qw, qw_n * 4)); CK(hipMalloc(&sets[i].sc, sc_n * 2)); CK(hipMalloc(&sets[i].qz, qz_n * 4));
            }
            hipLaunchKernelGGL(fill_u32, dim3(2048), dim3(256), 0, s, sets[i].qw, qw_n, 1000u + i);
            hipLaunchKernelGGL(fill_scales, dim3(256), dim3(256), 0, s, sets[i].sc, sc_n, 2000u + i);
            hipLaunchKernelGGL(fill_u32, dim3(256), dim3(256), 0, s, sets[i].qz, qz_n, 3000u + i);
        }
        half_t *x, *y; double *yref;
        CK(hipMalloc(&x, K * 2)); CK(hipMalloc(&y, N * 2)); CK(hipMalloc(&yref, N * 8));
        hipLaunchKernelGGL(fill_x, dim3(64), dim3(256), 0, s, x, (size_t)K, 77u);
        CK(hipStreamSynchronize(s));
        printf("== K=%d N=%d: %.2f MB algorithmic x %d sets\n", K, N, bytes / 1e6, nbuf);

        P base{};
        base.x = x; base.y = y; base.ws = ws; base.part = part; base.K = K; base.N = N;
        base.ntile = N / 256;
        const int rows = K / 8;
        std::vector<double> href(N); std::vector<half_t> hy(N);

        auto check = [&](launch_fn fn, P p) -> double {
            CK(hipMemsetAsync(y, 0, N * 2, s));
            p.qw = sets[1 % sets.size()].qw; p.sc = sets[1 % sets.size()].sc; p.qz = sets[1 % sets.size()].qz;
            hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256), dim3(256), 0, s, x, p.qw, p.sc, p.qz, yref, K, N);
            fn(p, s);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(href.data(), yref, N * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hy.data(), y, N * 2, hipMemcpyDeviceToHost));
            double mx = 0, err = 0;
            for (int n = 0; n < N; n++) { mx = fmax(mx, fabs(href[n])); err = fmax(err, fabs(href[n] - (double)(float)hy[n])); }
            return err / mx;
        };

        if (N % 256 == 0) {
            if (rows % 64 == 0) {
                constexpr int U = 16;
                P p = base; p.nchunk = rows / (4 * U); p.S = p.nchunk;
                double e = check(launch_rowwave<U, 0>, p);
                float t0 = time_graph(launch_rowwave<U, 0>, p, sets, s, 5);
                float t2 = time_graph(launch_rowwave<U, 2>, p, sets, s, 5);
                printf("  rowwave U16 S%-3d wgs %5d | full %6.2f us %5.0f GB/s err %.1e | loadsonly %6.2f\n", p.S, p.ntile * p.S, t0, bytes / t0 / 1e3, e, t2);
            }
            int Ss[] = {16};
            for (int S : Ss) {
                do {
                    constexpr int U = 8;
                    if (rows % (4 * U)) break;
                    P p = base; p.nchunk = rows / (4 * U); p.S = S;
                    if (S > p.nchunk) break;
                    if (S > SPLITK_MAX_SINGLE) {
                        float t2 = time_graph(launch_rowwave<U, 2>, p, sets, s, 5);
                        float t1 = time_graph(launch_rowwave<U, 1>, p, sets, s, 5);
                        printf("  rowwave U8 S%-3d wgs %5d |                          | noatomic %6.2f us | loadsonly %6.2f us\n", S, p.ntile * S, t1, t2);
                        break;
                    }
                    double e = check(launch_rowwave<U, 0>, p);
                    float t0 = time_graph(launch_rowwave<U, 0>, p, sets, s, 5);
                    float t1 = time_graph(launch_rowwave<U, 1>, p, sets, s, 5);
                    float t2 = time_graph(launch_rowwave<U, 2>, p, sets, s, 5);
                    float t3 = time_graph(launch_rowwave<U, 3>, p, sets, s, 5);
                    printf("  rowwave U8 S%-3d wgs %5d | full %6.2f us %5.0f GB/s err %.1e | noatomic %6.2f | loadsonly %6.2f | nooutput %6.2f\n", S, p.ntile * S, t0,
                           bytes / t0 / 1e3, e, t1, t2, t3);
                } while (0);
                do {
                    constexpr int U = 4;
                    if (rows % (4 * U)) break;
                    P p = base; p.nchunk = rows / (4 * U); p.S = S;
                    if (S > p.nchunk || S > SPLITK_MAX_SINGLE) break;
                    double e = check(launch_rowwave<U, 0>, p);
                    float t0 = time_graph(launch_rowwave<U, 0>, p, sets, s, 5);
                    float t1 = time_graph(launch_rowwave<U, 1>, p, sets, s, 5);
                    float t2 = time_graph(launch_rowwave<U, 2>, p, sets, s, 5);
                    float t3 = time_graph(launch_rowwave<U, 3>, p, sets, s, 5);
                    printf("  rowwave U4 S%-3d wgs %5d | full %6.2f us %5.0f GB/s err %.1e | noatomic %6.2f | loadsonly %6.2f | nooutput %6.2f\n", S, p.ntile * S, t0,
                           bytes / t0 / 1e3, e, t1, t2, t3);
                } while (0);
            }
        }
        if (N % 256 == 0 && rows % 32 == 0) {
            constexpr int U = 8;
            P p = base; p.nchunk = rows / 32; p.S = p.nchunk < 32 ? p.nchunk : 32;
            for (int nch = 1; nch <= 4; nch++) {
                float t = time_graph_chains(launch_rowwave<U, 0>, p, sets, s, 5, nch);
                printf("  rowwave U8 S%d on %d parallel graph chain(s): %6.2f us per launch  %5.0f GB/s\n", p.S, nch, t, bytes / t / 1e3);
            }
        }
        if (N % 16 == 0 && rows % 16 == 0 && getenv("LAB_COLSTRIPE")) {
            P p = base;
            // reference on the logical view of set 1
            uint32_t *logical; CK(hipMalloc(&logical, qw_n * 4));
            const WSet &w1 = sets[1 % sets.size()];
            hipLaunchKernelGGL(unrepack_kernel, dim3((unsigned)((qw_n + 255) / 256)), dim3(256), 0, s, w1.qw, logical, rows, N);
            hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256), dim3(256), 0, s, x, logical, w1.sc, w1.qz, yref, K, N);
            CK(hipMemsetAsync(y, 0, N * 2, s));
            { P q = p; q.qw = w1.qw; q.sc = w1.sc; q.qz = w1.qz; launch_colstripe(q, s); }
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(href.data(), yref, N * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hy.data(), y, N * 2, hipMemcpyDeviceToHost));
            double mx = 0, err = 0;
            for (int n = 0; n < N; n++) { mx = fmax(mx, fabs(href[n])); err = fmax(err, fabs(href[n] - (double)(float)hy[n])); }
            const float t = time_graph(launch_colstripe, p, sets, s, 5);
            printf("  colstripe (no K split, repacked [wg][row][16]) wgs %5d: %6.2f us %5.0f GB/s err %.1e\n", N / 16, t, bytes / t / 1e3, err / mx);
            CK(hipFree(logical));
        }
        if (N % 256 == 0 && rows % 32 == 0 && getenv("LAB_PREFETCH")) {
            constexpr int U = 8;
            P p = base; p.nchunk = rows / 32; p.S = p.nchunk < 16 ? p.nchunk : 16;
            if (p.ntile * p.S > 1024) p.S = 1024 / p.ntile;
            const float t0 = time_graph(launch_rowwave<U, 0>, p, sets, s, 5);
            printf("  rowwave U8 S%d: plain chain %6.2f us  %5.0f GB/s\n", p.S, t0, bytes / t0 / 1e3);
            for (int limit : {1, 2, 100}) {
                const float t = time_graph_prefetch<U>(launch_rowwave<U, 0>, p, sets, s, 5, limit);
                printf("  rowwave U8 S%d + L2 prefetch branch (limit %3d chunks/slice): %6.2f us  %5.0f GB/s\n", p.S, limit, t, bytes / t / 1e3);
            }
        }
        if (N % 256 == 0 && rows % 32 == 0) {
            constexpr int U = 8;
            P p = base; p.nchunk = rows / 32; p.S = (rows / 32 >= 32 && N <= 4096) ? 32 : 16;
            const int nwg = p.ntile * p.S;
            u64_t *dbg; CK(hipMalloc(&dbg, (size_t)nwg * 4 * 10 * 8));
            p.dbg = dbg;
            // a few back-to-back launches on cold sets; the stamps of the LAST one are analysed
            const size_t tstride = getenv("LAB_TOUCH") ? (size_t)atol(getenv("LAB_TOUCH")) : 0;
            for (int i = 0; i < 6; i++) { P q = p; const WSet &ws_ = sets[i % sets.size()]; q.qw = ws_.qw;
                if (tstride) hipLaunchKernelGGL(touch_kernel, dim3(getenv("LAB_TOUCH_WGS") ? atoi(getenv("LAB_TOUCH_WGS")) : 8), dim3(256), 0, s, ws_.qw, qw_n * 4, tstride, part); q.sc = ws_.sc; q.qz = ws_.qz; launch_rowwave<U, 4>(q, s); }
            CK(hipStreamSynchronize(s));
            std::vector<u64_t> h((size_t)nwg * 4 * 10);
            CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            const int nw = nwg * 4;
            u64_t r0 = ~0ull, r1 = 0;
            for (int i = 0; i < nw; i++) { r0 = std::min(r0, h[i * 10 + 0]); r1 = std::max(r1, h[i * 10 + 8]); }
            printf("  timeline rowwave U8 S%d (%d waves): kernel span by s_memrealtime %.2f us (100 MHz ticks)\n", p.S, nw, (r1 - r0) / 100.0);
            const char *nm[] = {"start(real,us)", "issued", "x arrived", "first w", "last w", "math done", "atomic back", "end(real,us)"};
            for (int k = 0; k < 8; k++) {
                std::vector<double> v(nw);
                for (int i = 0; i < nw; i++) {
                    const u64_t *d = &h[i * 10];
                    if (k == 0) v[i] = (d[0] - r0) / 100.0;
                    else if (k == 7) v[i] = (d[8] - r0) / 100.0;
                    else v[i] = (double)(d[k + 1] - d[1]);  // shader cycles since the wave's own start
                }
                std::sort(v.begin(), v.end());
                printf("    %-16s min %8.2f  p10 %8.2f  p50 %8.2f  p90 %8.2f  max %8.2f %s\n", nm[k], v[0], v[nw / 10], v[nw / 2], v[nw * 9 / 10], v[nw - 1], (k == 0 || k == 7) ? "us" : "cycles");
            }
            CK(hipFree(dbg));
        }
        if (N % 128 == 0 && false) {
            P p = base;
            auto run_stripe = [&](auto ni) {
                constexpr int NI = decltype(ni)::value;
                CK(hipFuncSetAttribute((const void *)k_stripe<NI, 0, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                CK(hipFuncSetAttribute((const void *)k_stripe<NI, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                CK(hipFuncSetAttribute((const void *)k_stripe<NI, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                double e = check(launch_stripe<NI, 0, true, false>, p);
                float a0 = time_graph(launch_stripe<NI, 0, true, false>, p, sets, s, 5);
                float a1 = time_graph(launch_stripe<NI, 0, true, true>, p, sets, s, 5);
                float a2 = time_graph(launch_stripe<NI, 0, false, true>, p, sets, s, 5);
                float l0 = time_graph(launch_stripe<NI, 2, true, false>, p, sets, s, 5);
                float l1 = time_graph(launch_stripe<NI, 2, true, true>, p, sets, s, 5);
                float l2 = time_graph(launch_stripe<NI, 2, false, true>, p, sets, s, 5);
                float l3 = time_graph(launch_stripe<NI, 2, false, false>, p, sets, s, 5);
                printf("  stripe64 NI%d wgs %5d | full remap/plain %6.2f us %5.0f GB/s err %.1e | remap/nt %6.2f | noremap/nt %6.2f | loadsonly: remap/plain %6.2f remap/nt %6.2f noremap/nt %6.2f noremap/plain %6.2f\n",
                       NI, N / 16, a0, bytes / a0 / 1e3, e, a1, a2, l0, l1, l2, l3);
            };
            if (rows <= 512) run_stripe(std::integral_constant<int, 2>{});
            else if (rows <= 1536) run_stripe(std::integral_constant<int, 6>{});
        }
        if (big) CK(hipFree(arena)); else for (auto &w : sets) { CK(hipFree(w.qw)); CK(hipFree(w.sc)); CK(hipFree(w.qz)); }
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(yref));
    }
    return 0;
}
