// ovlab.hip -- laboratory for the OVERLAPPED batch-1 decode chain (round 3; development tool, not shipped).
//
// Question: one LLaMA-7B decode pass is 128 dependent matvec launches; each pays ~1.35 us of launch boundary + first-byte latency
// before its stream runs at ~6.5 TB/s.  The WEIGHTS of op i+1 do not depend on op i -- only x does.  Can op i+1 be
// co-resident with op i (a second hardware queue), request its weights at once, and wait for op i through DEVICE flags instead
// of a kernel boundary?
//
// Protocol (MI355X_MICROARCH.md "Valid forms": write-through payload + drained flag, sc1 loads on the consumer, no fences):
//   producer workgroup: y stored with 8-byte agent-scope (sc1) stores by wave 0 -> s_waitcnt vmcnt(0) -> lane 0 stores
//                       flags[op][blockIdx.x] = 1 (sc1).  One flag word per producer workgroup: no atomics, no fan-in chain.
//   consumer workgroup: requests the weight blocks of its first stripe, THEN wave 0 polls the producer's flag words
//                       (64 lanes x 16-byte sc1 loads = 256 flags per instruction, s_sleep between polls, bounded), barrier,
//                       every wave loads the x of its own row blocks with sc1 loads (L1 is never consulted), private LDS staging.
//   flags are zeroed by ONE memset node at the head of every pass; op i runs on stream i % S; the streams fork after the memset and
//   join at the end of the pass, there is NO event between them in between.
// Deadlock freedom: op i+1 is enqueued behind op i+1-S on its stream, which has completed only after op i-S+1 ... so at most S ops
// are in flight; grids are capped so that S of them fit the chip at the measured occupancy; every spin is bounded (err word).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -o tools/ovlab tools/ovlab.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include <type_traits>

#include "../gptq-for-llama_amd/csrc/gptq_device.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return a;
}
__global__ void fill_u32(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = hash32((uint32_t)i * 2654435761u + seed);
}
// table entry: half2 {scale, 64 + zero + 1}; mult balances the gain of a chained pass (values stay finite over 128 ops)
__global__ void fill_tab(uint32_t *p, size_t n, uint32_t seed, float mult) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t h = hash32((uint32_t)i + seed);
        const half_t s = (half_t)(mult * (0.001f + 0.01f * (h >> 8) * (1.0f / 16777216.0f)));
        const half_t z = (half_t)(65.0f + (float)(h & 15u));
        half2_t e = {s, z};
        p[i] = as_u32(e);
    }
}
__global__ void fill_x(half_t *p, size_t n, uint32_t seed, float mult) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < 12; j++) s += (hash32((uint32_t)(i * 12 + j) + seed) >> 8) * (1.0f / 16777216.0f);
        p[i] = (half_t)(mult * (s - 6.0f));
    }
}

template <int CTRL>
GPTQ_DEV float dpp_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
GPTQ_DEV float fold_rows(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s16 = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    const uint32_t u2 = __builtin_bit_cast(uint32_t, s16);
    auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
    return __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
}
GPTQ_DEV u32x4 load_sc1_b128(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, /*aux: sc1*/ 16));
}

constexpr int FLAG_STRIDE = 1024;       // flag words reserved per op (>= the largest grid, multiple of 4)
constexpr unsigned SPIN_LIMIT = 4000;    // polls before a consumer gives up (err word set): a few ms

struct OP {
    const half_t *x;        // [K]; written by the producer op (or static)
    const uint32_t *R;      // stripe16 image [N/16][nrb][NS][64][4]
    const uint32_t *tab;    // [N/16][NS][G][16] half2 {s, 64 + z + 1}
    half_t *y;              // [N]
    half_t *ylog;           // LOG: a private copy of y per op
    const uint32_t *wait;   // the producer's flag words (nullptr: stream order only)
    uint32_t *sig;          // this op's flag words [gridDim.x]
    uint32_t *err;
    u64_t *dbg;             // [gridDim.x][8] stamps of wave 0 (nullptr: none)
    int nwait, K, N, nrb, G, opidx;
};

template <int NS>
__global__ void ref_kernel(const OP p, double *y) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= p.N) return;
    const int stripe = n / 16, col = n % 16;
    double acc[NS];
    for (int s = 0; s < NS; s++) {
        acc[s] = 0;
        for (int rb = 0; rb < p.nrb; rb++) {
            const half2_t e = as_half2(p.tab[(((size_t)stripe * NS + s) * p.G + rb) * 16 + col]);
            for (int rq = 0; rq < 4; rq++)
                for (int j = 0; j < 4; j++) {
                    const int row = rb * 16 + rq * 4 + j;
                    const uint32_t w = p.R[((((size_t)stripe * p.nrb + rb) * NS + s) * 64 + rq * 16 + col) * 4 + j];
                    for (int i = 0; i < 4; i++) {
                        const int q0 = (w >> (4 * i)) & 15, q1 = (w >> (4 * (i + 4))) & 15;
                        // nibble order of the stripe16 image: (k0 k2 k4 k6 | k1 k3 k5 k7); natural pairs (2i, 2i+1)
                        acc[s] += (double)(float)p.x[row * 8 + 2 * i] * ((double)q0 + 64.0 - (double)(float)e[1]) * (double)(float)e[0];
                        acc[s] += (double)(float)p.x[row * 8 + 2 * i + 1] * ((double)q1 + 64.0 - (double)(float)e[1]) * (double)(float)e[0];
                    }
                }
        }
    }
    if (NS == 2) y[n] = acc[0] / (1.0 + exp(-acc[0])) * acc[1];
    else y[n] = acc[0];
}

// One decode matvec of the chain.  Persistent stripes (grid <= N / 16), private per-wave x staging (no staging barrier),
// 1-shift unpack + v_mfma_f32_4x4x4 dot products (= the product kernel's arithmetic, csrc/stripe_kernel.inc).
// PREALL: ALL row blocks of the first stripe are requested before the wait (they land while the producer finishes);
// otherwise DU blocks, the rest one ahead of the math as in the product kernel.
template <int NU, int NS, int DU, bool PREALL, bool LOG>
__global__ void __launch_bounds__(512) k_chain(const OP p) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XI = (NU + 3) / 4;                 // x instructions per lane (four row blocks each)
    constexpr int WB = XI * 4 * (256 + 32);          // private bytes per wave: per row block 256 B of x + 4 float2 sums
    constexpr int D0 = DU < NU ? DU : NU;
    constexpr int P0 = PREALL ? NU : D0;
    const int nrb = p.nrb, G = p.G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *mine = smem + wave * WB;
    float *red = (float *)(smem + NW * WB);
    const half2_t ones = {(half_t)1.f, (half_t)1.f};
    const uint32_t MSK = sreg_const(0x00F000F0u), MAG = vreg_const(0x54005400u);
    const uint32_t MSK0 = sreg_const(0x000F000Fu), MAG0 = vreg_const(0x64006400u);
    const half2_t c1024 = {(half_t)1024.f, (half_t)1024.f}, c64 = {(half_t)64.f, (half_t)64.f};
    const int rq = lane >> 4, col = lane & 15;
    u64_t st[5] = {0, 0, 0, 0, 0};
    if (p.dbg) st[0] = stamp_realtime();

    // ---- 1. the weights of the first stripe: no dependency on the producer ----
    u32x4 w[NU][NS];
    uint32_t tw[NU][NS];
    const uint32_t *wbase = nullptr, *tbase = nullptr;
    auto set_stripe = [&](int stripe) {
        wbase = p.R + ((size_t)stripe * nrb * NS * 64 + lane) * 4;
        tbase = p.tab + (size_t)stripe * NS * G * 16 + col;
    };
    auto issue = [&](int u) {
        const int rb = min(wave + NW * u, nrb - 1);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            w[u][s] = __builtin_nontemporal_load((const u32x4 *)(wbase + ((size_t)rb * NS + s) * 256));
            tw[u][s] = tbase[((size_t)s * G + rb) * 16];
        }
    };
    const int nstripes = p.N / 16;
    int stripe = blockIdx.x;
    set_stripe(stripe);
#pragma unroll
    for (int u = 0; u < P0; u++) issue(u);
    __builtin_amdgcn_sched_barrier(0);
    if (p.dbg) st[1] = stamp_realtime();

    // ---- 2. wait for the producer op: wave 0 polls its flag words ----
    if (p.wait) {
        if (wave == 0) {
            const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)p.wait, 0, FLAG_STRIDE * 4, 0x00020000);
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                for (int c = 0; c * 256 < p.nwait; c++) {
                    const int b = c * 256 + lane * 4;
                    const u32x4 f = load_sc1_b128(frs, b * 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) ok &= (b + j >= p.nwait) | (f[j] == 1u);
                }
                if (__all(ok)) break;
                if (++spins > SPIN_LIMIT) {
                    if (lane == 0) *p.err = 1u + (uint32_t)p.opidx;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
    }
    if (p.dbg) st[2] = stamp_realtime();

    // ---- 3. x of this wave's row blocks: sc1 loads (the producer stored write-through; L1 is bypassed), private staging ----
    {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, p.K * 2, 0x00020000);
        u32x4 xv[XI];
#pragma unroll
        for (int i = 0; i < XI; i++) {
            const int u = 4 * i + (lane >> 4);
            const int rb = min(wave + NW * u, nrb - 1);
            xv[i] = load_sc1_b128(xrs, (rb * 128 + (lane & 15) * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < XI; i++) {
            float s8 = 0.f, o8 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) s8 = __builtin_amdgcn_fdot2(as_half2(xv[i][q]), ones, s8, false);
            o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][0]), c1024, o8, false);
            o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][1]), c64, o8, false);
            o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][2]), c1024, o8, false);
            o8 = __builtin_amdgcn_fdot2(as_half2(xv[i][3]), c64, o8, false);
            s8 += dpp_quad<0xB1>(s8);
            s8 += dpp_quad<0x4E>(s8);
            o8 += dpp_quad<0xB1>(o8);
            o8 += dpp_quad<0x4E>(o8);
            const int u = 4 * i + (lane >> 4), pc = lane & 15;
            *(u32x4 *)(mine + u * 288 + pc * 16) = xv[i];
            if ((pc & 3) == 0) *(float2 *)(mine + u * 288 + 256 + (pc >> 2) * 8) = float2{o8, s8};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (p.dbg) st[3] = stamp_realtime();

    // ---- 4. stripes ----
    auto body = [&](auto pre_tag) {
        constexpr int PI = decltype(pre_tag)::value;   // row blocks of this stripe already requested
        float y[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) y[s] = 0.f;
#pragma unroll
        for (int u = 0; u < NU; u++) {
            if (u + DU >= PI && u + DU < NU) {
                issue(u + DU);
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool valid = wave + NW * u < nrb;
            const u32x4 *xp = (const u32x4 *)(mine + u * 288 + rq * 64);
            u32x4 X[4];
#pragma unroll
            for (int j = 0; j < 4; j++) X[j] = xp[j];
            const float2 xs = *(const float2 *)(mine + u * 288 + 256 + rq * 8);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                float4_t accv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t ww = w[u][s][j];
                    const uint32_t hi = ww >> 8;
                    const uint32_t t0 = (ww & MSK0) | MAG0, t1 = (ww & MSK) | MAG, t2 = (hi & MSK0) | MAG0, t3 = (hi & MSK) | MAG;
                    const h4_t B1 = __builtin_bit_cast(h4_t, u32x2{t0, t1}), B2 = __builtin_bit_cast(h4_t, u32x2{t2, t3});
                    const h4_t A1 = __builtin_bit_cast(h4_t, u32x2{X[j][0], X[j][1]}), A2 = __builtin_bit_cast(h4_t, u32x2{X[j][2], X[j][3]});
                    accv = __builtin_amdgcn_mfma_f32_4x4x4f16(A1, B1, accv, 0, 0, 0);
                    accv = __builtin_amdgcn_mfma_f32_4x4x4f16(A2, B2, accv, 0, 0, 0);
                }
                const half2_t e = as_half2(tw[u][s]);
                const float sc = valid ? (float)e[0] : 0.f;
                const float zs = -((float)e[1] - 64.f) * sc;
                y[s] = fmaf(sc, accv[0] - xs.x, y[s]);
                y[s] = fmaf(zs, xs.y, y[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; s++) y[s] = fold_rows(y[s]);
        if (lane < 16) {
#pragma unroll
            for (int s = 0; s < NS; s++) red[(wave * NS + s) * 16 + lane] = y[s];
        }
        // the next stripe's first blocks go out before the reduce
        const int nxt = stripe + (int)gridDim.x;
        if (nxt < nstripes) {
            set_stripe(nxt);
#pragma unroll
            for (int u = 0; u < D0; u++) issue(u);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (tid < 4) {
            half_t h[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float a[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    a[s] = 0.f;
#pragma unroll
                    for (int wv = 0; wv < NW; wv++) a[s] += red[(wv * NS + s) * 16 + tid * 4 + c];
                }
                float v = a[0];
                if constexpr (NS == 2) v = a[0] * (1.0f / (1.0f + __expf(-a[0]))) * a[1];
                h[c] = (half_t)v;
            }
            const half2_t lo = {h[0], h[1]}, hi = {h[2], h[3]};
            const u64_t pk = (u64_t)as_u32(lo) | ((u64_t)as_u32(hi) << 32);
            __hip_atomic_store((u64_t *)(p.y + (size_t)stripe * 16 + tid * 4), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
            if constexpr (LOG) *(u64_t *)(p.ylog + (size_t)stripe * 16 + tid * 4) = pk;
        }
        __syncthreads();   // red is rewritten by the next stripe
    };
    body(std::integral_constant<int, P0>{});
    for (stripe += (int)gridDim.x; stripe < nstripes; stripe += (int)gridDim.x) body(std::integral_constant<int, D0>{});

    // ---- 5. publish: every y store of this workgroup came from wave 0 ----
    if (wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.sig + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.dbg) {
            st[4] = stamp_realtime();
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 5; k++) p.dbg[(size_t)blockIdx.x * 8 + k] = st[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
struct Shape { int K, N, NS; const char *name; };
static const Shape SHAPES[4] = {{4096, 12288, 1, "qkv"}, {4096, 4096, 1, "o"}, {4096, 11008, 2, "gate/up"}, {11008, 4096, 1, "down"}};

template <int NU, int NS, int DU, bool PREALL, bool LOG>
static void launch_k(const OP &p, int grid, hipStream_t s) {
    constexpr int XI = (NU + 3) / 4;
    hipLaunchKernelGGL((k_chain<NU, NS, DU, PREALL, LOG>), dim3(grid), dim3(512), 8 * XI * 4 * 288 + 8 * NS * 64, s, p);
}
template <bool PREALL, bool LOG>
static void launch_op(int type, const OP &p, int grid, hipStream_t s) {
    switch (type) {
        case 0: case 1: launch_k<4, 1, 2, PREALL, LOG>(p, grid, s); break;
        case 2: launch_k<4, 2, 1, PREALL, LOG>(p, grid, s); break;
        default: launch_k<11, 1, 3, PREALL, LOG>(p, grid, s); break;
    }
}
template <int NU, int NS, int DU, bool PREALL>
static void report_occ(const char *name) {
    constexpr int XI = (NU + 3) / 4;
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_chain<NU, NS, DU, PREALL, false>, 512, 8 * XI * 4 * 288 + 8 * NS * 64));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)k_chain<NU, NS, DU, PREALL, false>));
    printf("  occupancy %-8s preall=%d: %d workgroups of 512 per CU (API), %d VGPR, lds %d B\n", name, (int)PREALL, nb, fa.numRegs, 8 * XI * 4 * 288 + 8 * NS * 64);
}

struct Chain {
    int layers, nops;
    std::vector<OP> ops;
    std::vector<int> type;
    uint32_t *flags, *err;
    half_t *x0, *xi, *yqkv, *yo, *ymlp, *ydown, *ylog;
    double bytes;
};

static Chain build_chain(int layers, bool flow, hipStream_t s) {
    Chain c;
    c.layers = layers; c.nops = 4 * layers;
    c.ops.resize(c.nops); c.type.resize(c.nops);
    CK(hipMalloc(&c.flags, (size_t)c.nops * FLAG_STRIDE * 4));
    CK(hipMalloc(&c.err, 64));
    CK(hipMemset(c.err, 0, 64));
    CK(hipMalloc(&c.x0, 4096 * 2)); CK(hipMalloc(&c.xi, 11008 * 2)); CK(hipMalloc(&c.yqkv, 12288 * 2)); CK(hipMalloc(&c.yo, 4096 * 2));
    CK(hipMalloc(&c.ymlp, 11008 * 2)); CK(hipMalloc(&c.ydown, 4096 * 2));
    CK(hipMalloc(&c.ylog, (size_t)c.nops * 12288 * 2));
    hipLaunchKernelGGL(fill_x, dim3(64), dim3(256), 0, s, c.x0, (size_t)4096, 77u, 1.0f);
    hipLaunchKernelGGL(fill_x, dim3(64), dim3(256), 0, s, c.xi, (size_t)11008, 78u, 0.5f);
    c.bytes = 0;
    // gain balance for the flow chain (values must stay finite and non-degenerate over 128 ops): std(w) = 0.028 mult
    const float mult[4] = {0.3f, 0.3f, 0.3f, 0.5f};   // run 1 used 0.55 / 1.0: the chain overflowed fp16 at op 3
    for (int l = 0; l < layers; l++)
        for (int t = 0; t < 4; t++) {
            const Shape &sh = SHAPES[t];
            const int i = 4 * l + t;
            const int G = sh.K / 128;
            const size_t r_n = (size_t)(sh.K / 8) * sh.N * sh.NS, t_n = (size_t)sh.NS * G * sh.N;
            uint32_t *R, *tab;
            CK(hipMalloc(&R, r_n * 4)); CK(hipMalloc(&tab, t_n * 4));
            hipLaunchKernelGGL(fill_u32, dim3(2048), dim3(256), 0, s, R, r_n, 1000u + i);
            hipLaunchKernelGGL(fill_tab, dim3(256), dim3(256), 0, s, tab, t_n, 2000u + i, flow ? mult[t] : 1.0f);
            OP &p = c.ops[i];
            memset(&p, 0, sizeof(p));
            p.R = R; p.tab = tab; p.K = sh.K; p.N = sh.N; p.nrb = sh.K / 128; p.G = G; p.opidx = i;
            p.err = c.err; p.sig = c.flags + (size_t)i * FLAG_STRIDE;
            p.ylog = c.ylog + (size_t)i * 12288;
            c.type[i] = t;
            switch (t) {
                case 0: p.x = flow ? (l == 0 ? c.x0 : c.ydown) : c.x0; p.y = c.yqkv; break;
                case 1: p.x = flow ? c.yqkv : c.x0; p.y = c.yo; break;
                case 2: p.x = flow ? c.yo : c.x0; p.y = c.ymlp; break;
                default: p.x = flow ? c.ymlp : c.xi; p.y = c.ydown; break;
            }
            c.bytes += sh.NS * ((double)(sh.K / 8) * sh.N * 4 + (double)G * (sh.N / 8) * 4 + (double)G * sh.N * 2) + 2.0 * sh.K + 2.0 * sh.N;
        }
    CK(hipStreamSynchronize(s));
    return c;
}

struct Cfg { int S; bool flags; bool preall; int gridcap; bool log; u64_t *dbg; };

static void enqueue_pass(Chain &c, const Cfg &cfg, hipStream_t *st, hipEvent_t *ev) {
    // st[0] is the main stream; flags zeroed first, then the fork
    CK(hipMemsetAsync(c.flags, 0, (size_t)c.nops * FLAG_STRIDE * 4, st[0]));
    if (cfg.S > 1) {
        CK(hipEventRecord(ev[0], st[0]));
        for (int k = 1; k < cfg.S; k++) CK(hipStreamWaitEvent(st[k], ev[0], 0));
    }
    int prev_grid = 0;
    for (int i = 0; i < c.nops; i++) {
        OP p = c.ops[i];
        const int grid = std::min(p.N / 16, cfg.gridcap);
        if (cfg.flags && i > 0) { p.wait = c.ops[i - 1].sig; p.nwait = prev_grid; }
        if (cfg.dbg) p.dbg = cfg.dbg + (size_t)i * 1024 * 8;
        hipStream_t s = st[i % cfg.S];
        if (cfg.log) { if (cfg.preall) launch_op<true, true>(c.type[i], p, grid, s); else launch_op<false, true>(c.type[i], p, grid, s); }
        else { if (cfg.preall) launch_op<true, false>(c.type[i], p, grid, s); else launch_op<false, false>(c.type[i], p, grid, s); }
        prev_grid = grid;
    }
    for (int k = 1; k < cfg.S; k++) {
        CK(hipEventRecord(ev[k], st[k]));
        CK(hipStreamWaitEvent(st[0], ev[k], 0));
    }
}

static uint32_t read_err(Chain &c) {
    uint32_t e = 0;
    CK(hipMemcpy(&e, c.err, 4, hipMemcpyDeviceToHost));
    return e;
}

// returns us per pass (best of 3 x reps graph replays), or -1 on a protocol timeout
static float time_cfg(Chain &c, const Cfg &cfg, hipStream_t *st, hipEvent_t *ev, bool graph, int reps) {
    CK(hipMemset(c.err, 0, 64));
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    if (graph) {
        CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
        enqueue_pass(c, cfg, st, ev);
        CK(hipStreamEndCapture(st[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    auto once = [&]() { if (graph) CK(hipGraphLaunch(ge, st[0])); else enqueue_pass(c, cfg, st, ev); };
    once();
    CK(hipStreamSynchronize(st[0]));
    float best = -1.f;
    if (read_err(c) == 0) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0, st[0]));
            for (int i = 0; i < reps; i++) once();
            CK(hipEventRecord(e1, st[0])); CK(hipStreamSynchronize(st[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const float us = ms * 1e3f / reps;
            if (best < 0 || us < best) best = us;
        }
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        if (read_err(c) != 0) best = -1.f;
    }
    if (graph) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
    return best;
}

int main(int argc, char **argv) {
    const int layers = getenv("LAB_LAYERS") ? atoi(getenv("LAB_LAYERS")) : 32;
    const int reps = getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 10;
    hipStream_t st[4]; hipEvent_t ev[4];
    for (int k = 0; k < 4; k++) { CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
    printf("ovlab: %d layers x {qkv 4096x12288, o 4096x4096, gate/up 2x4096x11008, down 11008x4096}, 4-bit g128 stripe16, M = 1\n", layers);
    report_occ<4, 1, 2, false>("K4096"); report_occ<4, 1, 2, true>("K4096");
    report_occ<4, 2, 1, false>("gate/up"); report_occ<4, 2, 1, true>("gate/up");
    report_occ<11, 1, 3, false>("down"); report_occ<11, 1, 3, true>("down");

    // ---------------- A. correctness: a REAL data chain (x of op i+1 = y of op i, buffers reused every layer) ----------------
    {
        Chain c = build_chain(layers, true, st[0]);
        const size_t logn = (size_t)c.nops * 12288;
        std::vector<half_t> ref(logn), got(logn);
        Cfg serial{1, false, false, 1 << 20, true, nullptr};
        CK(hipMemset(c.ylog, 0, logn * 2));
        enqueue_pass(c, serial, st, ev);
        CK(hipStreamSynchronize(st[0]));
        CK(hipMemcpy(ref.data(), c.ylog, logn * 2, hipMemcpyDeviceToHost));
        // arithmetic check of the first layer against a double-precision kernel on the inputs the serial pass used
        for (int i = 0; i < 4; i++) {
            OP p = c.ops[i];
            half_t *xin; CK(hipMalloc(&xin, p.K * 2));
            if (i == 0) CK(hipMemcpy(xin, c.x0, p.K * 2, hipMemcpyDeviceToDevice));
            else CK(hipMemcpy(xin, c.ops[i - 1].ylog, p.K * 2, hipMemcpyDeviceToDevice));
            p.x = xin;
            double *yd; CK(hipMalloc(&yd, p.N * 8));
            if (c.type[i] == 2) hipLaunchKernelGGL((ref_kernel<2>), dim3((p.N + 255) / 256), dim3(256), 0, st[0], p, yd);
            else hipLaunchKernelGGL((ref_kernel<1>), dim3((p.N + 255) / 256), dim3(256), 0, st[0], p, yd);
            CK(hipStreamSynchronize(st[0]));
            std::vector<double> h(p.N);
            CK(hipMemcpy(h.data(), yd, p.N * 8, hipMemcpyDeviceToHost));
            double mx = 0, err = 0;
            for (int n = 0; n < p.N; n++) { mx = fmax(mx, fabs(h[n])); err = fmax(err, fabs(h[n] - (double)(float)ref[(size_t)i * 12288 + n])); }
            printf("  check op %d (%s): max|y| %.3f, max err / max|y| %.2e\n", i, SHAPES[c.type[i]].name, mx, err / mx);
            CK(hipFree(xin)); CK(hipFree(yd));
        }
        {
            double mxl = 0; int nonfinite = 0;
            for (int n = 0; n < 4096; n++) { const float v = (float)ref[(size_t)(c.nops - 1) * 12288 + n]; if (!std::isfinite(v)) nonfinite++; else mxl = fmax(mxl, fabs(v)); }
            printf("  flow chain: last op max|y| %.4g, non-finite %d\n", mxl, nonfinite);
        }
        for (int S = 1; S <= 4; S++)
            for (int pre = 0; pre < 2; pre++)
                for (int graph = 0; graph < 2; graph++) {
                    const int cap = S <= 2 ? 512 : (S == 3 ? 320 : 256);
                    Cfg cfg{S, true, pre != 0, cap, true, nullptr};
                    int bad_total = 0, first_bad = -1; uint32_t e = 0;
                    for (int rep = 0; rep < 3 && e == 0; rep++) {
                        CK(hipMemset(c.ylog, 0, logn * 2));
                        CK(hipMemset(c.err, 0, 64));
                        if (graph) {
                            hipGraph_t g; hipGraphExec_t ge;
                            CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
                            enqueue_pass(c, cfg, st, ev);
                            CK(hipStreamEndCapture(st[0], &g));
                            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                            CK(hipGraphLaunch(ge, st[0])); CK(hipGraphLaunch(ge, st[0]));
                            CK(hipStreamSynchronize(st[0]));
                            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
                        } else {
                            enqueue_pass(c, cfg, st, ev); enqueue_pass(c, cfg, st, ev);
                            CK(hipStreamSynchronize(st[0]));
                        }
                        e = read_err(c);
                        CK(hipMemcpy(got.data(), c.ylog, logn * 2, hipMemcpyDeviceToHost));
                        for (int i = 0; i < c.nops; i++) {
                            const int N = c.ops[i].N;
                            int bad = 0;
                            for (int n = 0; n < N; n++) bad += memcmp(&got[(size_t)i * 12288 + n], &ref[(size_t)i * 12288 + n], 2) != 0;
                            if (bad) { bad_total += bad; if (first_bad < 0) first_bad = i; }
                        }
                    }
                    printf("  flow S=%d preall=%d cap=%d %s: err word %u, mismatching outputs %d (first bad op %d)  %s\n", S, pre, cap, graph ? "graph" : "eager", e,
                           bad_total, first_bad, (e == 0 && bad_total == 0) ? "BIT-EXACT vs serial" : "FAILED");
                    fflush(stdout);
                }
        // timing in flow mode
        printf("  -- timing, real data chain --\n");
        for (int graph = 1; graph >= 0; graph--) {
            Cfg base{1, false, false, 1 << 20, false, nullptr};
            const float t0 = time_cfg(c, base, st, ev, graph != 0, reps);
            printf("  %s S=1 no flags (stream order, full grids): %8.1f us/pass  %6.0f GB/s\n", graph ? "graph" : "eager", t0, c.bytes / t0 / 1e3);
            for (int S = 1; S <= 4; S++)
                for (int pre = 0; pre < 2; pre++) {
                    const int caps[3] = {S <= 2 ? 512 : (S == 3 ? 320 : 256), 256, 1 << 20};
                    for (int ci = 0; ci < 3; ci++) {
                        if (ci == 1 && caps[0] == 256) continue;
                        if (ci == 2 && S > 2) continue;      // uncapped grids: two co-resident ops still fit (1024 workgroup slots)
                        Cfg cfg{S, true, pre != 0, caps[ci], false, nullptr};
                        const float t = time_cfg(c, cfg, st, ev, graph != 0, reps);
                        printf("  %s S=%d preall=%d cap=%-7d: %8.1f us/pass  %6.0f GB/s  (%.3f of 8 TB/s)\n", graph ? "graph" : "eager", S, pre, caps[ci], t, c.bytes / t / 1e3,
                               c.bytes / t / 1e3 / 8000.0);
                        fflush(stdout);
                    }
                }
        }
        // per-workgroup stamps of the best-looking configuration (S=2, preall): where does an op's time go?
        {
            u64_t *dbg; CK(hipMalloc(&dbg, (size_t)c.nops * 1024 * 8 * 8));
            for (int S = 1; S <= 2; S++) {
                CK(hipMemset(dbg, 0, (size_t)c.nops * 1024 * 8 * 8));
                Cfg cfg{S, true, true, 512, false, dbg};
                enqueue_pass(c, cfg, st, ev); CK(hipStreamSynchronize(st[0]));
                enqueue_pass(c, cfg, st, ev); CK(hipStreamSynchronize(st[0]));
                std::vector<u64_t> h((size_t)c.nops * 1024 * 8);
                CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
                printf("  -- stamps (eager, S=%d, preall, cap 512; us relative to the op's first workgroup start; 100 MHz clock) --\n", S);
                for (int i = 40; i < 48; i++) {
                    const int grid = std::min(c.ops[i].N / 16, 512);
                    u64_t t0 = ~0ull, tend = 0;
                    for (int b = 0; b < grid; b++) { const u64_t *d = &h[((size_t)i * 1024 + b) * 8]; t0 = std::min(t0, d[0]); tend = std::max(tend, d[4]); }
                    u64_t pend = 0;   // the producer's last flag store
                    { const int pg = std::min(c.ops[i - 1].N / 16, 512); for (int b = 0; b < pg; b++) pend = std::max(pend, h[((size_t)(i - 1) * 1024 + b) * 8 + 4]); }
                    const char *nm[5] = {"start", "weights issued", "flag seen", "x staged", "published"};
                    printf("    op %d %-8s grid %4d: span %.2f us; producer published at %+.2f us\n", i, SHAPES[c.type[i]].name, grid, (tend - t0) / 100.0, ((double)pend - (double)t0) / 100.0);
                    for (int k = 0; k < 5; k++) {
                        std::vector<double> v(grid);
                        for (int b = 0; b < grid; b++) v[b] = ((double)h[((size_t)i * 1024 + b) * 8 + k] - (double)t0) / 100.0;
                        std::sort(v.begin(), v.end());
                        printf("        %-15s min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f\n", nm[k], v[0], v[grid / 2], v[grid * 9 / 10], v[grid - 1]);
                    }
                }
            }
            CK(hipFree(dbg));
        }
    }
    return 0;
}
