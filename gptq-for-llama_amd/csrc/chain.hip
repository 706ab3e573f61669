// chain.hip -- a CHAIN of dependent batch-1 dequant-matvecs in ONE persistent launch.
//
// Why: a decode step is a chain of small matvecs (reference call order: QuantLlamaAttention.forward
// quant/fused_attn.py:117-161 -> qkv_proj, o_proj; QuantLlamaMLP.forward quant/fused_mlp.py:203-218 ->
// gate/up + SiLU, down_proj; TritonLlamaRMSNorm quant/triton_norm.py:50-67 in front of each block).
// Launched one kernel per QuantLinear.forward (quant_linear.py:373-377) every op pays, serially,
// launch -> first-byte latency (1.3 us under load) -> math -> combine round trip -> kernel boundary
// (1.6 us): HBM idles ~3 us per op.  Here one workgroup per CU stays resident for
// the whole chain and the weights of op i+1 -- which do not depend on op i -- are already streaming
// into registers while op i's outputs are being combined, published and re-read.
//
// Workgroup = 8 waves:
//   waves 0-3 "compute": the rowwave GEMV math (gemv.hip): a task = 8 x 1-KiB row segments (8 rows of one
//             weight set or 4 rows of two), CH_D tasks in flight per wave across job AND op boundaries;
//             x comes from LDS (wave-uniform address = broadcast read); per-job partial sums go to an
//             LDS ring slot.
//   waves 4-7 "service": hold no long-latency loads, so their memory operations are never queued behind
//             weight prefetches (vector memory returns in order per wave).  They (a) wait for the previous
//             op (16 striped arrival counters, system-scope polling), (b) stage x of the op into LDS --
//             optionally RMS-normalised with the arithmetic of rms_norm_fwd_fused (triton_norm.py:22-39),
//             (c) per job: sum the 4 waves' partials, ONE returning fixed-point atomic per output
//             (gptq_device.h splitk_add1/2), SiLU*up / residual epilogue, system-scope store of y, and
//             (d) publish "these columns are final" on the op's counters.
// Compute and service waves hand over through LDS counters only (no s_barrier, no compiler fences: a
// workgroup fence would drain the weight prefetches), so neither side waits for the other's memory
// round trips unless a ring slot is genuinely still in use.  Cross-workgroup visibility follows
// MI355X_MICROARCH.md "Valid forms": system-scope stores, vmcnt(0), then the counter atomic; consumers
// poll and read x with system-scope loads.  Every spin is bounded (status word bit 0 = expired).
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

constexpr int CH_D_DEFAULT = 4;    // tasks (8 KiB per wave each) in flight per compute wave (template parameter CH_D)
constexpr int CH_R = 4;            // LDS ring slots for per-job partial sums
constexpr uint32_t CH_SPIN_LIMIT = 1u << 19;

// Explicit address spaces: pointers that come out of the descriptor (or out of a cast of the dynamic LDS block)
// are generic to the compiler, and FLAT accesses count against BOTH vmcnt and lgkmcnt -- every LDS poll would
// then wait for all weights in flight.
#define CH_LDS __attribute__((address_space(3)))
#define CH_GLB __attribute__((address_space(1)))
typedef CH_LDS uint32_t *lds_u32p;

struct ChainLds {
    float red[CH_R][2][4][256];    // per-job partial sums [slot][set][compute wave][column]
    uint32_t arrive[CH_R];         // compute waves that have written the slot (monotonic)
    uint32_t released[CH_R];       // service waves that have consumed the slot (monotonic)
    uint32_t xready;               // service waves that have staged x (monotonic: 4 per op)
    uint32_t depok;                // dependent ops whose producer wave 4 has seen complete (monotonic)
    uint32_t aborted;              // a bounded spin expired: stop waiting anywhere
    uint32_t ssq_ready;            // service waves that have published their sum(x^2) share (monotonic)
    float ssq[2][4];
    uint32_t pad[4];
};
static_assert(sizeof(ChainLds) % 16 == 0, "x buffer alignment");

// LDS hand-over without compiler fences (a fence would also wait for vmcnt(0), i.e. for every weight in flight)
GPTQ_DEV void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
GPTQ_DEV u64_t stamp_cycles_dep(float dep) {  // realtime stamp that cannot be scheduled before dep is computed
    u64_t t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return t;
}
GPTQ_DEV void lds_wait_ge(lds_u32p p, uint32_t target, lds_u32p aborted) {
    uint32_t spins = 0;
    while ((int32_t)(*(volatile CH_LDS uint32_t *)p - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > CH_SPIN_LIMIT) {
            if (*(volatile CH_LDS uint32_t *)aborted) break;
            if (spins > 4 * CH_SPIN_LIMIT) { *(volatile CH_LDS uint32_t *)aborted = 1; break; }
        }
    }
    lds_order();
}
GPTQ_DEV void lds_signal(lds_u32p p) {
    lds_order();
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

constexpr uint32_t TF_VALID = 1, TF_FIRST_OP = 2, TF_FIRST_JOB = 4, TF_LAST_JOB = 8, TF_FUSED = 16, TF_SET1 = 32, TF_NOP = 64;
struct ChainTask {
    uint32_t flags, row, op;
};
struct ChainLoadCtx {  // where the task's weights live (all wave-uniform)
    const CH_GLB uint32_t *qw;
    const CH_GLB half_t *sc;
    const CH_GLB int32_t *qz;
    uint32_t N, tile;
    int gshift;
};

// The per-wave task sequence: ops in order; within an op the jobs wg, wg+G, ... (job = tile + tiles * slice);
// within a job the chunks slice, slice+S, ...; one task per chunk (this wave's rows of the chunk).
struct ChainIter {
    const ChainOpDev *ops;
    int n_ops, o;
    uint32_t wg, G, wave;
    ChainLoadCtx cx;  // cached fields of ops[o] (+ the current tile)
    const CH_GLB uint32_t *qw1;
    const CH_GLB half_t *sc1;
    const CH_GLB int32_t *qz1;
    uint32_t tiles, S, nchunk, jobs, fused;
    uint32_t j, c, slice, set;
    bool fresh_op;

    __device__ __forceinline__ void set_job() {
        if (j < jobs) {
            slice = j / tiles;
            cx.tile = j - slice * tiles;
            c = slice;
            set = 0;
        }
    }
    __device__ __forceinline__ void load_op() {
        const ChainOpDev &d = ops[o];
        cx.qw = (const CH_GLB uint32_t *)d.qw[0]; qw1 = (const CH_GLB uint32_t *)d.qw[1];
        cx.sc = (const CH_GLB half_t *)d.sc[0]; sc1 = (const CH_GLB half_t *)d.sc[1];
        cx.qz = (const CH_GLB int32_t *)d.qz[0]; qz1 = (const CH_GLB int32_t *)d.qz[1];
        cx.N = (uint32_t)d.N; cx.gshift = d.gshift;
        tiles = (uint32_t)d.tiles; S = (uint32_t)d.S; nchunk = (uint32_t)d.nchunk; jobs = (uint32_t)d.jobs;
        fused = d.ns == 2;
        j = wg;
        fresh_op = true;
        set_job();
    }
    __device__ __forceinline__ void init(const ChainOpDev *ops_, int n, uint32_t wg_, uint32_t G_, uint32_t wave_) {
        ops = ops_; n_ops = n; wg = wg_; G = G_; wave = wave_; o = 0;
        if (n_ops > 0) load_op();
    }
    // Produce the next task and the load context it was produced under, then advance.  Straight-line on
    // purpose (no loop): an op in which this workgroup has no job yields ONE no-op task (TF_VALID only with
    // TF_NOP) instead of being skipped in a loop -- loops between the issue and the use of the weight loads
    // make the compiler's s_waitcnt bookkeeping fall back to "wait for everything".
    __device__ __forceinline__ void next(ChainTask &t, ChainLoadCtx &ctx) {
        t.flags = 0; t.row = 0; t.op = (uint32_t)o;
        ctx = cx;
        if (o >= n_ops) { ctx.tile = 0; return; }  // cx still describes the last op: valid addresses
        if (j < jobs) {
            const bool last_set = !fused || set == 1;
            t.flags = TF_VALID | (fresh_op ? TF_FIRST_OP : 0u) | (c == slice && set == 0 ? TF_FIRST_JOB : 0u) |
                      (c + S >= nchunk && last_set ? TF_LAST_JOB : 0u) | (fused ? TF_FUSED : 0u) | (set ? TF_SET1 : 0u);
            t.row = (c * 4 + wave) * 8;
            if (set) { ctx.qw = qw1; ctx.sc = sc1; ctx.qz = qz1; }
            fresh_op = false;
            if (last_set) {
                set = 0;
                c += S;
                if (c >= nchunk) { j += G; set_job(); }
            } else {
                set = 1;
            }
        } else {
            t.flags = TF_VALID | TF_NOP;
            ctx.tile = 0;
        }
        if (j >= jobs) {  // this op is exhausted for this workgroup: the next call starts the next op
            o++;
            if (o < n_ops) load_op();
        }
    }
};

// The weight loads of the compute waves are issued and waited for BY HAND (inline asm): hipcc's s_waitcnt
// insertion does not keep loads in flight across a loop back-edge (it falls back to vmcnt(0) in front of the
// first use -- checked on a 20-line reproducer), which would serialise the whole pipeline.  Rules that make the
// manual count exact: (1) every chain_produce issues exactly CH_LPT loads, also past the end of the chain
// (dummy re-reads of one line of the last op), (2) the compute waves contain no other vector-memory loads,
// (3) chain_wait_task ties the registers to the wait so no use can be scheduled in front of it.
constexpr int CH_LPT = 10;  // loads per task: 8 weight rows + scales + zero word
GPTQ_DEV void ld_nt16(u32x4 &w, const CH_GLB u32x4 *p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(w) : "v"(p)); }
GPTQ_DEV void ld_8(u32x2 &w, const CH_GLB u32x2 *p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(w) : "v"(p)); }
GPTQ_DEV void ld_4(uint32_t &w, const CH_GLB uint32_t *p) { asm volatile("global_load_dword %0, %1, off" : "=&v"(w) : "v"(p)); }
template <int YOUNGER>
GPTQ_DEV void chain_wait_task(u32x4 (&w)[8], u32x2 &s4, uint32_t &zw) {
    static_assert(YOUNGER >= 0 && YOUNGER <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(s4), "+v"(zw)
                 : "n"(YOUNGER)
                 : "memory");
}

template <int BITS>
GPTQ_DEV void chain_produce(ChainIter &it, ChainTask &t, u32x4 (&w)[8], u32x2 &s4, uint32_t &zw, int lane) {
    constexpr int KPW = Unpack<BITS>::KPW;
    ChainLoadCtx c;
    it.next(t, c);
    const bool valid = (t.flags & (TF_VALID | TF_NOP)) == TF_VALID;
    const uint32_t n0 = c.tile * 256 + (uint32_t)lane * 4;
    const uint32_t nc = (valid && n0 < c.N) ? n0 : 0;  // ragged N: idle lanes read column 0, results are dropped
    const uint32_t g = c.gshift >= 0 ? (t.row >> c.gshift) : 0u;
    const uint32_t stride = valid ? c.N : 0u;
    const CH_GLB uint32_t *base = c.qw + (size_t)t.row * c.N + nc;
#pragma unroll
    for (int u = 0; u < 8; u++) ld_nt16(w[u], (const CH_GLB u32x4 *)(base + (size_t)u * stride));
    ld_8(s4, (const CH_GLB u32x2 *)(c.sc + (size_t)g * c.N + nc));
    ld_4(zw, (const CH_GLB uint32_t *)c.qz + (size_t)g * (c.N / KPW) + nc / KPW);
}

// one task = 8 packed rows x this lane's 4 columns of ONE weight set (the rowwave math of gemv.hip)
template <int BITS>
GPTQ_DEV void chain_math(const ChainTask &t, const u32x4 (&w)[8], const u32x2 &s4u, uint32_t zw, float (&yv)[2][4], const CH_LDS half_t *xl, int lane,
                         uint32_t MSK, uint32_t MAG) {
    using UP = Unpack<BITS>;
    constexpr int KPW = UP::KPW, NP = UP::NP;
    const half2_t ones = {(half_t)1.0f, (half_t)1.0f};
    const half4_t s4 = __builtin_bit_cast(half4_t, s4u);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float xs = 0.f;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const u32x4 xv = *(const CH_LDS u32x4 *)(xl + (t.row + u) * KPW);  // same address in every lane: broadcast
        half2_t X[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) {
            X[q] = as_half2(xv[q]);
            xs = __builtin_amdgcn_fdot2(X[q], ones, xs, false);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            half2_t tt[NP];
            UP::pairs_rc(w[u][j], tt, MSK, MAG);
#pragma unroll
            for (int q = 0; q < NP; q++) acc[j] = __builtin_amdgcn_fdot2(tt[q], X[q], acc[j], false);
        }
    }
    float contrib[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float zf = (float)(((zw >> (BITS * ((lane * 4 + j) % KPW))) & ((1u << BITS) - 1u)) + 1u) + UP::OFF;
        contrib[j] = (float)s4[j] * (acc[j] - zf * xs);
    }
    const bool set1 = (t.flags & TF_SET1) != 0;  // value selects, not a pointer select: yv must stay in registers
#pragma unroll
    for (int j = 0; j < 4; j++) {
        yv[0][j] += set1 ? 0.f : contrib[j];
        yv[1][j] += set1 ? contrib[j] : 0.f;
    }
}

// DBG: per (op, workgroup) s_memrealtime stamps [op][wg][16] (tools/chain_check.py --timeline):
//  service wave 4: 0 dependency seen | 1 x staged | 2 first job: partials arrived | 3 first job: combine atomic returned |
//                  9 first job: y stored and acknowledged | 4 last job published
//  compute wave 0: 5 x ready seen | 6 first task's weights landed | 8 first task's math done | 7 last job handed over
template <int BITS, bool DBG, int CH_D>
__global__ void __launch_bounds__(512) chain_kernel(const ChainOpDev *__restrict__ ops, int n_ops, uint32_t *__restrict__ counters,
                                                    u64_t *__restrict__ ws, uint32_t *__restrict__ status, u64_t *__restrict__ dbg) {
    static_assert(BITS == 4, "chain kernel: 4-bit only (the decode headline); other widths use the per-op kernels");
    using UP = Unpack<BITS>;
    constexpr int NP = UP::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
    CH_LDS ChainLds &L = *(CH_LDS ChainLds *)chain_smem;
    CH_LDS half_t *xl = (CH_LDS half_t *)(chain_smem + sizeof(ChainLds));  // x of the current op: KPW halves per packed row, staged (pair) order

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wg = blockIdx.x, G = gridDim.x;

    if (threadIdx.x < CH_R) { L.arrive[threadIdx.x] = 0; L.released[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { L.xready = 0; L.depok = 0; L.aborted = 0; L.ssq_ready = 0; }
    __syncthreads();

    if (wave >= 4) {
        // =============================== service waves ===============================
        const int sw = wave - 4;
        const int t = (int)threadIdx.x - 256;  // column of the tile this thread finishes
        uint32_t depseq = 0, jobseq = 0, normseq = 0;
        for (int o = 0; o < n_ops; o++) {
            const ChainOpDev &op = ops[o];
            const uint32_t jobs = (uint32_t)op.jobs;
            if (wg >= jobs) continue;
            const uint32_t nj = (jobs - wg + G - 1) / G;
            const int K = op.K, N = op.N, S = op.S;
            const bool fused = op.ns == 2;
            // ---- (a) dependency: every column of the previous op is final ----
            if (op.dep_count > 0) {
                depseq++;
                if (sw == 0) {
                    const uint32_t *c = counters + (size_t)(o - 1) * CHAIN_NCNT * CHAIN_CNT_STRIDE + (lane & (CHAIN_NCNT - 1)) * CHAIN_CNT_STRIDE;
                    uint32_t spins = 0;
                    for (;;) {
                        uint32_t v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (lane >= CHAIN_NCNT) v = 0;
#pragma unroll
                        for (int off = 1; off < CHAIN_NCNT; off <<= 1) v += __shfl_xor(v, off, 64);
                        if ((int)__builtin_amdgcn_readfirstlane(v) >= op.dep_count) break;
                        if (++spins > CH_SPIN_LIMIT || *(volatile CH_LDS uint32_t *)&L.aborted) {
                            *(volatile CH_LDS uint32_t *)&L.aborted = 1;
                            if (lane == 0) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    lds_signal(&L.depok);
                } else {
                    lds_wait_ge(&L.depok, depseq, &L.aborted);
                }
            }
            u64_t *dslot = nullptr;
            if constexpr (DBG) dslot = dbg + ((size_t)o * G + wg) * 16;
            if (DBG && sw == 0 && lane == 0) dslot[0] = stamp_realtime();
            // ---- (b) stage x: service thread t takes the 16-byte units t, t+256, ... (one batch of 8 loads) ----
            {
                const int nunits = K / 8;
                float4_t v[8];
                const float *src[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int u = i * 256 + t;
                    src[i] = (const float *)(op.x + (size_t)(u < nunits ? u : nunits - 1) * 8);
                }
                load_sys16_x8(v, src);
                float rstd = 1.f;
                const half_t *nw = op.nw;
                if (nw) {
                    float ss = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (i * 256 + t < nunits) {
                            const half8_t h = __builtin_bit_cast(half8_t, v[i]);
#pragma unroll
                            for (int e = 0; e < 8; e++) ss += (float)h[e] * (float)h[e];
                        }
                    ss = wave_sum_xor(ss, 1);
                    normseq++;
                    if (lane == 0) *(volatile CH_LDS float *)&L.ssq[normseq & 1][sw] = ss;
                    lds_signal(&L.ssq_ready);
                    lds_wait_ge(&L.ssq_ready, 4 * normseq, &L.aborted);
                    const volatile CH_LDS float *q = L.ssq[normseq & 1];
                    rstd = 1.0f / sqrtf((q[0] + q[1] + q[2] + q[3]) / (float)K + op.eps);  // fixed order: reproducible
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int u = i * 256 + t;
                    if (u < nunits) {
                        half8_t h = __builtin_bit_cast(half8_t, v[i]);
                        if (nw) {
                            const half8_t w8 = *(const CH_GLB half8_t *)((const CH_GLB half_t *)nw + (size_t)u * 8);
#pragma unroll
                            for (int e = 0; e < 8; e++) h[e] = (half_t)((float)h[e] * rstd * (float)w8[e]);
                        }
                        half8_t st;  // staged order: dword q = {x_q, x_{q+NP}}
#pragma unroll
                        for (int q = 0; q < NP; q++) { st[2 * q] = h[q]; st[2 * q + 1] = h[q + NP]; }
                        *(CH_LDS half8_t *)(xl + u * 8) = st;
                    }
                }
            }
            lds_signal(&L.xready);
            if (DBG && sw == 0 && lane == 0) dslot[1] = stamp_realtime();
            // ---- (c,d) finish the jobs of this op ----
            u64_t *wsop = ws + (size_t)(o & 1) * CHAIN_MAX_N;
            uint32_t *cnt_op = counters + (size_t)o * CHAIN_NCNT * CHAIN_CNT_STRIDE;
            for (uint32_t r = 0; r < nj; r++) {
                const uint32_t job = wg + r * G;
                const uint32_t tile = job % (uint32_t)op.tiles;
                const uint32_t slot = jobseq % CH_R;
                const int n = (int)tile * 256 + t;
                const bool inb = n < N;
                // residual: requested before anything else, needed last
                unsigned short rbits = 0;
                if (op.resid && inb) rbits = __hip_atomic_load((const CH_GLB unsigned short *)op.resid + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                lds_wait_ge(&L.arrive[slot], 4 * (jobseq / CH_R + 1), &L.aborted);
                if (DBG && r == 0 && sw == 0 && lane == 0) dslot[2] = stamp_realtime();
                float t0 = L.red[slot][0][0][t] + L.red[slot][0][1][t] + L.red[slot][0][2][t] + L.red[slot][0][3][t], t1 = 0.f;
                if (fused) t1 = L.red[slot][1][0][t] + L.red[slot][1][1][t] + L.red[slot][1][2][t] + L.red[slot][1][3][t];
                asm volatile("" : "+v"(t0), "+v"(t1));
                lds_signal(&L.released[slot]);
                jobseq++;
                bool mine = inb;
                if (inb && S > 1) {
                    if (fused) mine = splitk_add2(wsop + n, t0, t1, S, t0, t1);
                    else mine = splitk_add1(wsop + n, t0, S, t0);
                }
                if (DBG && r == 0 && sw == 0 && lane == 0) dslot[3] = stamp_realtime();
                if (mine) {
                    float v = t0;
                    if (fused) v = t0 * (1.0f / (1.0f + __expf(-t0))) * t1;  // silu on the fp32 accumulator
                    half_t h = (half_t)v;
                    if (op.resid) h = (half_t)((float)h + (float)__builtin_bit_cast(half_t, rbits));
                    __hip_atomic_store((CH_GLB unsigned short *)op.y + n, __builtin_bit_cast(unsigned short, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my y stores (and the word reset) are acknowledged
                if (DBG && r == 0 && sw == 0 && lane == 0) dslot[9] = stamp_realtime();
                const uint32_t done = (uint32_t)__builtin_popcountll(__ballot(mine));
                if (lane == 0 && done)
                    __hip_atomic_fetch_add(cnt_op + (job & (CHAIN_NCNT - 1)) * CHAIN_CNT_STRIDE, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (DBG && sw == 0 && lane == 0) dslot[4] = stamp_realtime();
        }
        return;
    }

    // =============================== compute waves ===============================
    ChainIter it;
    it.init(ops, n_ops, wg, G, (uint32_t)wave);
    ChainTask T[CH_D];
    u32x4 W[CH_D][8];
    u32x2 S4[CH_D];
    uint32_t ZW[CH_D];
    const uint32_t MSK = sreg_const(UP::MSK_C), MAG = vreg_const(UP::MAG_C);
    float yv[2][4];
    uint32_t opseq = 0, jobseq = 0;
#pragma unroll
    for (int i = 0; i < CH_D; i++) chain_produce<BITS>(it, T[i], W[i], S4[i], ZW[i], lane);
    for (;;) {
#pragma unroll
        for (int i = 0; i < CH_D; i++) {
            const ChainTask t = T[i];
            if (!(t.flags & TF_VALID)) return;  // tasks are produced in order: nothing valid follows
            chain_wait_task<CH_LPT *(CH_D - 1)>(W[i], S4[i], ZW[i]);  // everything but the CH_D-1 younger tasks has landed
            bool first_of_op = false;
            u64_t t_landed = 0;
            if constexpr (DBG) t_landed = stamp_realtime();
            if (!(t.flags & TF_NOP)) {
            if (t.flags & TF_FIRST_OP) {
                opseq++;
                lds_wait_ge(&L.xready, 4 * opseq, &L.aborted);
                if (DBG && wave == 0 && lane == 0) {
                    dbg[((size_t)t.op * G + wg) * 16 + 5] = stamp_realtime();
                    dbg[((size_t)t.op * G + wg) * 16 + 6] = t_landed;
                }
                first_of_op = true;
            }
            if (t.flags & TF_FIRST_JOB) {
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int j = 0; j < 4; j++) yv[s][j] = 0.f;
            }
            chain_math<BITS>(t, W[i], S4[i], ZW[i], yv, xl, lane, MSK, MAG);
            if (DBG && first_of_op && wave == 0 && lane == 0) dbg[((size_t)t.op * G + wg) * 16 + 8] = stamp_cycles_dep(yv[0][0] + yv[1][0]);
            if (t.flags & TF_LAST_JOB) {
                const uint32_t slot = jobseq % CH_R;
                lds_wait_ge(&L.released[slot], 4 * (jobseq / CH_R), &L.aborted);
                *(CH_LDS float4_t *)&L.red[slot][0][wave][4 * lane] = float4_t{yv[0][0], yv[0][1], yv[0][2], yv[0][3]};
                if (t.flags & TF_FUSED) *(CH_LDS float4_t *)&L.red[slot][1][wave][4 * lane] = float4_t{yv[1][0], yv[1][1], yv[1][2], yv[1][3]};
                lds_signal(&L.arrive[slot]);
                jobseq++;
                if (DBG && wave == 0 && lane == 0) dbg[((size_t)t.op * G + wg) * 16 + 7] = stamp_realtime();
            }
            }
            chain_produce<BITS>(it, T[i], W[i], S4[i], ZW[i], lane);
        }
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static int g_chain_depth = CH_D_DEFAULT;
int chain_set_depth(int d) {
    const int prev = g_chain_depth;
    if (d >= 2 && d <= CH_D_DEFAULT) g_chain_depth = d;
    return prev;
}

int chain_launch(int bits, const ChainOpDev *ops_dev, int n_ops, int max_k, uint32_t *counters, u64_t *ws, uint32_t *status, u64_t *dbg,
                 int nwg, hipStream_t s) {
    if (bits != 4) return GPTQ_E_BITS;
    if (n_ops <= 0 || max_k <= 0 || max_k > CHAIN_MAX_K || nwg <= 0) return GPTQ_E_SHAPE;
    const size_t lds = sizeof(ChainLds) + (size_t)max_k * sizeof(half_t);
    const int depth = g_chain_depth;
    auto kern = dbg ? chain_kernel<4, true, CH_D_DEFAULT>
                    : depth == 2 ? chain_kernel<4, false, 2> : depth == 3 ? chain_kernel<4, false, 3> : chain_kernel<4, false, CH_D_DEFAULT>;
    static size_t configured[8] = {0};
    size_t &conf = configured[(dbg ? 4 : 0) + (depth & 3)];
    if (lds > 48 * 1024 && lds > conf) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        conf = lds;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, s, ops_dev, n_ops, counters, ws, status, dbg);
    return (int)hipGetLastError();
}

}  // namespace gptq
