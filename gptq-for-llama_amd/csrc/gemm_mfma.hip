// gemm_mfma.hip -- MFMA-tiled dequant-GEMM for batched prefill (M > 64) on gfx950:
// C[M,N] = A[M,K] . deq(B) (+ bias), the large-M regime of the reference's matmul_248_kernel
// (quant/quant_linear.py:72-137).  Bound: fp16 MFMA (2.5 PFLOP/s dense); flops = 2*M*N*K.
//
//  * workgroup tile 256(M) x 256(N) x 64(K), 8 waves in a 2 x 4 grid, each wave owns 128 x 64 =
//    4 x 2 tiles of v_mfma_f32_32x32x16_f16 (128 fp32 accumulators per lane): TWO waves per SIMD, so
//    one wave's LDS reads / dequant VALU / barrier waits sit under the other's MFMAs without
//    hand scheduling (measured: 4 waves x 128x128 reached 0.8 PF);
//  * B is dequantised ONCE per (workgroup, K slab) on the way into LDS, with the reference's own
//    numerics -- fp16(q - z) exact (magic-exponent unpack + packed fp16 subtract), times the fp16
//    scale, one fp16 rounding (quant_linear.py:128) -- so the MFMA consumes exactly the weights
//    the reference's tl.dot does; a packed word (8 consecutive k of one column) IS one B fragment
//    (8 halves of one column per lane), so the LDS layout is [n][k8-block] rows of 16-byte
//    fragments (row stride 144 B: conflict-free ds_write_b128 and ds_read_b128);
//  * A goes global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write): a wave
//    instruction lands 8 rows x 128 B contiguously; rows are unpadded and the 16-byte chunks of a row
//    are XOR-swizzled (slot = k8-block ^ ((row >> 1) & 7)) on the GLOBAL side, which makes the
//    ds_read_b128 of 32 consecutive rows conflict-free; LDS is double buffered;
//  * operands are swapped in the MFMA (D = Bfrag^T-major) so that each lane ends with 4
//    consecutive n of one m: the epilogue transposes through LDS and writes 16 B per lane,
//    256 contiguous bytes per row;
//  * workgroups are numbered so that all N tiles of an M tile run on the same XCD (block b lands
//    on XCD b % 8): the A slab is fetched from HBM once and shared through that XCD's L2.
#include <atomic>

#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

std::atomic<int> g_gemm_version{2};   // 2 = ping-pong kernel (default), 3 = all-LDS-DMA kernel with packed B in LDS
std::atomic<int> g_gemm_diag{0};      // timing experiments: see gemm_mfma_v3_kernel<DIAG>

struct GemmParams {
    const half_t *a;
    int64_t lda;
    const uint32_t *qw;
    const half_t *sc;
    const int32_t *qz;
    const half_t *bias;
    // PAIR mode (fused gate/up, reference fused_mlp.py:84-168): the second weight set.  A workgroup tile is then 256 rows x 128 output
    // columns; each wave's 64 tile columns are 32 gate columns + the SAME 32 columns of up, so both fp32 sums of an output sit in
    // one lane and SiLU is applied to the accumulator (fused_mlp.py:160-165) -- no rounded intermediate, no second launch.
    const uint32_t *qw1;
    const half_t *sc1;
    const int32_t *qz1;
    half_t *c;
    int64_t ldc;
    int M, K, N, groupsize;
    int gshift;    // log2(groupsize) or -1
    int ntm, ntn;  // tiles along M and N
    u64_t *dbg;    // development: per-wave phase stamps of slab 8 ([block][wave][8]) or nullptr
};

constexpr int GM = 256, GN = 256, GK = 64;
constexpr int KB = GK / 8;                 // k8-blocks per slab
constexpr int ROWB = KB * 16 + 16;         // LDS row stride in bytes (144): 16-B pad kills bank conflicts
constexpr int TILE_BYTES = GM * ROWB;      // one B buffer: 36 864 B
constexpr int AROW = KB * 16;              // A rows are unpadded (LDS-DMA writes 1 KiB contiguously)
constexpr int ATILE_BYTES = GM * AROW;     // one A buffer: 32 768 B
constexpr int NWAVE = 8, WN = 4;            // waves per workgroup, waves along N (2 along M)
constexpr int WTN = GN / WN;               // columns per wave (64)
constexpr int CROW = WTN * 2 + 16;         // epilogue row stride (bytes) of a wave's 128 x 64 fp16 block

// 8 consecutive k (one k8-block) of one column -> 8 halves with the reference's numerics.
//   words : the packed words covering the block (4-bit: 1, 8-bit: 2; 2-bit handled by the caller)
template <int BITS>
GPTQ_DEV half8_t dequant8(const uint32_t *w, half2_t zc, half2_t s2, uint32_t msk, uint32_t mag);

template <>
__device__ __forceinline__ half8_t dequant8<4>(const uint32_t *w, half2_t zc, half2_t s2, uint32_t msk, uint32_t mag) {
    half2_t t[4];
    Unpack<4>::pairs_rc(w[0], t, msk, mag);  // {OFF+q_i, OFF+q_{i+4}}
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = (t[i] - zc) * s2;  // exact subtract, one fp16 rounding
    return half8_t{t[0][0], t[1][0], t[2][0], t[3][0], t[0][1], t[1][1], t[2][1], t[3][1]};
}

template <>
__device__ __forceinline__ half8_t dequant8<8>(const uint32_t *w, half2_t zc, half2_t s2, uint32_t msk, uint32_t mag) {
    half2_t a[2], b[2];
    Unpack<8>::pairs_rc(w[0], a, msk, mag);  // bytes (0,2), (1,3)
    Unpack<8>::pairs_rc(w[1], b, msk, mag);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        a[i] = (a[i] - zc) * s2;
        b[i] = (b[i] - zc) * s2;
    }
    return half8_t{a[0][0], a[1][0], a[0][1], a[1][1], b[0][0], b[1][0], b[0][1], b[1][1]};
}

// fp32 accumulators -> fp16, transposed through LDS (wave-private region), 16-B row stores (+bias).
// TN = 32-column tiles per wave (wave tile = 128 rows x 32*TN columns), wn = the wave's column index.
constexpr int TN_ = WTN / 32;
template <int TN>
GPTQ_DEV void gemm_epilogue_t(const float16_t (&acc)[TN][4], char *smem, int wave, int lane, int wm, int wn, int m0, int n0, int M, int N,
                              const GemmParams &p) {
    constexpr int WCOLS = 32 * TN;         // columns per wave
    constexpr int CR = WCOLS * 2 + 16;     // row stride (bytes) of the wave's 128 x WCOLS fp16 block
    char *cs = smem + wave * (128 * CR);
    const int ml = lane & 31, nq = (lane >> 5) * 4;
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nl = i * 32 + 8 * r + nq;  // D row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
                const half4_t h = {(half_t)acc[i][jj][4 * r + 0], (half_t)acc[i][jj][4 * r + 1], (half_t)acc[i][jj][4 * r + 2],
                                   (half_t)acc[i][jj][4 * r + 3]};
                *(half4_t *)(cs + (jj * 32 + ml) * CR + nl * 2) = h;
            }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the region is private to this wave
    __builtin_amdgcn_wave_barrier();
    constexpr int LPR = WCOLS / 8;         // lanes per output row (16-B pieces)
    constexpr int RPI = 64 / LPR;          // rows per store instruction
    const int c16 = lane % LPR, rsub = lane / LPR;
    const int ncol = n0 + wn * WCOLS + c16 * 8;
#pragma unroll 4
    for (int r = 0; r < 128 / RPI; r++) {
        const int mloc = r * RPI + rsub;
        const int m = m0 + wm * 128 + mloc;
        half8_t v = *(const half8_t *)(cs + mloc * CR + c16 * 16);
        if (m < M && ncol < N) {
            if (p.bias) {
                const half8_t b = *(const half8_t *)(p.bias + ncol);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (half_t)((float)v[e] + (float)b[e]);
            }
            *(half8_t *)(p.c + (size_t)m * p.ldc + ncol) = v;
        }
    }
}
GPTQ_DEV void gemm_epilogue(const float16_t (&acc)[TN_][4], char *smem, int wave, int lane, int wm, int wn, int m0, int n0, int M, int N,
                            const GemmParams &p) {
    gemm_epilogue_t<TN_>(acc, smem, wave, lane, wm, wn, m0, n0, M, N, p);
}

template <int BITS, bool PAIR>
__global__ void __launch_bounds__(512) gemm_mfma_kernel(const GemmParams p) {
    using UP = Unpack<BITS>;
    constexpr int KPW = UP::KPW;
    constexpr int WPB = 8 / KPW;      // words per k8-block (4-bit: 1, 8-bit: 2)
    constexpr int KBT = KB / 2;       // k8-blocks per thread per slab (two threads share a column)
    constexpr int NW = KBT * WPB;     // words per thread per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *As = smem;                       // [2][GM][AROW], swizzled
    char *Bs = smem + 2 * ATILE_BYTES;     // [2][GN][ROWB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: XCD x (= block % 8) walks M tiles x, x+8, ... and all N tiles of each
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tm = (j / p.ntn) * 8 + xcd, tn = j % p.ntn;
    if (tm >= p.ntm) return;
    const int m0 = tm * GM, n0 = tn * (PAIR ? GN / 2 : GN);
    const int M = p.M, N = p.N, K = p.K;

    // ---- A: LDS-DMA map ----------------------------------------------------------------------
    // wave w issues NA instructions per slab; instruction i covers rows 8*(4w+i) .. +7 (1 KiB of LDS);
    // lane l lands at row 8*(4w+i) + l/8, slot l%8 and therefore FETCHES k8-block (l%8) ^ ((row>>1)&7)
    constexpr int NA = GM * KB / 512;
    const half_t *aptr[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int row = 8 * (NA * wave + i) + (lane >> 3);
        const int kb = (lane & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, M - 1);
        aptr[i] = p.a + (size_t)m * p.lda + kb * 8;
    }
    auto dma_a = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < NA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(aptr[i] + k0),
                                             (__attribute__((address_space(3))) void *)(As + buf * ATILE_BYTES + (NA * wave + i) * 1024), 16,
                                             0, 0);
    };
    // B: thread = column n0 + (tid & 255), k8-blocks kb0 .. kb0 + KBT - 1 of the slab.  4-bit: the quad
    // (4 lanes = 4 adjacent columns) fetches its 4 x 4 words as ONE dwordx4 per lane (lane i takes packed
    // row kb0 + i, columns of the whole quad) and transposes in registers (2 DPP stages): a quarter of
    // the VMEM instructions -- the move phase is bound by their issue (tools/gemm_phases.py)
    // PAIR: tile column bcol = 64 wn + c -> weight set c / 32 (0 gate, 1 up), output column n0 + 32 wn + c % 32
    const int bcol = tid & 255, kb0 = (tid >> 8) * KBT;
    const int bset = PAIR ? (bcol >> 5) & 1 : 0;
    const int ocol = PAIR ? (bcol >> 6) * 32 + (bcol & 31) : bcol;
    const uint32_t *qwp = (PAIR && bset) ? p.qw1 : p.qw;
    const half_t *scp = (PAIR && bset) ? p.sc1 : p.sc;
    const int32_t *qzp = (PAIR && bset) ? p.qz1 : p.qz;
    const int nb = min(n0 + ocol, N - 1);
    const uint32_t *bptr = qwp + nb;
    const int nb4 = min(n0 + (ocol & ~3), N - 4);
    const uint32_t *bptr4 = qwp + nb4 + (size_t)(tid & 3) * N;
    const int boff = bcol * ROWB;
    const uint32_t MSK = sreg_const(UP::MSK_C), MAG = vreg_const(UP::MAG_C);
    const int ldz = N / KPW;

    uint32_t breg[NW];
    int g_loaded = -1;
    half_t sreg;      // scale and packed-zero word of the slab's group for this thread's column: prefetched
    uint32_t zreg;    // with the weights so that their L2 latency is not paid inside the move phase
    auto load_global = [&](int k0) {
        if constexpr (BITS == 4) {
            const u32x4 v = *(const u32x4 *)(bptr4 + (size_t)((k0 + kb0 * 8) / KPW) * N);
            breg[0] = v[0]; breg[1] = v[1]; breg[2] = v[2]; breg[3] = v[3];
        } else {
#pragma unroll
            for (int w = 0; w < NW; w++) breg[w] = bptr[(size_t)((k0 + kb0 * 8) / KPW + w) * N];
        }
        const int kfirst = k0 + kb0 * 8;
        const int g = p.gshift >= 0 ? (kfirst >> p.gshift) : (kfirst / p.groupsize);
        if (g != g_loaded) {   // uniform over each half of the workgroup: once per group, not per slab
            g_loaded = g;
            sreg = scp[(size_t)g * N + nb];
            zreg = (uint32_t)qzp[(size_t)g * ldz + nb / KPW];
        }
    };
    // 4 x 4 transpose inside a quad: afterwards lane i holds column i's words of packed rows kb0 .. kb0+3
    auto quad_transpose = [&]() {
        const bool odd = tid & 1, hi = tid & 2;
#pragma unroll
        for (int k = 0; k < 4; k += 2) {   // exchange with lane ^ 1
            const uint32_t send = odd ? breg[k] : breg[k + 1];
            const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            if (odd) breg[k] = recv; else breg[k + 1] = recv;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {      // exchange with lane ^ 2
            const uint32_t send = hi ? breg[k] : breg[k + 2];
            const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
            if (hi) breg[k] = recv; else breg[k + 2] = recv;
        }
    };
    // the thread's KBT k8-blocks (32 consecutive k) lie in ONE group (groupsize % 32 == 0)
    u64_t tm1 = 0, tm2 = 0;
    auto store_lds = [&](int buf) {
        char *bd = Bs + buf * TILE_BYTES;
        if (p.dbg) { __builtin_amdgcn_sched_barrier(0); tm1 = stamp_cycles(0); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (BITS == 4) quad_transpose();
        const float z = (float)(((zreg >> (BITS * (nb % KPW))) & ((1u << BITS) - 1u)) + 1u) + UP::OFF;
        const half2_t s2 = {sreg, sreg}, zc = {(half_t)z, (half_t)z};
#pragma unroll
        for (int kk = 0; kk < KBT; kk++) {
            const half8_t v = dequant8<BITS>(&breg[kk * WPB], zc, s2, MSK, MAG);
            *(half8_t *)(bd + boff + (kb0 + kk) * 16) = v;
        }
        if (p.dbg) { __builtin_amdgcn_sched_barrier(0); tm2 = stamp_cycles(0); __builtin_amdgcn_sched_barrier(0); }
    };

    constexpr int TN = WTN / 32;  // n tiles per wave (2)
    float16_t acc[TN][4];         // [n tile][m tile] (operands swapped: rows of D are n)
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) acc[i][jj] = (float16_t)0.f;

    // ---- main loop: ping-pong between the two wave sets of every SIMD --------------------------
    // Waves 0-3 (set 0) and 4-7 (set 1) share the four SIMDs pairwise.  In every phase one set issues
    // the MFMAs of its K slab while the other moves data for a later slab (wait for its prefetched
    // global loads, dequantise, write LDS, prefetch again); s_barrier separates the phases, so the
    // matrix pipe always has a wave feeding it while the partner's VALU / LDS / VMEM work runs in
    // its shadow (MI355X_MICROARCH.md, "Two waves per SIMD").  Set 1 runs half an iteration late:
    //   set 0:            compute(0) | move(1) | compute(1) | move(2) | ...
    //   set 1:  move(1) | compute(0) | move(2) | compute(1) | ...
    // slab j lives in LDS buffer j & 1; both shares of slab j+1 are stored before anyone computes it.
    const int nslab = K / GK;
    const int set = wave >> 2;
    auto phase_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);  // MFMAs are register-only: without this they drift across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto slab_k = [&](int j) { return min(j, nslab - 1) * GK; };
    const int frow = lane & 31, fkb = lane >> 5;
    auto compute = [&](int j) {
        const int buf = j & 1;
        const char *ab = As + buf * ATILE_BYTES + (wm * 128 + frow) * AROW;
        const char *bb = Bs + buf * TILE_BYTES + (wn * WTN + frow) * ROWB + fkb * 16;
        const int swz = (frow >> 1) & 7;   // rows wm*128 + t*32 + frow: (row >> 1) & 7 depends on frow only
#pragma unroll
        for (int ks = 0; ks < GK / 16; ks++) {
            half8_t af[4], bf[TN];
            const int aslot = ((ks * 2 + fkb) ^ swz) * 16;
#pragma unroll
            for (int t = 0; t < 4; t++) af[t] = *(const half8_t *)(ab + t * 32 * AROW + aslot);
#pragma unroll
            for (int t = 0; t < TN; t++) bf[t] = *(const half8_t *)(bb + t * 32 * ROWB + ks * 32);
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[i], af[jj], acc[i][jj], 0, 0, 0);
        }
    };
    // move(j): B of slab j from the prefetch registers into LDS, then prefetch B of slab j + 1
    auto move_b = [&](int j) {
        __builtin_amdgcn_s_setprio(3);
        store_lds(j & 1);
        load_global(slab_k(j + 1));
        __builtin_amdgcn_s_setprio(0);
    };
    auto wait_vm = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // A of slab m is requested by LDS-DMA one and a half phases before its first reader and awaited
    // (vmcnt) by the issuing wave just before the barrier that publishes it:
    //   set 0:  [dma A(j+1); compute(j)] | [B(j+1) -> LDS; wait; prefetch B(j+2)] | ...
    //   set 1:  [dma A(j+2); B(j+2) -> LDS; prefetch B(j+3)] | [compute(j+1); wait] | ...
    // buffer (j+1)&1 is free as soon as set 1 has finished compute(j-1), i.e. exactly when these
    // phases begin.
    dma_a(0, 0);
    load_global(0);
    store_lds(0);                       // slab 0: every thread stores its B share
    load_global(slab_k(1));
    wait_vm();
    __syncthreads();
    if (set == 0) {
        for (int j = 0; j < nslab; j++) {
            u64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            if (p.dbg && j == 8) t0 = stamp_cycles(0);
            dma_a((j + 1) & 1, slab_k(j + 1));
            compute(j);
            if (p.dbg && j == 8) t1 = stamp_cycles(__builtin_bit_cast(uint32_t, acc[0][0][0]) & 0);
            phase_barrier();
            if (p.dbg && j == 8) t2 = stamp_cycles(0);
            __builtin_amdgcn_s_setprio(3);
            store_lds((j + 1) & 1);
            wait_vm();                   // A(j+1) of this wave has landed (requested a phase and a half ago)
            load_global(slab_k(j + 2));
            __builtin_amdgcn_s_setprio(0);
            if (p.dbg && j == 8) t3 = stamp_cycles(0);
            phase_barrier();
            if (p.dbg && j == 8 && lane == 0 && blockIdx.x < 64) {
                u64_t *d = p.dbg + ((size_t)blockIdx.x * NWAVE + wave) * 8;
                d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = stamp_cycles(0); d[5] = tm1; d[6] = tm2;
            }
        }
        phase_barrier();
    } else {
        dma_a(1, slab_k(1));
        move_b(1);
        phase_barrier();
        for (int j = 0; j < nslab; j++) {
            u64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            if (p.dbg && j == 8) t0 = stamp_cycles(0);
            compute(j);
            wait_vm();                   // A(j+1) share (requested in the previous phase) + B prefetch
            if (p.dbg && j == 8) t1 = stamp_cycles(__builtin_bit_cast(uint32_t, acc[0][0][0]) & 0);
            phase_barrier();
            if (p.dbg && j == 8) t2 = stamp_cycles(0);
            dma_a(j & 1, slab_k(j + 2));
            move_b(j + 2);
            if (p.dbg && j == 8) t3 = stamp_cycles(0);
            phase_barrier();
            if (p.dbg && j == 8 && lane == 0 && blockIdx.x < 64) {
                u64_t *d = p.dbg + ((size_t)blockIdx.x * NWAVE + wave) * 8;
                d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = stamp_cycles(0); d[5] = tm1; d[6] = tm2;
            }
        }
    }
    __syncthreads();

    if constexpr (PAIR) {
        // n tile 0 = gate, n tile 1 = up of the same 32 columns: silu on the fp32 accumulator, then the usual transposing store
        float16_t comb[1][4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float g = acc[0][jj][e];
                comb[0][jj][e] = g * (1.0f / (1.0f + __expf(-g))) * acc[1][jj][e];
            }
        gemm_epilogue_t<1>(comb, smem, wave, lane, wm, wn, m0, n0, M, N, p);
    } else {
        gemm_epilogue(acc, smem, wave, lane, wm, wn, m0, n0, M, N, p);
    }
}


// ---------------------------------------------------------------------------------------
// v3 (4-bit, groupsize % 64 == 0): BOTH operands reach LDS by LDS-DMA and B stays PACKED there.
//  * per K slab a wave issues 4 A instructions (1 KiB each, swizzled rows as above) and ONE B
//    instruction: packed row `wave` of the slab, 256 columns x 4 B = 1 KiB -- 40 KiB per slab instead of
//    64, no VGPR staging, no VALU and no ds_write in the data-movement path at all;
//  * a B fragment (8 consecutive k of one column) is ONE packed word: the wave reads it with
//    ds_read_b32 and dequantises it in registers right before the MFMA (19 VALU per fragment, the
//    reference's fp16 sequence), i.e. in the issue slots the matrix pipe leaves free;
//  * three LDS stages, prefetch distance two, ONE barrier per slab; no phase split: both waves of a
//    SIMD stream MFMAs and the LDS pipe sees 144 KiB of reads + 40 KiB of DMA writes per slab
//    (the v2 ping-pong moved 192 + 64 KiB and was LDS / VMEM-issue bound, tools/gemm_phases.py).
// ---------------------------------------------------------------------------------------
constexpr int V3_STAGES = 3;
constexpr int V3_BBYTES = KB * 1024;                       // packed B of a slab: 8 rows x 256 columns x 4 B
constexpr int V3_STAGE_BYTES = ATILE_BYTES + V3_BBYTES;    // 40 960

// DIAG (timing experiments only, wrong results): 1 = no MFMA, 2 = no LDS reads / dequant (constant
// fragments), 3 = LDS reads but no dequant, 4 = no DMA in the loop
template <int DIAG>
__global__ void __launch_bounds__(512) gemm_mfma_v3_kernel(const GemmParams p) {
    using UP = Unpack<4>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3;
    const int tm = (j0 / p.ntn) * 8 + xcd, tn = j0 % p.ntn;
    if (tm >= p.ntm) return;
    const int m0 = tm * GM, n0 = tn * GN;
    const int M = p.M, N = p.N, K = p.K;
    constexpr int NA = GM * KB / 512;

    const half_t *aptr[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int row = 8 * (NA * wave + i) + (lane >> 3);
        const int kb = (lane & 7) ^ ((row >> 1) & 7);
        aptr[i] = p.a + (size_t)min(m0 + row, M - 1) * p.lda + kb * 8;
    }
    // B: this wave moves packed row `wave` of every slab; lane l -> columns n0 + 4l .. 4l + 3 (clamped)
    const uint32_t *bsrc = p.qw + (size_t)wave * N + min(n0 + 4 * lane, N - 4);
    auto dma = [&](int stage, int k0) {
        char *base = smem + stage * V3_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < NA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(aptr[i] + k0),
                                             (__attribute__((address_space(3))) void *)(base + (NA * wave + i) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bsrc + (size_t)(k0 / 8) * N),
                                         (__attribute__((address_space(3))) void *)(base + ATILE_BYTES + wave * 1024), 16, 0, 0);
    };

    // per-lane scale / zero of the two fragment columns, reloaded when the slab enters a new group
    const int frow = lane & 31, fkb = lane >> 5;
    int colv[TN_];
#pragma unroll
    for (int t = 0; t < TN_; t++) colv[t] = min(n0 + wn * WTN + t * 32 + frow, N - 1);
    const int ldz = N / 8;
    const uint32_t MSK = sreg_const(UP::MSK_C), MAG = vreg_const(UP::MAG_C);
    half2_t s2[TN_], zc[TN_];
    int g_cur = -1;
    auto load_sz = [&](int k0) {
        const int g = p.gshift >= 0 ? (k0 >> p.gshift) : (k0 / p.groupsize);
        if (g != g_cur) {          // uniform
            g_cur = g;
#pragma unroll
            for (int t = 0; t < TN_; t++) {
                const half_t sv = p.sc[(size_t)g * N + colv[t]];
                const uint32_t zw = (uint32_t)p.qz[(size_t)g * ldz + colv[t] / 8];
                const float z = (float)(((zw >> (4 * (colv[t] & 7))) & 15u) + 1u) + UP::OFF;
                s2[t] = half2_t{sv, sv};
                zc[t] = half2_t{(half_t)z, (half_t)z};
            }
        }
    };

    float16_t acc[TN_][4];
#pragma unroll
    for (int i = 0; i < TN_; i++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) acc[i][jj] = (float16_t)0.f;

    const int nslab = K / GK;
    const int swz = (frow >> 1) & 7;
    dma(0, 0);
    dma(1, min(1, nslab - 1) * GK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < nslab; j++) {
        load_sz(j * GK);
        if (DIAG != 4) dma((j + 2) % V3_STAGES, min(j + 2, nslab - 1) * GK);   // stage (j+2)%3 was last read in iteration j-1
        const char *st = smem + (j % V3_STAGES) * V3_STAGE_BYTES;
        const char *ab = st + (wm * 128 + frow) * AROW;
        const char *bw = st + ATILE_BYTES + (wn * WTN + frow) * 4;
        // Software pipeline over the four K16 steps of the slab: the LDS reads and the dequantisation of
        // step ks+1 are issued between the MFMAs of step ks (an in-order wave can only hide ~5 VALU in
        // the 32-cycle issue gap of each MFMA, so they must sit BETWEEN the MFMAs in program order).
        half8_t af[2][4], bf[2][TN_];
        uint32_t wq[TN_];
        auto read_step = [&](int ks, half8_t (&a4)[4]) {
            const int kb = ks * 2 + fkb;
            const int aslot = (kb ^ swz) * 16;
            if constexpr (DIAG == 2) {
#pragma unroll
                for (int t = 0; t < 4; t++) a4[t] = (half8_t)(half_t)(float)(ks + t);
#pragma unroll
                for (int t = 0; t < TN_; t++) wq[t] = (uint32_t)(j + t);
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++) a4[t] = *(const half8_t *)(ab + t * 32 * AROW + aslot);
#pragma unroll
                for (int t = 0; t < TN_; t++) wq[t] = *(const uint32_t *)(bw + kb * 1024 + t * 128);
            }
        };
        auto dequant_step = [&](half8_t (&b2)[TN_]) {
#pragma unroll
            for (int t = 0; t < TN_; t++) {
                if constexpr (DIAG == 2 || DIAG == 3) b2[t] = (half8_t)(half_t)(float)(wq[t] & 3u);
                else b2[t] = dequant8<4>(&wq[t], zc[t], s2[t], MSK, MAG);
            }
        };
        read_step(0, af[0]);
        dequant_step(bf[0]);
#pragma unroll
        for (int ks = 0; ks < GK / 16; ks++) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < GK / 16) read_step(ks + 1, af[nxt]);
            if constexpr (DIAG == 1) {
#pragma unroll
                for (int i = 0; i < TN_; i++)
#pragma unroll
                    for (int jj = 0; jj < 4; jj++)
                        acc[i][jj][0] += (float)bf[cur][i][0] * (float)af[cur][jj][0] + (float)bf[cur][i][7] * (float)af[cur][jj][7];
            } else {
#pragma unroll
                for (int i = 0; i < TN_; i++)
#pragma unroll
                    for (int jj = 0; jj < 4; jj++)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[cur][i], af[cur][jj], acc[i][jj], 0, 0, 0);
            }
            if (ks + 1 < GK / 16) dequant_step(bf[nxt]);
            if (ks + 1 < GK / 16) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + TN_, 0);   // the next step's LDS reads go out first
#pragma unroll
                for (int q = 0; q < 4 * TN_; q++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);     // five VALU of the next step's dequant
                }
            }
        }
        // slab j+1 (requested one iteration ago) must have landed; slab j+2's 5 instructions may stay in flight
        if (DIAG != 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_epilogue(acc, smem, wave, lane, wm, wn, m0, n0, M, N, p);
}


template <int BITS, bool PAIR>
static int launch_gemm(const GemmParams &p, hipStream_t s) {
    auto kern = gemm_mfma_kernel<BITS, PAIR>;
    const size_t lds = 4 * (size_t)TILE_BYTES;  // 147 456 B: epilogue staging NWAVE * 128 * CROW (main loop: 2 A + 2 B buffers = 139 264 B)
    static_assert(NWAVE * 128 * CROW <= 4 * TILE_BYTES, "epilogue staging must fit");
    static LdsOptIn opt_in;   // per instantiation; per device inside
    if (int rc = opt_in.ensure((const void *)kern, lds)) return rc;
    const int groups_of_8 = (p.ntm + 7) / 8;
    dim3 grid(groups_of_8 * 8 * p.ntn), block(512);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
    return (int)hipGetLastError();
}

int gemm_set_version(int v) {
    if (v >= 100) return g_gemm_diag.exchange(v - 100);   // 100 + DIAG: development timing variants of v3
    return g_gemm_version.exchange(v);
}

// Eligibility: bits in {4, 8}, trivial g_idx (checked by the caller), K % 64 == 0, groupsize % 32 == 0,
// N % 8 == 0 (always: N % 32 == 0), rows 16-byte aligned.  Everything else -> GPTQ_E_VARIANT and
// the caller falls back to the weight-streaming kernel.
// pair: q holds two weight sets and y = silu(x W0) * (x W1) (fused gate/up): ONE launch, SiLU on the fp32 accumulators.
int gemm_dispatch(int bits, bool pair, const GemvParams &q, hipStream_t s) {
    // q.dbg: development stamps (gptq_set_debug_buffer)
    if (pair && q.bias) return GPTQ_E_VARIANT;
    if (bits != 4 && bits != 8) return GPTQ_E_VARIANT;
    if (q.K % GK != 0 || q.groupsize % 32 != 0 || q.ldx % 8 != 0 || q.ldy % 8 != 0) return GPTQ_E_VARIANT;
    if (((uintptr_t)q.y % 16) != 0 || (q.bias && ((uintptr_t)q.bias % 16) != 0)) return GPTQ_E_VARIANT;
    GemmParams p;
    p.a = q.x;
    p.lda = q.ldx;
    p.qw = q.qw[0];
    p.sc = q.sc[0];
    p.qz = q.qz[0];
    p.qw1 = pair ? q.qw[1] : nullptr;
    p.sc1 = pair ? q.sc[1] : nullptr;
    p.qz1 = pair ? q.qz[1] : nullptr;
    p.bias = q.bias;
    p.c = q.y;
    p.ldc = q.ldy;
    p.M = q.M;
    p.K = q.K;
    p.N = q.N;
    p.groupsize = q.groupsize;
    p.gshift = -1;
    for (int i = 0; i < 31; i++)
        if ((1 << i) == q.groupsize) p.gshift = i;
    p.dbg = q.dbg;
    p.ntm = (q.M + GM - 1) / GM;
    p.ntn = pair ? (q.N + GN / 2 - 1) / (GN / 2) : (q.N + GN - 1) / GN;
    if (pair) return bits == 4 ? launch_gemm<4, true>(p, s) : launch_gemm<8, true>(p, s);
    if (bits == 4 && q.groupsize % GK == 0 && q.N >= 4 && g_gemm_version.load() == 3) {
        const size_t lds = 4 * (size_t)TILE_BYTES;
        static LdsOptIn opt_in[5];
        {
            int i = 0;
            for (const void *f : {(const void *)gemm_mfma_v3_kernel<0>, (const void *)gemm_mfma_v3_kernel<1>, (const void *)gemm_mfma_v3_kernel<2>,
                                  (const void *)gemm_mfma_v3_kernel<3>, (const void *)gemm_mfma_v3_kernel<4>}) {
                if (int rc = opt_in[i++].ensure(f, lds)) return rc;
            }
        }
        const int groups_of_8 = (p.ntm + 7) / 8;
        dim3 grid(groups_of_8 * 8 * p.ntn), block(512);
        switch (g_gemm_diag.load()) {
            case 1: hipLaunchKernelGGL(gemm_mfma_v3_kernel<1>, grid, block, lds, s, p); break;
            case 2: hipLaunchKernelGGL(gemm_mfma_v3_kernel<2>, grid, block, lds, s, p); break;
            case 3: hipLaunchKernelGGL(gemm_mfma_v3_kernel<3>, grid, block, lds, s, p); break;
            case 4: hipLaunchKernelGGL(gemm_mfma_v3_kernel<4>, grid, block, lds, s, p); break;
            default: hipLaunchKernelGGL(gemm_mfma_v3_kernel<0>, grid, block, lds, s, p); break;
        }
        return (int)hipGetLastError();
    }
    return bits == 4 ? launch_gemm<4, false>(p, s) : launch_gemm<8, false>(p, s);
}

}  // namespace gptq
