// gemm8.hip -- the hand-written prefill GEMM of the product route (round 3): C[M, N] = X[M, K] . Wt[N, K]^T on fp16 operands with
// fp32 accumulation, the large-M regime of the reference's matmul_248_kernel (quant/quant_linear.py:72-137), of its fused MLP
// kernel (quant/fused_mlp.py:84-168: PAIR mode, SiLU on the fp32 accumulators) and, with the roles of K and N exchanged, of
// transpose_matmul_248_kernel (quant_linear.py:191-258).  Bound: fp16 MFMA (2.5 PFLOP/s dense); flops = 2 M N K.
//
// Wt is the layer dequantised ONCE PER CALL into the caller's workspace with the reference's own rounding (fp16(q - z) * fp16
// scale, elementwise.hip dequant_t_kernel) in K-CONTIGUOUS order, so that BOTH operands reach LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR staging, no ds_write, no VALU in the K loop) and a fragment is one ds_read_b128.
// At prefill sizes the packed weight's bytes do not matter (2 M N K flops against K N / 2 bytes); the 10-20 us dequantise pass
// is what the round-2 fused tile kernel (gemm_mfma.hip) paid per workgroup and K slab instead.
//
// Structure (MI355X guide, "256^2 8-phase" schedule, rebuilt for this operand layout):
//  * workgroup tile 256 (m) x 256 (n) x 64 (k), 8 waves = 2 (m) x 4 (n), wave tile 128 x 64 = 8 x 4 v_mfma_f32_16x16x32_f16
//    tiles (128 accumulators per lane), operands swapped (D = W-fragment x X-fragment) so a lane ends with 4 consecutive n of
//    one m: 8-byte stores;
//  * LDS: two K-tile buffers of 64 KB; a buffer is four 16-KB UNITS ordered by CONSUMPTION, not by tile row:
//        X0 = the first 64 rows of each wave row's 128 (quadrants (0, *)), X1 = the other 64, W0 = the first 32 columns of
//        each wave column's 64, W1 = the other 32 (PAIR mode: W0 = gate columns, W1 = the SAME columns of up);
//    rows are 128 B (64 k), the 16-byte chunks of a row XOR-swizzled by (row >> 1) & 7 on the SOURCE address (LDS-DMA writes
//    lane-linear), which makes every ds_read_b128 lane group conflict-free;
//  * a K tile is four phases, each {R: ds_reads of one quadrant's missing fragments + the 2 LDS-DMA instructions of ONE unit
//    of a later tile + a COUNTED vmcnt, barrier, M: 16 MFMAs under s_setprio(1), barrier}:
//        R1 reads X0, W0 (12)  stages W1(t+1)   vmcnt(8)   M (0,0)
//        R2 reads W1 (4)       stages X1(t+1)   vmcnt(8)   M (0,1)
//        R3 reads X1 (8)       stages X0(t+2)              M (1,1)
//        R4 -                  stages W0(t+2)   vmcnt(8)   M (1,0)
//    so every unit is requested 5-6 phases before its first reader and re-staged at least 2 phases after its last one, four
//    units (8 instructions per wave) stay in flight across every barrier, and the wait constant is the same everywhere;
//  * the two wave rows run ONE BARRIER apart (wave row 1 takes an extra s_barrier at the start, row 0 at the end): the
//    waves that share a SIMD (w and w + 4) are always in opposite sections -- one issues MFMAs while the other reads LDS and
//    issues DMA.  Waits sit at the END of an R section, i.e. one barrier before the staggered partner's first read
//    (guide: "one barrier MORE when two wave groups run staggered");
//  * workgroups are numbered so that every XCD runs a contiguous, equally long run of the m-major tile list (X panel shared through its L2; round 5).
#include <atomic>
#include <type_traits>

#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {

namespace {

constexpr int TK = 64;   // (m tile: 4 XH rows = 256, or 192 in the round-5 instance; n tile: 256 columns, 128 in PAIR mode)
constexpr int UNIT_BYTES = 128 * 128;          // 128 rows x 64 k x 2 B
constexpr int BUF_BYTES = 4 * UNIT_BYTES;      // X0 X1 W0 W1
constexpr int OFF_X0 = 0, OFF_X1 = UNIT_BYTES, OFF_W0 = 2 * UNIT_BYTES, OFF_W1 = 3 * UNIT_BYTES;

struct Gemm8Params {
    const half_t *x;     // [M, K], row stride ldx
    const half_t *wt;    // [N (PAIR: 2 N), K], row stride ldw, k contiguous; PAIR: rows 0..N-1 gate, N..2N-1 up
    const half_t *bias;  // [N] or nullptr (plain mode only)
    half_t *c;           // [M, N], row stride ldc
    int64_t ldx, ldw, ldc;
    int M, K, N;
    int ntm, ntn;
    int per;             // tiles per XCD: ceil(ntm ntn / 8)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define G8_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)
#define G8_VMCNT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define G8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// MF32: v_mfma_f32_32x32x16_f16 instead of 16x16x32 (8 instead of 16 MFMAs per phase at the same LDS traffic; the larger shape reaches
// the full 1024 flop / cycle / SIMD, the smaller one ~85-94 % of it) -- same units, phases and waits, other fragment maps.
//
// XH (round 5): rows of an X unit per wave row -- 64 gives the 256-row tile above; 48 gives a 192-row tile (wave tile 96 x 64,
// 12 MFMAs per phase instead of 16, 96 accumulators) for the row counts where 256-row tiles leave the last round of workgroups mostly empty:
// 3072 x 4096 is 12 x 16 = 192 tiles of 256 rows on 256 CUs (a quarter of the chip idle) but 16 x 16 = 256 tiles of 192 rows, one full round of
// 3/4 the work; 3072 x 12288 is 576 tiles = 2.25 rounds (three are paid) against 768 = three rounds of 3/4 the work.  Same units, phases, waits
// and vmcnt bookkeeping: an X unit is 96 rows instead of 128, so the DMA instructions of waves 6 and 7 repeat those of waves 0 and 1 (same
// bytes to the same LDS address -- benign, and every wave keeps the same number of loads in flight).  The K order of every accumulator is
// unchanged: results are bit-identical to the 256-row tile.
template <bool PAIR, bool MF32, int XH>
__global__ void __launch_bounds__(512) gemm8_kernel(const Gemm8Params p) {
    static_assert(XH == 64 || (XH == 48 && !MF32), "X unit half: 64 rows, or 48 (16-row fragments only)");
    constexpr int TM = 4 * XH, XU = 2 * XH;   // rows of the workgroup tile; rows of an X unit (both wave rows)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ALL of the kernel's LDS (one object: see the guide's .s traps)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // XCD-aware tile order.  Workgroup b runs on XCD b % 8; the tiles are numbered m-major (all n tiles of m tile 0, then of m tile 1, ...) and XCD x
    // owns the CONTIGUOUS run [x per, (x + 1) per) of that list, per = ceil(tiles / 8): the workgroups an XCD runs at the same time share one X
    // panel (two at a seam) through its L2, and every XCD gets the same number of tiles.  (Rounds 3-4 gave XCD x the WHOLE m tiles x, x + 8, ...:
    // with ntm % 8 != 0 some XCDs carried twice the tiles of the others -- 2304 x 12288: m tiles 0 and 8 on XCD 0 = 96 tiles = three rounds of its
    // 32 CUs while seven XCDs finished after two; measured 305 us against 229 us for the library, profiles/r5e_gemm8_tile/.)
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile_id = xcd * p.per + jb;
    if (tile_id >= p.ntm * p.ntn) return;
    const int tm = tile_id / p.ntn, tn = tile_id % p.ntn;
    constexpr int NCOLS = PAIR ? 128 : 256;    // output columns of a workgroup tile
    const int m0 = tm * TM, n0 = tn * NCOLS;
    const int M = p.M, N = p.N, K = p.K;

    // ---- LDS-DMA sources.  Instruction q (0, 1) of a unit covers unit rows 16 wave + 8 q + lane / 8; lane lands at chunk
    // position lane % 8 and therefore FETCHES chunk (lane % 8) ^ ((row >> 1) & 7) of that row.
    const half_t *sx[2][2], *sw[2][2];   // [unit half][q]
    int xdst[2];                         // LDS row of this wave's X instruction q (XH = 48: waves 6, 7 repeat waves 0, 1)
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int u = 16 * wave + 8 * q + (lane >> 3);
        const int ch = (lane & 7) ^ ((u >> 1) & 7);
        xdst[q] = (16 * wave + 8 * q) % XU;
        const int ux = u % XU, chx = (lane & 7) ^ ((ux >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // X unit h, unit row ux: wave row ux / XH, local row ux % XH -> tile row (ux / XH) * 2 XH + h * XH + ux % XH
            const int mrow = min(m0 + (ux / XH) * XU + h * XH + (ux % XH), M - 1);
            sx[h][q] = p.x + (size_t)mrow * p.ldx + chx * 8;
            // W unit h, unit row u: wave column u / 32, local column u % 32
            int nrow;
            if constexpr (PAIR) nrow = h * N + min(n0 + (u >> 5) * 32 + (u & 31), N - 1);             // gate | up rows of the stacked matrix
            else nrow = min(n0 + (u >> 5) * 64 + h * 32 + (u & 31), N - 1);
            sw[h][q] = p.wt + (size_t)nrow * p.ldw + ch * 8;
        }
    }
    const int nt = K / TK;
    auto stage_rows = [&](const half_t *(&src)[2], int buf_unit_off, int t, int r0, int r1) {
        const int k0 = min(t, nt - 1) * TK;   // past the end: re-fetch the last tile (keeps the vmcnt bookkeeping uniform)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[0] + k0),
                                         (__attribute__((address_space(3))) void *)(smem + buf_unit_off + r0 * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[1] + k0),
                                         (__attribute__((address_space(3))) void *)(smem + buf_unit_off + r1 * 128), 16, 0, 0);
    };
    auto stage = [&](const half_t *(&src)[2], int buf_unit_off, int t) { stage_rows(src, buf_unit_off, t, 16 * wave, 16 * wave + 8); };   // W units
    auto stage_x = [&](const half_t *(&src)[2], int buf_unit_off, int t) { stage_rows(src, buf_unit_off, t, xdst[0], xdst[1]); };

    // ---- fragment reads.  A 16x16x32 operand: lane l holds row l % 16, k = 8 (l / 16) .. +7 of a 32-k step; both operands
    // come out of LDS the same way.  Chunk of k-step ks: (4 ks + l / 16) ^ swz, swz = ((l % 16) >> 1) & 7 (unit rows start at
    // multiples of 16, so only the low row bits reach the swizzle): the ks = 1 address is the ks = 0 address ^ 64.
    // A 32x32x16 operand (MF32): row l % 32, k = 8 (l / 32) .. +7 of a 16-k step: chunk (2 ks + l / 32) ^ swz, swz = ((l % 32) >> 1) & 7,
    // i.e. the ks address is the ks = 0 address ^ (32 ks).
    constexpr int RL = MF32 ? 32 : 16;                       // rows of a fragment
    const int frow = lane & (RL - 1), fkb = MF32 ? (lane >> 5) : (lane >> 4);
    const int swz = (frow >> 1) & 7;
    const int lb0 = frow * 128 + ((fkb ^ swz) << 4);
    const int lb1 = lb0 ^ 64;
    const int xbase = wr * XH * 128, wbase = wc * 32 * 128;
    constexpr int XF = XH / RL, WF = MF32 ? 1 : 2, KS = MF32 ? 4 : 2;   // m reps of an X unit, n reps of a W unit, k steps of a tile
    half8_t xf[XF][KS], wf[2][WF][KS];   // X fragments of the current m half [ii][ks]; W fragments [nh][jj][ks]
    auto read_x = [&](int buf, int mh) {
        const char *b = smem + buf * BUF_BYTES + (mh ? OFF_X1 : OFF_X0) + xbase;
#pragma unroll
        for (int ii = 0; ii < XF; ii++)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) xf[ii][ks] = *(const half8_t *)(b + ii * (RL * 128) + (MF32 ? (lb0 ^ (ks * 32)) : (ks ? lb1 : lb0)));
    };
    auto read_w = [&](int buf, int nh) {
        const char *b = smem + buf * BUF_BYTES + (nh ? OFF_W1 : OFF_W0) + wbase;
#pragma unroll
        for (int jj = 0; jj < WF; jj++)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) wf[nh][jj][ks] = *(const half8_t *)(b + jj * (RL * 128) + (MF32 ? (lb0 ^ (ks * 32)) : (ks ? lb1 : lb0)));
    };
    // accumulators: 16x16: [8 m reps][4 n reps] x 4 floats; 32x32: [4 m reps][2 n reps] x 16 floats -- 128 registers either way
    typedef typename std::conditional<MF32, float16_t, f32x4>::type acc_t;
    constexpr int AM = 2 * XF, AN = MF32 ? 2 : 4;
    acc_t acc[AM][AN];   // n rep j = WF nh + jj
#pragma unroll
    for (int i = 0; i < AM; i++)
#pragma unroll
        for (int j = 0; j < AN; j++) acc[i][j] = (acc_t)0.f;
    auto mma = [&](int mh, int nh) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int ii = 0; ii < XF; ii++)
#pragma unroll
                for (int jj = 0; jj < WF; jj++) {
                    acc_t &a = acc[mh * XF + ii][nh * WF + jj];
                    if constexpr (MF32) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nh][jj][ks], xf[ii][ks], a, 0, 0, 0);
                    else a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nh][jj][ks], xf[ii][ks], a, 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: tile 0 completely, X0 / W0 of tile 1; issue order per tile is X0, W0, W1, X1 everywhere
    stage_x(sx[0], 0 * BUF_BYTES + OFF_X0, 0);
    stage(sw[0], 0 * BUF_BYTES + OFF_W0, 0);
    stage(sw[1], 0 * BUF_BYTES + OFF_W1, 0);
    stage_x(sx[1], 0 * BUF_BYTES + OFF_X1, 0);
    stage_x(sx[0], 1 * BUF_BYTES + OFF_X0, 1);
    stage(sw[0], 1 * BUF_BYTES + OFF_W0, 1);
    G8_VMCNT8();          // X0(0), W0(0) of this wave have landed
    G8_BAR();
    if (wr == 1) G8_BAR();   // the stagger: wave row 1 runs one barrier behind wave row 0

    auto tile = [&](int t, auto buf_tag) {
        constexpr int b = decltype(buf_tag)::value;
        constexpr int cur = b * BUF_BYTES, oth = (b ^ 1) * BUF_BYTES;
        // R1
        read_w(b, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_x(b, 0);
        stage(sw[1], oth + OFF_W1, t + 1);
        G8_VMCNT8();      // W1(t) landed (read in R2, one barrier later for the staggered partner)
        G8_BAR();
        G8_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        G8_BAR();
        // R2
        read_w(b, 1);
        stage_x(sx[1], oth + OFF_X1, t + 1);
        G8_VMCNT8();      // X1(t) landed
        G8_BAR();
        G8_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1);
        G8_BAR();
        // R3
        read_x(b, 1);
        stage_x(sx[0], cur + OFF_X0, t + 2);   // X0(t) was last read two phases ago
        G8_BAR();
        G8_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1);
        G8_BAR();
        // R4
        stage(sw[0], cur + OFF_W0, t + 2);
        G8_VMCNT8();      // X0(t+1), W0(t+1) landed
        G8_BAR();
        mma(1, 0);
        G8_BAR();
    };
    for (int t = 0; t < nt; t += 2) {   // K % 128 == 0 (checked by the launcher): the buffer index is a compile-time constant
        tile(t, std::integral_constant<int, 0>{});
        tile(t + 1, std::integral_constant<int, 1>{});
    }
    if (wr == 0) G8_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-fetched tail tiles: no LDS-DMA may outlive the workgroup

    // ---- epilogue.  16x16: lane l holds, per (i, j), m = m0 + 128 wr + 16 i + l % 16 and four consecutive n starting at 4 (l / 16).
    // 32x32: per (i, j) and register group rg = reg / 4: m = m0 + 128 wr + 32 i + l % 32, n starting at 8 rg + 4 (l / 32).
    // Either way: one 8-byte store per four accumulators.
    constexpr int NG = MF32 ? 4 : 1;                  // groups of four consecutive n per accumulator tile
    const int mloc = wr * XU + frow;
    const int n4 = MF32 ? (lane >> 5) * 4 : (lane >> 4) * 4;
    auto acc4 = [&](const acc_t &a, int rg, int r) -> float {
        if constexpr (MF32) return a[4 * rg + r];
        else return a[r];
    };
#pragma unroll
    for (int i = 0; i < AM; i++) {
        const int m = m0 + mloc + i * RL;
        if (m >= M) continue;
        half_t *crow = p.c + (size_t)m * p.ldc;
#pragma unroll
        for (int rg = 0; rg < NG; rg++) {
            if constexpr (PAIR) {
#pragma unroll
                for (int jj = 0; jj < WF; jj++) {
                    const int n = n0 + wc * 32 + jj * RL + (MF32 ? 8 * rg : 0) + n4;
                    if (n >= N) continue;
                    half4_t h;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float g = acc4(acc[i][jj], rg, r), u = acc4(acc[i][WF + jj], rg, r);
                        h[r] = (half_t)(g * (1.0f / (1.0f + __expf(-g))) * u);   // SiLU on the fp32 accumulator (fused_mlp.py:160-165)
                    }
                    *(half4_t *)(crow + n) = h;
                }
            } else {
#pragma unroll
                for (int j = 0; j < AN; j++) {
                    const int n = n0 + wc * 64 + (j / WF) * 32 + (j % WF) * RL + (MF32 ? 8 * rg : 0) + n4;
                    if (n >= N) continue;
                    half4_t h;
#pragma unroll
                    for (int r = 0; r < 4; r++) h[r] = (half_t)acc4(acc[i][j], rg, r);
                    if (p.bias) {   // fp16(fp16(acc) + bias): the reference adds the bias to the rounded product (quant_linear.py:376)
                        const half4_t bv = *(const half4_t *)(p.bias + n);
#pragma unroll
                        for (int r = 0; r < 4; r++) h[r] = (half_t)((float)h[r] + (float)bv[r]);
                    }
                    *(half4_t *)(crow + n) = h;
                }
            }
        }
    }
}

template <bool PAIR, bool MF32, int XH>
int gemm8_launch(const Gemm8Params &p, hipStream_t s) {
    auto kern = gemm8_kernel<PAIR, MF32, XH>;
    constexpr int lds = 2 * BUF_BYTES;   // 131 072 B
    static LdsOptIn opt_in;   // per instantiation; per device inside
    if (int rc = opt_in.ensure((const void *)kern, lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(8 * p.per), dim3(512), lds, s, p);
    return (int)hipGetLastError();
}

std::atomic<int> g_gemm8_mfma{16};   // MFMA shape of the tile GEMM: 16 = 16x16x32, 32 = 32x32x16 (gemm8_set_mfma: tests / A-B runs)
std::atomic<int> g_gemm8_tile{0};    // rows of the workgroup tile: 0 = chosen per launch (gemm8_tile_rows), 192, 256 (gemm8_set_tile: tests / A-B runs)

// Rows of the workgroup tile for an [M, ntn n-tiles] product: the tile whose rounds cost less.  One workgroup per CU (128 KB of LDS), an XCD runs its
// ceil(tiles / 8) tiles in rounds of its 32 CUs, a workgroup's time is proportional to its rows; the 192-row tile pays G8_EFF192 per flop (12 MFMAs
// per phase against the same barriers, W fragment reads and DMA instructions as 16).  Measured: tools/bench_gemm8_tile.py, profiles/r5e_gemm8_tile/
// (3072 rows: 0.85-0.88 of the 256-row tile's time on all four LLaMA-7B shapes; 3584 and 4096 rows: 1.19-1.49, one round more).
constexpr double G8_EFF192 = 0.93;
int gemm8_tile_rows(int M, int ntn) {
    const int forced = g_gemm8_tile.load();
    if (forced) return forced;
    auto cost = [&](int rows, double eff) {
        const long tiles = (long)((M + rows - 1) / rows) * ntn, per_xcd = (tiles + 7) / 8;
        return (double)((per_xcd + 31) / 32) * rows / eff;
    };
    return cost(192, G8_EFF192) < cost(256, 1.0) ? 192 : 256;
}

}  // namespace

// c[M, N] = x[M, K] . wt[N, K]^T (+ bias); pair: c = silu(x . wt[0:N]^T) * (x . wt[N:2N]^T).
// Serves K % 128 == 0, N % 4 == 0, 16-byte aligned rows of x / wt, 8-byte aligned rows of c; GPTQ_E_VARIANT otherwise.
int gemm8_dense_f16(const half_t *x, int64_t ldx, const half_t *wt, int64_t ldw, const half_t *bias, half_t *c, int64_t ldc, int M, int K, int N,
                    bool pair, hipStream_t s) {
    if (M <= 0 || K <= 0 || N <= 0 || K % 128 != 0 || N % 4 != 0) return GPTQ_E_VARIANT;
    if (ldx % 8 != 0 || ldw % 8 != 0 || ldc % 4 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)wt % 16) != 0 || ((uintptr_t)c % 8) != 0) return GPTQ_E_VARIANT;
    if (bias && (pair || ((uintptr_t)bias % 8) != 0)) return GPTQ_E_VARIANT;
    Gemm8Params p;
    p.x = x; p.wt = wt; p.bias = bias; p.c = c;
    p.ldx = ldx; p.ldw = ldw; p.ldc = ldc;
    p.M = M; p.K = K; p.N = N;
    p.ntn = (N + (pair ? 128 : 256) - 1) / (pair ? 128 : 256);
    const bool mf32 = g_gemm8_mfma.load() == 32;
    const int rows = mf32 ? 256 : gemm8_tile_rows(M, p.ntn);
    p.ntm = (M + rows - 1) / rows;
    p.per = (p.ntm * p.ntn + 7) / 8;
    if (mf32) return pair ? gemm8_launch<true, true, 64>(p, s) : gemm8_launch<false, true, 64>(p, s);
    if (rows == 192) return pair ? gemm8_launch<true, false, 48>(p, s) : gemm8_launch<false, false, 48>(p, s);
    return pair ? gemm8_launch<true, false, 64>(p, s) : gemm8_launch<false, false, 64>(p, s);
}

int gemm8_set_mfma(int shape) { return (shape == 16 || shape == 32) ? g_gemm8_mfma.exchange(shape) : GPTQ_E_VARIANT; }
int gemm8_set_tile(int rows) { return (rows == 0 || rows == 192 || rows == 256) ? g_gemm8_tile.exchange(rows) : GPTQ_E_VARIANT; }

}  // namespace gptq
