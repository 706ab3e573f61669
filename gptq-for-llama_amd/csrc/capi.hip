// capi.hip -- extern "C" entry points of libgptq_mi355x.so (declared in include/gptq_mi355x.h)
// and the static shape -> kernel dispatch that replaces the reference's Triton autotuner
// (quant/custom_autotune.py:14-127, pruner :167-193) behind the same warm-up surface.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "gptq_internal.h"
#include "stripe_common.h"

namespace gptq {

static std::atomic<int> g_force_variant{-1};
static std::atomic<int> g_force_split_k{-1};
static std::atomic<void *> g_debug_buffer{nullptr};

static bool aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

struct Problem {
    const void *x;
    int64_t ldx;
    const int32_t *qw[2];
    const void *sc[2];
    const int32_t *qz[2];
    const int32_t *gi[2];
    const void *bias;
    void *y;
    int64_t ldy;
    int M, K, N, bits, groupsize;
    bool fused2;
    void *ws;
    size_t ws_bytes;
    const int32_t *xperm;  // x gathered through this permutation (re-sorted act-order layer; M == 1 rowwave only)
    const void *norm_w;  // fused RMSNorm prologue (M == 1, rowwave only)
    float norm_eps;
};

static int validate(const Problem &q) {
    if (q.bits != 2 && q.bits != 3 && q.bits != 4 && q.bits != 8) return GPTQ_E_BITS;
    if (q.M < 0 || q.K <= 0 || q.N <= 0 || q.groupsize <= 0) return GPTQ_E_SHAPE;
    if (q.K % 32 != 0 || q.N % 32 != 0) return GPTQ_E_SHAPE;
    const int ns = q.fused2 ? 2 : 1;
    if (!q.x || !q.y) return GPTQ_E_NULL;
    for (int s = 0; s < ns; s++)
        if (!q.qw[s] || !q.sc[s] || !q.qz[s]) return GPTQ_E_NULL;
    if (!aligned(q.x, 16) || q.ldx % 8 != 0 || !aligned(q.y, 8) || q.ldy % 4 != 0) return GPTQ_E_ALIGN;
    for (int s = 0; s < ns; s++)
        if (!aligned(q.qw[s], 16) || !aligned(q.sc[s], 8) || !aligned(q.qz[s], 4)) return GPTQ_E_ALIGN;
    if (q.bias && !aligned(q.bias, 2)) return GPTQ_E_ALIGN;
    return 0;
}

static int n_groups(int K, int gs) { return (K + gs - 1) / gs; }

static bool fast_eligible(const Problem &q, int unit_k) {
    if (q.bits == 3) return false;
    if (q.gi[0] || (q.fused2 && q.gi[1])) return false;
    return q.groupsize % unit_k == 0;
}

static void fill_params(const Problem &q, int m0, int mcount, GemvParams &p) {
    p.x = (const half_t *)q.x + (size_t)m0 * q.ldx;
    p.ldx = q.ldx;
    for (int s = 0; s < 2; s++) {
        p.qw[s] = (const uint32_t *)q.qw[s];
        p.sc[s] = (const half_t *)q.sc[s];
        p.qz[s] = q.qz[s];
        p.gi[s] = q.gi[s];
    }
    p.bias = (const half_t *)q.bias;
    p.y = (half_t *)q.y + (size_t)m0 * q.ldy;
    p.ldy = q.ldy;
    p.M = mcount;
    p.K = q.K;
    p.N = q.N;
    p.G = n_groups(q.K, q.groupsize);
    p.groupsize = q.groupsize;
    p.ws = (u64_t *)q.ws;
    p.xperm = q.xperm;
    p.norm_w = (const half_t *)q.norm_w;
    p.norm_eps = q.norm_eps;
    p.dbg = (u64_t *)g_debug_buffer.load();
}

static int ilog2_exact(int v) {
    for (int i = 0; i < 31; i++)
        if ((1 << i) == v) return i;
    return -1;
}

// The rowwave GEMV (gemv.hip) serves M == 1 (2 <= M <= 4 at 4 bits: run_rowwave_mr, one launch); larger M here means M launches (the dispatcher
// sends M >= 2 to the weight-streaming MFMA kernel instead).  Picks U = packed rows in flight
// per wave and S = workgroups per 256-column tile (DESIGN.md "dispatch").
// 3-bit rowwave: units are 32-k blocks (3 packed rows each)
static int run_rowwave3(const Problem &q, hipStream_t s) {
    if (q.norm_w || q.xperm) return GPTQ_E_VARIANT;
    const int nblocks = q.K / 32;
    const int G = n_groups(q.K, q.groupsize);
    int gshift = -1;
    if (G > 1) {
        if (q.groupsize % 32 != 0) return GPTQ_E_VARIANT;
        gshift = ilog2_exact(q.groupsize / 32);
        if (gshift < 0) return GPTQ_E_VARIANT;
    }
    const int bpg = G > 1 ? q.groupsize / 32 : nblocks;
    int u = 0;
    const int fv = g_force_variant.load();
    for (int c = 2; c >= 1 && !u; c >>= 1)
        if (nblocks % c == 0 && (G == 1 || bpg % c == 0)) u = c;
    if (fv >= 0) {
        if (fv > 1) return GPTQ_E_VARIANT;
        u = 2 >> fv;
        if (!(nblocks % u == 0 && (G == 1 || bpg % u == 0))) return GPTQ_E_VARIANT;
    }
    const int ntile = (q.N + 255) / 256;
    const int nchunk = (nblocks + 4 * u - 1) / (4 * u);
    int split_k = nchunk;
    if (ntile * split_k > 1024) split_k = 1024 / ntile;
    if (split_k < 1) split_k = 1;
    const int fs = g_force_split_k.load();
    if (fs >= 1) split_k = fs;
    if (split_k > nchunk) split_k = nchunk;
    if (split_k > (q.fused2 ? SPLITK_MAX_PAIR : SPLITK_MAX_SINGLE)) split_k = q.fused2 ? SPLITK_MAX_PAIR : SPLITK_MAX_SINGLE;
    // the per-column combine words live in the first SPLITK_TICKET_OFFSET bytes of the workspace (the rest belongs
    // to the stream kernel's tickets / partial tiles and is not zero): wider layers run without a K split
    const size_t ws_need = (size_t)q.N * 8 * (q.fused2 ? 2 : 1);   // one combine word per column (two for gate/up)
    const bool ws_ok = q.ws && aligned(q.ws, 8) && q.ws_bytes >= ws_need && ws_need <= SPLITK_TICKET_OFFSET;
    if (split_k > 1 && !ws_ok) {
        if (fs >= 1) return GPTQ_E_WORKSPACE;
        split_k = 1;
    }
    for (int m = 0; m < q.M; m++) {
        GemvParams p;
        fill_params(q, m, 1, p);
        p.split_k = split_k;
        p.upg_shift = gshift;
        int rc = gemv_fast_dispatch(3, q.fused2, u, p, s);
        if (rc) return rc;
    }
    return 0;
}

static int run_rowwave(const Problem &q, hipStream_t s) {
    if (q.bits == 3) return run_rowwave3(q, s);
    const int kpw = 32 / q.bits;
    const int rows = q.K / kpw;
    const int G = n_groups(q.K, q.groupsize);
    const int rpg = q.groupsize / kpw;  // packed rows per group
    int gshift = -1;
    if (G > 1) {
        if (q.groupsize % kpw != 0) return GPTQ_E_VARIANT;
        gshift = ilog2_exact(rpg);
        if (gshift < 0) return GPTQ_E_VARIANT;
    }
    auto u_ok = [&](int u) { return rows % u == 0 && (G == 1 || rpg % u == 0); };
    int u = 0;
    const int fv = g_force_variant.load();
    if (fv >= 0) {
        if (fv >= GEMV_NUM_VARIANTS) return GPTQ_E_VARIANT;
        u = 8 >> fv;
        if (!u_ok(u)) return GPTQ_E_VARIANT;
    } else {
        for (int c = 8; c >= 2 && !u; c >>= 1)
            if (u_ok(c)) u = c;
        if (!u) return GPTQ_E_VARIANT;
    }
    if (q.norm_w && (u != 8 || q.bits != 4 || q.M != 1)) return GPTQ_E_VARIANT;
    if (q.xperm && q.bits != 4) return GPTQ_E_VARIANT;
    const int ntile = (q.N + 255) / 256;
    const int nchunk = (rows + 4 * u - 1) / (4 * u);
    const int split_max = q.fused2 ? SPLITK_MAX_PAIR : SPLITK_MAX_SINGLE;
    int split_k = nchunk;                                  // one chunk per workgroup ...
    if (ntile * split_k > 1024) split_k = 1024 / ntile;    // ... unless that is > 4 workgroups per CU
    if (split_k < 1) split_k = 1;
    const int fs = g_force_split_k.load();
    if (fs >= 1) split_k = fs;
    if (split_k > nchunk) split_k = nchunk;
    if (split_k > split_max) split_k = split_max;
    // the per-column combine words live in the first SPLITK_TICKET_OFFSET bytes of the workspace (the rest belongs
    // to the stream kernel's tickets / partial tiles and is not zero): wider layers run without a K split
    const size_t ws_need = (size_t)q.N * 8 * (q.fused2 ? 2 : 1);   // one combine word per column (two for gate/up)
    const bool ws_ok = q.ws && aligned(q.ws, 8) && q.ws_bytes >= ws_need && ws_need <= SPLITK_TICKET_OFFSET;
    if (split_k > 1 && !ws_ok) {
        if (fs >= 1) return GPTQ_E_WORKSPACE;
        split_k = 1;
    }
    for (int m = 0; m < q.M; m++) {
        GemvParams p;
        fill_params(q, m, 1, p);
        p.split_k = split_k;
        p.upg_shift = gshift;
        int rc = gemv_fast_dispatch(q.bits, q.fused2, u, p, s);
        if (rc) return rc;
    }
    return 0;
}

static int run_generic(const Problem &q, hipStream_t s) {
    const int ns = q.fused2 ? 2 : 1;
    const int nchunks = q.K / 32;
    for (int m0 = 0; m0 < q.M;) {
        int mr = q.M - m0 >= 4 ? 4 : (q.M - m0 >= 2 ? 2 : 1);
        if (q.fused2 && mr > 2) mr = 2;
        GemvParams p;
        // LDS: x [mr][K] + g_idx + {s,z} table for the tile; narrow tiles when G is large
        const int G = n_groups(q.K, q.groupsize);
        int nl = 16;
        auto need = [&](int mr_, int nl_) {
            return (size_t)mr_ * q.K * 2 + (size_t)ns * q.K * 2 + 16 + (size_t)ns * G * 4 * nl_ * 4;
        };
        if (need(mr, nl) > 150 * 1024 || (q.N / 64) < 128) nl = 4;
        while (mr > 1 && need(mr, nl) > 150 * 1024) mr >>= 1;
        fill_params(q, m0, mr, p);
        p.ntiles = (q.N + 4 * nl - 1) / (4 * nl);
        p.split_k = 1;
        p.nchunks = nchunks;
        p.chunks_per_slice = nchunks;
        int rc = gemv_generic_dispatch(q.bits, q.fused2, nl, p, s);
        if (rc) return rc;
        m0 += mr;
    }
    return 0;
}

// 2 <= M <= 4, 4-bit, trivial g_idx: ONE rowwave launch for all rows.  M = 2: dot2 kernel with both x rows in SGPRs
// (gemv_rowwave_mr_kernel); M = 3, 4: MFMA 4x4x4 kernel (gemv_rowwave_mfma_kernel), 14 % faster there -- measured
// 4096^2 / 4096x11008, us: M = 2: 5.3 / 8.5 (MFMA 5.5 / 8.6); M = 4: 6.9 / 10.9 (dot2 7.9 / 12.8).  The time grows with
// M because the combine issues M * N * S returning atomics.  The two-block MFMA variant (M <= 8) is compiled and
// tested (gptq_set_gemv_variant(101)) but loses to the stream kernel (M = 8: 10.4 vs 8.5 us) and is not dispatched.
// gptq_set_gemv_variant(100) forces the dot2 kernel, 101 the MFMA kernel (A/B measurements, tests).
static int run_rowwave_mr(const Problem &q, hipStream_t s) {
    const int fv = g_force_variant.load();
    if (fv >= 0 && fv != 100 && fv != 101) return GPTQ_E_VARIANT;
    const bool mfma = fv == 101 || (fv != 100 && q.M >= 3);
    const int mmax = fv == 101 ? (q.fused2 ? 4 : 8) : 4;
    if (q.bits != 4 || q.M < 2 || q.M > mmax || q.norm_w || q.xperm) return GPTQ_E_VARIANT;
    if (!fast_eligible(q, 8)) return GPTQ_E_VARIANT;
    const int rows = q.K / 8;
    const int G = n_groups(q.K, q.groupsize);
    const int rpg = q.groupsize / 8;
    int gshift = -1;
    if (G > 1) {
        if (q.groupsize % 8 != 0) return GPTQ_E_VARIANT;
        gshift = ilog2_exact(rpg);
        if (gshift < 0) return GPTQ_E_VARIANT;
    }
    auto u_ok = [&](int u) { return rows % u == 0 && (G == 1 || rpg % u == 0); };
    int u = 0;
    const int uhi = (mfma || q.M == 2) ? 8 : 4, ulo = (mfma || q.M == 2) ? 4 : 2;   // dot2 variant: x rows live in SGPRs (MR * u * 4 dwords)
    for (int c = uhi; c >= ulo && !u; c >>= 1)
        if (u_ok(c)) u = c;
    if (!u) return GPTQ_E_VARIANT;
    const int ntile = (q.N + 255) / 256;
    const int nchunk = (rows + 4 * u - 1) / (4 * u);
    int split_k = nchunk;
    if (ntile * split_k > 1024) split_k = 1024 / ntile;
    if (split_k < 1) split_k = 1;
    const int fs = g_force_split_k.load();
    if (fs >= 1) split_k = fs;
    if (split_k > nchunk) split_k = nchunk;
    const int split_max = q.fused2 ? SPLITK_MAX_PAIR : SPLITK_MAX_SINGLE;
    if (split_k > split_max) split_k = split_max;
    const size_t words = (size_t)q.M * q.N * 8 * (q.fused2 ? 2 : 1);
    const bool ws_ok = q.ws && aligned(q.ws, 8) && q.ws_bytes >= words && words <= SPLITK_TICKET_OFFSET;
    if (split_k > 1 && !ws_ok) return GPTQ_E_VARIANT;   // the per-row / stream paths handle it
    GemvParams p;
    fill_params(q, 0, q.M, p);
    p.split_k = split_k;
    p.upg_shift = gshift;
    return mfma ? gemv_rowwave_mfma_dispatch(q.fused2, u, p, s) : gemv_rowwave_mr_dispatch(q.fused2, u, p, s);
}

static int run_gemv(const Problem &q, hipStream_t s) {
    if (q.M >= 2 && q.M <= 8) {   // (M > 4 only when the MFMA variant is forced)
        const int rc = run_rowwave_mr(q, s);
        if (rc != GPTQ_E_VARIANT) return rc;
    }
    const bool eligible3 = q.bits == 3 && !q.gi[0] && !(q.fused2 && q.gi[1]) && q.M <= GEMV_MAX_M;   // one launch per row
    if (eligible3 || fast_eligible(q, 32 / (q.bits == 3 ? 4 : q.bits))) {
        int rc = run_rowwave(q, s);
        if (rc != GPTQ_E_VARIANT || g_force_variant.load() >= 0) return rc;
    }
    return run_generic(q, s);
}

// The weight-streaming MFMA kernel (skinny_mfma.hip): M <= 64, trivial g_idx, whole groups.
static int run_skinny(const Problem &q, hipStream_t s) {
    const int unit_k = (q.bits == 2) ? 64 : 32;
    if (!fast_eligible(q, unit_k) || q.K % q.groupsize != 0) return run_gemv(q, s);
    const int upg = q.groupsize / unit_k;
    const int nunits = q.K / unit_k;
    const int mmax = q.fused2 ? 32 : SKINNY_MAX_M;
    const int split_max = 32;
    int waves = 4;
    const int fv = g_force_variant.load();
    if (fv == 2 || fv == 4 || fv == 8) waves = fv;
    for (int m0 = 0; m0 < q.M; m0 += mmax) {
        const int mc = (q.M - m0 < mmax) ? (q.M - m0) : mmax;
        GemvParams p;
        fill_params(q, m0, mc, p);
        p.ntiles = (q.N + 63) / 64;
        p.units_per_group = upg;
        p.upg_shift = ilog2_exact(upg);
        // x rows staged in LDS: M <= 16 always, M <= 32 with the 4-unit-stage kernel (M = 64 measured slower than
        // re-reading x from L2: four A fragments per unit make the wave LDS-bound)
        bool xlds = mc <= 16 || (mc <= 32 && upg % 4 == 0);
        int stg = (upg % 4 == 0) ? 4 : ((upg % 2 == 0 && xlds) ? 2 : 1);
        const int nstages = nunits / stg;
        const int w = (xlds && stg == 4) ? waves : 4;
        // K slices publish partial tiles [tile][slice][sets][rows][64] fp32 behind the ticket words
        const size_t part_per_slice = (size_t)p.ntiles * (q.fused2 ? 2 : 1) * (mc < 64 ? mc : 64) * 64 * 4;
        const bool ws_ok = q.ws && aligned(q.ws, 8) && q.ws_bytes > SPLITK_PART_OFFSET + 2 * part_per_slice && p.ntiles <= 1024;
        const int split_fit = ws_ok ? (int)((q.ws_bytes - SPLITK_PART_OFFSET) / part_per_slice) : 1;
        // K split: ~2 workgroups per CU, every wave of a workgroup gets at least one stage
        int split_k = 1;
        const int fs = g_force_split_k.load();
        if (fs >= 1) {
            split_k = fs;
            if (split_k > 1 && (!ws_ok || split_k > split_fit)) return GPTQ_E_WORKSPACE;
        } else if (ws_ok) {
            split_k = (512 + p.ntiles - 1) / p.ntiles;
            if (split_k > split_fit) split_k = split_fit;
        }
        if (split_k > split_max) split_k = split_max;
        if (split_k * w > nstages) split_k = nstages / w;
        if (split_k < 1) split_k = 1;
        int sps = (nstages + split_k - 1) / split_k;  // stages per slice
        // staged x must fit: rows * (units * unit_k + 8) halves
        const int xrows = mc;
        auto x_bytes = [&](int sps_) { return (size_t)xrows * ((size_t)sps_ * stg * unit_k + 8) * 2 + 16; };
        if (xlds && x_bytes(sps) > 64 * 1024) {
            if (ws_ok && fs < 1) {
                while (x_bytes(sps) > 64 * 1024 && split_k < split_max && (split_k + 1) * w <= nstages) {
                    split_k++;
                    sps = (nstages + split_k - 1) / split_k;
                }
            }
            if (x_bytes(sps) > 64 * 1024) xlds = false;
        }
        if (!xlds && stg == 2) return run_gemv(q, s);
        split_k = (nstages + sps - 1) / sps;
        if (split_k > 1 && split_k > split_fit) return GPTQ_E_WORKSPACE;
        p.split_k = split_k;
        p.nchunks = nunits;
        p.chunks_per_slice = sps * stg;
        int rc = skinny_dispatch(q.bits, q.fused2, stg, (xlds && stg == 4) ? waves : 4, xlds, p, s);
        if (rc) return rc;
    }
    return 0;
}

static int run_auto(const Problem &q, hipStream_t s) {
    if (q.M == 0) return 0;
    if (q.M <= 2) return run_gemv(q, s);               // rowwave GEMV (M = 2: both rows in one launch; generic kernel for act-order)
    if (q.M <= 8) {                                    // small decode batch (M <= 4; M <= 8 only when the MFMA variant is forced)
        const int rc = run_rowwave_mr(q, s);
        if (rc != GPTQ_E_VARIANT) return rc;
    }
    if (q.M <= SKINNY_MAX_M) return run_skinny(q, s);  // falls back to the GEMV for act-order / 3-bit
    // 65 <= M <= 256: a single 256-row tile per column block cannot fill 256 CUs (4096^2: 111-113 us at M = 128 / 256); 64-row
    // passes of the weight-streaming kernel are faster there (44 / 88 us; from M = 257 the tile GEMM has two tile rows and wins).  (The Python layer prefers dequantise-once +
    // dense GEMM for this regime; this is for callers of the C ABI.)
    if (!q.fused2 && q.M <= 4 * SKINNY_MAX_M) return run_skinny(q, s);
    const int unit_k = (q.bits == 2) ? 64 : 32;
    if (fast_eligible(q, unit_k)) {
        GemvParams p;
        fill_params(q, 0, q.M, p);
        // fused gate/up at prefill sizes: ONE tile GEMM whose workgroups hold gate and up of the same columns; SiLU on the fp32 sums
        const int rc = gemm_dispatch(q.bits, q.fused2, p, s);
        if (rc != GPTQ_E_VARIANT) return rc;
    }
    return run_skinny(q, s);
}

// debug hook: a device uint32 every stripe decode launch (M = 1) increments when it starts (tools/warmlab.hip B paces its run-ahead
// prefetcher on it); NULL = no tick (the product never sets it)
static std::atomic<uint32_t *> g_progress_counter{nullptr};
uint32_t *stripe_progress_counter() { return g_progress_counter.load(std::memory_order_relaxed); }

}  // namespace gptq

using namespace gptq;

extern "C" {

int gptq_set_progress_counter(void *device_u32) {
    g_progress_counter.store((uint32_t *)device_u32);
    return 0;
}

int gptq_query(int what) {
    switch (what) {
        case GPTQ_Q_ABI_VERSION: return 1;
        case GPTQ_Q_GEMV_MAX_M: return GEMV_MAX_M;
        case GPTQ_Q_SKINNY_MAX_M: return SKINNY_MAX_M;
        case GPTQ_Q_WORKSPACE_BYTES: return (int)WS_BYTES;
        case GPTQ_Q_NUM_GEMV_VARIANTS: return GEMV_NUM_VARIANTS;
        case GPTQ_Q_STRIPE_MM_WORKSPACE_BYTES: return (int)STRIPE_MM_WS_BYTES;
    }
    return -1;
}

const char *gptq_strerror(int code) {
    switch (code) {
        case GPTQ_OK: return "ok";
        case GPTQ_E_BITS: return "Only 2,3,4,8 bits are supported.";
        case GPTQ_E_SHAPE: return "bad shape: K and N must be positive multiples of 32, groupsize > 0";
        case GPTQ_E_ALIGN: return "pointer or leading-dimension alignment";
        case GPTQ_E_NULL: return "required pointer is NULL";
        case GPTQ_E_WORKSPACE: return "workspace missing or too small for the requested split-K";
        case GPTQ_E_VARIANT: return "unknown or inapplicable kernel variant";
        case GPTQ_E_NORM_WIDTH: return "This layer norm doesn't support feature dim >= 64KB.";
        case GPTQ_E_LIBRARY: return "prefill route: hipBLASLt could not be loaded or refused the product";
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

int gptq_set_gemv_variant(int variant) { return g_force_variant.exchange(variant); }
int gptq_set_split_k(int split_k) { return g_force_split_k.exchange(split_k); }
void *gptq_set_debug_buffer(void *buf) { return g_debug_buffer.exchange(buf); }
/* test support (tests/test_gpu_soak.py): one launch that overwrites the whole LDS of every CU with pattern ^ index */
int gptq_debug_dirty_lds(uint32_t pattern, gptq_stream_t stream) { return dirty_lds_launch(pattern, (hipStream_t)stream); }
int gptq_set_gemm_kernel(int version) {
    return (version == 2 || version == 3 || (version >= 100 && version <= 104)) ? gemm_set_version(version) : GPTQ_E_VARIANT;
}

static Problem make_problem(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros,
                            const int32_t *g_idx, const void *bias, void *y, int64_t ldy, int M, int K, int N, int bits,
                            int groupsize, void *ws, size_t ws_bytes) {
    Problem q{};
    q.x = x;
    q.ldx = ldx;
    q.qw[0] = qweight;
    q.sc[0] = scales;
    q.qz[0] = qzeros;
    q.gi[0] = g_idx;
    q.bias = bias;
    q.y = y;
    q.ldy = ldy;
    q.M = M;
    q.K = K;
    q.N = N;
    q.bits = bits;
    q.groupsize = groupsize;
    q.fused2 = false;
    q.ws = ws;
    q.ws_bytes = ws_bytes;
    return q;
}

int gptq_matmul248_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros,
                       const int32_t *g_idx, const void *bias, void *y, int64_t ldy, int M, int K, int N, int bits,
                       int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    Problem q = make_problem(x, ldx, qweight, scales, qzeros, g_idx, bias, y, ldy, M, K, N, bits, groupsize, workspace,
                             workspace_bytes);
    if (int rc = validate(q)) return rc;
    return run_auto(q, (hipStream_t)stream);
}

/* round 6: the fp32 partial product of a ROW SHARD of a layer with any g_idx (an act-order layer cut for tensor parallelism: the shard's k range
 * is fixed by the heads / by gate-up's columns, its rows point into ALL groups of the layer) -- y32[M][ldy] = x[M][K] . W[rows of the shard],
 * the sums unrounded: the caller's all-reduce adds the ranks' partials and rounds ONCE (north_star: fp32 partials, one rounding per layer).
 * qweight: the shard's K / 32 * bits packed rows; scales [n_groups][N] / qzeros [n_groups][N / 32 * bits]: the WHOLE layer's tables;
 * g_idx [K]: the group of each of the shard's rows (required).  The weight is dequantised as the reference does (quant_linear.py:128). */
int gptq_matmul248_partial_f32(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *g_idx, float *y32,
                               int64_t ldy, int M, int K, int N, int bits, int n_groups, gptq_stream_t stream) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (M < 0 || K <= 0 || N <= 0 || n_groups <= 0 || n_groups > 65535 || K % 32 != 0 || N % 32 != 0 || ldx < K || ldy < N) return GPTQ_E_SHAPE;
    if (!x || !qweight || !scales || !qzeros || !g_idx || !y32) return GPTQ_E_NULL;
    if (!aligned(x, 2) || !aligned(qweight, 16) || !aligned(scales, 2) || !aligned(qzeros, 4) || !aligned(g_idx, 4) || !aligned(y32, 4)) return GPTQ_E_ALIGN;
    for (int m0 = 0; m0 < M;) {
        int mr = M - m0 >= 4 ? 4 : (M - m0 >= 2 ? 2 : 1);
        int nl = 16;
        auto need = [&](int mr_, int nl_) { return (size_t)mr_ * K * 2 + (size_t)K * 2 + 16 + (size_t)n_groups * 4 * nl_ * 4; };
        if (need(mr, nl) > 150 * 1024 || (N / 64) < 128) nl = 4;
        while (mr > 1 && need(mr, nl) > 150 * 1024) mr >>= 1;
        if (need(mr, nl) > 150 * 1024) return GPTQ_E_SHAPE;
        GemvParams p{};
        p.x = (const half_t *)x + (size_t)m0 * ldx;
        p.ldx = ldx;
        p.qw[0] = (const uint32_t *)qweight; p.sc[0] = (const half_t *)scales; p.qz[0] = qzeros; p.gi[0] = g_idx;
        p.y32 = y32 + (size_t)m0 * ldy;
        p.ldy = ldy;
        p.M = mr; p.K = K; p.N = N; p.G = n_groups; p.groupsize = 1;
        p.ntiles = (N + 4 * nl - 1) / (4 * nl);
        p.split_k = 1;
        p.nchunks = K / 32;
        p.chunks_per_slice = K / 32;
        if (int rc = gemv_generic_dispatch(bits, false, nl, p, (hipStream_t)stream)) return rc;
        m0 += mr;
    }
    return GPTQ_OK;
}

int gptq_gemv_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros,
                  const int32_t *g_idx, const void *bias, void *y, int64_t ldy, int M, int K, int N, int bits,
                  int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    Problem q = make_problem(x, ldx, qweight, scales, qzeros, g_idx, bias, y, ldy, M, K, N, bits, groupsize, workspace,
                             workspace_bytes);
    if (int rc = validate(q)) return rc;
    return run_gemv(q, (hipStream_t)stream);
}

int gptq_skinny_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros,
                    const int32_t *g_idx, const void *bias, void *y, int64_t ldy, int M, int K, int N, int bits,
                    int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    Problem q = make_problem(x, ldx, qweight, scales, qzeros, g_idx, bias, y, ldy, M, K, N, bits, groupsize, workspace,
                             workspace_bytes);
    if (int rc = validate(q)) return rc;
    return run_skinny(q, (hipStream_t)stream);
}

int gptq_gemm_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros,
                  const int32_t *g_idx, const void *bias, void *y, int64_t ldy, int M, int K, int N, int bits,
                  int groupsize, gptq_stream_t stream) {
    Problem q = make_problem(x, ldx, qweight, scales, qzeros, g_idx, bias, y, ldy, M, K, N, bits, groupsize, nullptr, 0);
    if (int rc = validate(q)) return rc;
    GemvParams p;
    fill_params(q, 0, M, p);
    return gemm_dispatch(bits, false, p, (hipStream_t)stream);
}

int gptq_fused_mlp_f16(const void *x, int64_t ldx, const int32_t *qweight_gate, const void *scales_gate,
                       const int32_t *qzeros_gate, const int32_t *g_idx_gate, const int32_t *qweight_up,
                       const void *scales_up, const int32_t *qzeros_up, const int32_t *g_idx_up, void *c, int64_t ldc,
                       int M, int K, int N, int bits, int groupsize, void *workspace, size_t workspace_bytes,
                       gptq_stream_t stream) {
    Problem q = make_problem(x, ldx, qweight_gate, scales_gate, qzeros_gate, g_idx_gate, nullptr, c, ldc, M, K, N, bits,
                             groupsize, workspace, workspace_bytes);
    q.fused2 = true;
    q.qw[1] = qweight_up;
    q.sc[1] = scales_up;
    q.qz[1] = qzeros_up;
    q.gi[1] = g_idx_up;
    if (int rc = validate(q)) return rc;
    return run_auto(q, (hipStream_t)stream);
}

int gptq_transpose_matmul248_f16(const void *dy, int64_t lddy, const int32_t *qweight, const void *scales,
                                 const int32_t *qzeros, const int32_t *g_idx, void *dx, int64_t lddx, int M, int K,
                                 int N, int bits, int groupsize, gptq_stream_t stream) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (M < 0 || K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0) return GPTQ_E_SHAPE;
    if (!dy || !dx || !qweight || !scales || !qzeros) return GPTQ_E_NULL;
    if (!aligned(dy, 16) || lddy % 8 != 0 || !aligned(dx, 16) || lddx % 8 != 0 || !aligned(qweight, 16)) return GPTQ_E_ALIGN;
    if (M == 0) return 0;
    return transpose_dispatch(bits, (const half_t *)dy, lddy, (const uint32_t *)qweight, (const half_t *)scales, qzeros,
                              g_idx, (half_t *)dx, lddx, M, K, N, n_groups(K, groupsize), groupsize, (hipStream_t)stream);
}

int gptq_rmsnorm_f16(const void *x, int64_t ldx, const void *weight, void *y, int64_t ldy, int M, int N, float eps,
                     gptq_stream_t stream) {
    if (!x || !weight || !y) return GPTQ_E_NULL;
    if (M < 0 || N <= 0) return GPTQ_E_SHAPE;
    if ((size_t)N * 2 > 65536) return GPTQ_E_NORM_WIDTH;
    return rmsnorm_launch((const half_t *)x, ldx, (const half_t *)weight, (half_t *)y, ldy, M, N, eps, (hipStream_t)stream);
}

int gptq_dense_matvec_f16(const void *x, const void *weight, int64_t ldw, const void *bias, void *y, int N, int K, const void *norm_weight,
                          float norm_eps, gptq_stream_t stream) {
    if (!x || !weight || !y) return GPTQ_E_NULL;
    if (N <= 0 || K <= 0 || K % 8 != 0 || ldw < K) return GPTQ_E_SHAPE;
    if (!aligned(x, 16) || !aligned(weight, 16) || ldw % 8 != 0 || !aligned(y, 2) || (norm_weight && !aligned(norm_weight, 16)) || (bias && !aligned(bias, 2)))
        return GPTQ_E_ALIGN;
    return dense_gemv_launch((const half_t *)x, (const half_t *)weight, ldw, (const half_t *)bias, (half_t *)y, N, K, (const half_t *)norm_weight, norm_eps,
                             (hipStream_t)stream);
}

/* M <= 16 rows of x against a dense fp16 weight [N][K] in ONE pass over the weight: the LM head of a decode batch */
int gptq_dense_matmat_f16(const void *x, int64_t ldx, const void *weight, int64_t ldw, const void *bias, void *y, int64_t ldy, int M, int N, int K,
                          const void *norm_weight, float norm_eps, gptq_stream_t stream) {
    if (!x || !weight || !y) return GPTQ_E_NULL;
    if (M < 0 || N <= 0 || K <= 0 || K % 8 != 0 || ldw < K || (M > 1 && (ldx < K || ldy < N))) return GPTQ_E_SHAPE;
    if (M > 16) return GPTQ_E_VARIANT;
    if (!aligned(x, 16) || !aligned(weight, 16) || ldw % 8 != 0 || !aligned(y, 2) || (norm_weight && !aligned(norm_weight, 16)) || (bias && !aligned(bias, 2)) ||
        (M > 1 && ldx % 8 != 0))
        return GPTQ_E_ALIGN;
    if (M == 0) return GPTQ_OK;
    return dense_gemv_launch((const half_t *)x, (const half_t *)weight, ldw, (const half_t *)bias, (half_t *)y, N, K, (const half_t *)norm_weight, norm_eps,
                             (hipStream_t)stream, M, ldx, ldy);
}

/* y[m][n] = fp16(y[m][n] + r[m][n]) */
int gptq_add_rows_f16(void *y, int64_t ldy, const void *r, int64_t ldr, int M, int N, gptq_stream_t stream) {
    if (!y || !r) return GPTQ_E_NULL;
    if (M < 0 || N <= 0 || ldy < N || ldr < N) return GPTQ_E_SHAPE;
    if (M == 0) return GPTQ_OK;
    return add_rows_launch((half_t *)y, ldy, (const half_t *)r, ldr, M, N, (hipStream_t)stream);
}

int gptq_rope_f16(void *qk, int64_t row_stride, const int64_t *position_ids, int64_t pos_batch_stride, int bsz, int seq,
                  int heads, int head_dim, float base, gptq_stream_t stream) {
    if (!qk || !position_ids) return GPTQ_E_NULL;
    if (bsz < 0 || seq < 0 || heads <= 0 || head_dim <= 0 || head_dim % 2 != 0) return GPTQ_E_SHAPE;
    return rope_launch((half_t *)qk, row_stride, position_ids, pos_batch_stride, bsz, seq, heads, head_dim, base,
                       (hipStream_t)stream);
}

int gptq_pack_f32(const float *weight, const float *scales, const float *zeros, const int32_t *g_idx, int K, int N,
                  int bits, int groupsize, int32_t *qweight, int32_t *qzeros, void *scales_f16, gptq_stream_t stream) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0) return GPTQ_E_SHAPE;
    if (!weight || !scales || !zeros || !qweight || !qzeros || !scales_f16) return GPTQ_E_NULL;
    return pack_launch(weight, scales, zeros, g_idx, K, N, n_groups(K, groupsize), bits, groupsize, qweight, qzeros,
                       (half_t *)scales_f16, (hipStream_t)stream);
}

int gptq_g_idx_is_trivial(const int32_t *g_idx, int K, int groupsize, int32_t *out, gptq_stream_t stream) {
    if (!g_idx || !out) return GPTQ_E_NULL;
    if (K <= 0 || groupsize <= 0) return GPTQ_E_SHAPE;
    return gidx_trivial_launch(g_idx, K, groupsize, out, (hipStream_t)stream);
}

int gptq_rmsnorm_matmul248_f16(const void *x, const void *norm_weight, float eps, const int32_t *qweight, const void *scales,
                               const int32_t *qzeros, const int32_t *g_idx, const void *bias, void *y, int K, int N, int bits,
                               int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    if (!norm_weight) return GPTQ_E_NULL;
    Problem q = make_problem(x, K, qweight, scales, qzeros, g_idx, bias, y, N, 1, K, N, bits, groupsize, workspace, workspace_bytes);
    q.norm_w = norm_weight;
    q.norm_eps = eps;
    if (int rc = validate(q)) return rc;
    if (!aligned(norm_weight, 2) || !fast_eligible(q, 32 / (bits == 3 ? 4 : bits))) return GPTQ_E_VARIANT;
    return run_rowwave(q, (hipStream_t)stream);
}

int gptq_rmsnorm_fused_mlp_f16(const void *x, const void *norm_weight, float eps, const int32_t *qweight_gate, const void *scales_gate,
                               const int32_t *qzeros_gate, const int32_t *g_idx_gate, const int32_t *qweight_up,
                               const void *scales_up, const int32_t *qzeros_up, const int32_t *g_idx_up, void *c, int K, int N,
                               int bits, int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    if (!norm_weight) return GPTQ_E_NULL;
    Problem q = make_problem(x, K, qweight_gate, scales_gate, qzeros_gate, g_idx_gate, nullptr, c, N, 1, K, N, bits, groupsize, workspace,
                             workspace_bytes);
    q.fused2 = true;
    q.qw[1] = qweight_up;
    q.sc[1] = scales_up;
    q.qz[1] = qzeros_up;
    q.gi[1] = g_idx_up;
    q.norm_w = norm_weight;
    q.norm_eps = eps;
    if (int rc = validate(q)) return rc;
    if (!fast_eligible(q, 32 / (bits == 3 ? 4 : bits))) return GPTQ_E_VARIANT;
    return run_rowwave(q, (hipStream_t)stream);
}

int gptq_dequant_f16(const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *g_idx, void *w, int K, int N,
                     int bits, int groupsize, gptq_stream_t stream) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0) return GPTQ_E_SHAPE;
    if (!qweight || !scales || !qzeros || !w) return GPTQ_E_NULL;
    return dequant_launch((const uint32_t *)qweight, (const half_t *)scales, qzeros, g_idx, K, N, n_groups(K, groupsize), groupsize, bits,
                          (half_t *)w, N, (hipStream_t)stream);
}

int gptq_dequant_ld_f16(const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *g_idx, void *w, int64_t ldw, int K,
                        int N, int bits, int groupsize, gptq_stream_t stream) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0 || ldw < N) return GPTQ_E_SHAPE;
    if (!qweight || !scales || !qzeros || !w) return GPTQ_E_NULL;
    return dequant_launch((const uint32_t *)qweight, (const half_t *)scales, qzeros, g_idx, K, N, n_groups(K, groupsize), groupsize, bits,
                          (half_t *)w, ldw, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- prefill route
// route 1 (default) = own kernels only: the hand-written GEMM of gemm8.hip on the transposed dequantised weight for every dense
// product it can run (K % 128 == 0); route 0 = hipBLASLt on the dequantised weight (the reported ceiling); the library is otherwise
// reached only by shapes gemm8 does not serve (K % 128 != 0).
static std::atomic<int> g_prefill_route{1};
std::atomic<int> g_stripe_mm_pass_rows{128};     // rows per pass of the 16-row MFMA tiles: 128, or 64 (round 2's schedule: A/B runs)
// batches of 129 .. this many rows run the fused tile GEMM on the stripe16 image (0: never).  Measured against the dense route with each
// engine forced (tools/bench_mid_prefill.py ALL_ROUTES=1, profiles/r4e_routes/routes_1k_4k.txt): the image route is the faster OWN kernel
// up to 2048 rows on every LLaMA-7B shape (1536 rows: 70 / 194 / 193 / 171 us against 94 / 213 / 194 / 215 for gemm8), gemm8 above
// (3072 rows: 126 / 330 / 311 / 268 against 126 / 335 / 339 / 306); against hipBLASLt the own kernels sit at 0.83-1.47x between 1025 and
// 4095 rows (median 0.96).
std::atomic<int> g_stripe_gemm_max_rows{2048};
namespace {
// Round 3 handed dense products below one full round of 256 x 256 tiles (and below 2048 rows) to hipBLASLt; since round 4 the product
// path is hand-written end to end: route 1 and 2 are the same (own kernel wherever it can run), route 0 = library only.  What the
// tile GEMM loses on a partial round of tiles (0.6-0.7x the library at 1025 rows) is only paid by layers WITHOUT a stripe16 image
// (2-bit batches above 128 rows, groups smaller than a row block, irregular act-order, GPTQ_STRIPE=0): everything else runs the fused
// tile GEMM on the image up to gptq_set_stripe_gemm_max_rows().
// (round 6: no per-shape decision lives here any more -- the tile choice is gemm8.hip's, image against dense is image_gemm_wanted's: this is the
// route switch alone, gptq_set_prefill_route(0) = library only)
bool own_dense_route() { return g_prefill_route.load() != 0; }
constexpr size_t PREFILL_LIB_WS = (size_t)76 << 20;      // what the library may use for itself (split / stream-K algorithms)
constexpr int PREFILL_CHUNK_M = 8192;                    // rows of the transient FP32 [rows, 2N] gate | up product
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int prefill_validate(const void *x, int64_t ldx, const void *y, int64_t ldy, int M, int K, int N, int bits, int groupsize) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (M < 0 || K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0 || ldx < K || ldy < N) return GPTQ_E_SHAPE;
    if (M > 0 && (!x || !y)) return GPTQ_E_NULL;
    if (((uintptr_t)x | (uintptr_t)y) % 16 != 0 || ldx % 8 != 0 || ldy % 8 != 0) return GPTQ_E_ALIGN;
    return GPTQ_OK;
}
}  // namespace

size_t gptq_prefill_workspace_bytes(int M, int K, int N, int nsets) {
    if (M < 0 || K <= 0 || N <= 0 || (nsets != 1 && nsets != 2)) return 0;
    size_t b = align256((size_t)K * N * nsets * 2) + PREFILL_LIB_WS;
    if (nsets == 2) b += align256((size_t)(M < PREFILL_CHUNK_M ? M : PREFILL_CHUNK_M) * 2 * N * 4);   // fp32: SiLU sees unrounded sums
    return b;
}

int gptq_prefill_matmul_f16(const void *x, int64_t ldx, const int32_t *qweight, const void *scales, const int32_t *qzeros,
                            const int32_t *g_idx, const void *bias, void *y, int64_t ldy, int M, int K, int N, int bits, int groupsize,
                            void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    if (int rc = prefill_validate(x, ldx, y, ldy, M, K, N, bits, groupsize)) return rc;
    if (!qweight || !scales || !qzeros) return GPTQ_E_NULL;
    if (M == 0) return GPTQ_OK;
    if (!workspace || (uintptr_t)workspace % 256 != 0 || workspace_bytes < gptq_prefill_workspace_bytes(M, K, N, 1)) return GPTQ_E_WORKSPACE;
    half_t *W = (half_t *)workspace;
    char *lib_ws = (char *)workspace + align256((size_t)K * N * 2);
    if (own_dense_route() && K % 128 == 0 && (!bias || (uintptr_t)bias % 8 == 0)) {
        // own route: Wt[N][K] (k contiguous) + the LDS-DMA / MFMA tile GEMM; no library, no transient beyond the weight itself
        if (int rc = dequant_t_launch((const uint32_t *)qweight, (const half_t *)scales, qzeros, g_idx, K, N, n_groups(K, groupsize), groupsize, bits, W,
                                      K, (hipStream_t)stream))
            return rc;
        const int rc = gemm8_dense_f16((const half_t *)x, ldx, W, K, (const half_t *)bias, (half_t *)y, ldy, M, K, N, false, (hipStream_t)stream);
        if (rc != GPTQ_E_VARIANT) return rc;
    }
    if (int rc = dequant_launch((const uint32_t *)qweight, (const half_t *)scales, qzeros, g_idx, K, N, n_groups(K, groupsize), groupsize, bits, W, N,
                                (hipStream_t)stream))
        return rc;
    return dense_gemm_f16((const half_t *)x, ldx, W, N, (const half_t *)bias, (half_t *)y, ldy, M, K, N, lib_ws, PREFILL_LIB_WS, (hipStream_t)stream);
}

int gptq_prefill_transpose_matmul248_f16(const void *dy, int64_t lddy, const int32_t *qweight, const void *scales, const int32_t *qzeros,
                                         const int32_t *g_idx, void *dx, int64_t lddx, int M, int K, int N, int bits, int groupsize, void *workspace,
                                         size_t workspace_bytes, gptq_stream_t stream) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (M < 0 || K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0 || lddy < N || lddx < K) return GPTQ_E_SHAPE;
    if (!qweight || !scales || !qzeros || (M > 0 && (!dy || !dx))) return GPTQ_E_NULL;
    if (((uintptr_t)dy | (uintptr_t)dx) % 16 != 0 || lddy % 8 != 0 || lddx % 8 != 0) return GPTQ_E_ALIGN;
    if (M == 0) return GPTQ_OK;
    if (!workspace || (uintptr_t)workspace % 256 != 0 || workspace_bytes < gptq_prefill_workspace_bytes(M, K, N, 1)) return GPTQ_E_WORKSPACE;
    half_t *W = (half_t *)workspace;
    char *lib_ws = (char *)workspace + align256((size_t)K * N * 2);
    if (int rc = dequant_launch((const uint32_t *)qweight, (const half_t *)scales, qzeros, g_idx, K, N, n_groups(K, groupsize), groupsize, bits, W, N,
                                (hipStream_t)stream))
        return rc;
    // dx[M, K] = dy[M, N] . W[K, N]^T: the "K" of this product is N, its "N" is K, W is stored [out, in] -- which IS the k-contiguous
    // operand layout of gemm8 (reference transpose_matmul_248_kernel, quant_linear.py:191-258): same tile engine, roles exchanged
    if (own_dense_route() && N % 128 == 0) {
        const int rc = gemm8_dense_f16((const half_t *)dy, lddy, W, N, nullptr, (half_t *)dx, lddx, M, N, K, false, (hipStream_t)stream);
        if (rc != GPTQ_E_VARIANT) return rc;
    }
    return dense_gemm_f16((const half_t *)dy, lddy, W, N, nullptr, (half_t *)dx, lddx, M, N, K, lib_ws, PREFILL_LIB_WS, (hipStream_t)stream, true);
}

int gptq_prefill_fused_mlp_f16(const void *x, int64_t ldx, const int32_t *qweight_gate, const void *scales_gate, const int32_t *qzeros_gate,
                               const int32_t *g_idx_gate, const int32_t *qweight_up, const void *scales_up, const int32_t *qzeros_up,
                               const int32_t *g_idx_up, void *c, int64_t ldc, int M, int K, int N, int bits, int groupsize, void *workspace,
                               size_t workspace_bytes, gptq_stream_t stream) {
    if (int rc = prefill_validate(x, ldx, c, ldc, M, K, N, bits, groupsize)) return rc;
    if (!qweight_gate || !scales_gate || !qzeros_gate || !qweight_up || !scales_up || !qzeros_up) return GPTQ_E_NULL;
    if (M == 0) return GPTQ_OK;
    if (!workspace || (uintptr_t)workspace % 256 != 0 || workspace_bytes < gptq_prefill_workspace_bytes(M, K, N, 2)) return GPTQ_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t N2 = 2 * (int64_t)N;
    half_t *W = (half_t *)workspace;                                                      // [K, 2N] = gate | up
    char *lib_ws = (char *)workspace + align256((size_t)K * N2 * 2);
    float *prod = (float *)(lib_ws + PREFILL_LIB_WS);                                      // [rows, 2N] fp32: the reference applies SiLU to the
                                                                                           // fp32 accumulators (fused_mlp.py:160-165), not to rounded products
    const int G = n_groups(K, groupsize);
    if (own_dense_route() && K % 128 == 0) {
        // own route: gate and up stacked as Wt[2N][K]; ONE launch computes both products per tile and applies SiLU to the fp32
        // accumulators in its epilogue (fused_mlp.py:160-165) -- no [M, 2N] intermediate at all
        half_t *Wt = (half_t *)workspace;
        if (int rc = dequant_t_launch((const uint32_t *)qweight_gate, (const half_t *)scales_gate, qzeros_gate, g_idx_gate, K, N, G, groupsize, bits, Wt, K, s))
            return rc;
        if (int rc = dequant_t_launch((const uint32_t *)qweight_up, (const half_t *)scales_up, qzeros_up, g_idx_up, K, N, G, groupsize, bits,
                                      Wt + (size_t)N * K, K, s))
            return rc;
        const int rc = gemm8_dense_f16((const half_t *)x, ldx, Wt, K, nullptr, (half_t *)c, ldc, M, K, N, true, s);
        if (rc != GPTQ_E_VARIANT) return rc;
    }
    if (int rc = dequant_launch((const uint32_t *)qweight_gate, (const half_t *)scales_gate, qzeros_gate, g_idx_gate, K, N, G, groupsize, bits, W, N2, s))
        return rc;
    if (int rc = dequant_launch((const uint32_t *)qweight_up, (const half_t *)scales_up, qzeros_up, g_idx_up, K, N, G, groupsize, bits, W + N, N2, s))
        return rc;
    for (int m0 = 0; m0 < M; m0 += PREFILL_CHUNK_M) {
        const int rows = M - m0 < PREFILL_CHUNK_M ? M - m0 : PREFILL_CHUNK_M;
        if (int rc = dense_gemm_f16((const half_t *)x + (size_t)m0 * ldx, ldx, W, N2, nullptr, prod, N2, rows, K, (int)N2, lib_ws, PREFILL_LIB_WS, s,
                                    false, /*out_f32=*/true))
            return rc;
        if (int rc = silu_mul_f32_launch(prod, N2, prod + N, N2, (half_t *)c + (size_t)m0 * ldc, ldc, rows, N, s)) return rc;
    }
    return GPTQ_OK;
}

/* which engine a dense product of this shape takes under the current route switch: 1 = the tile GEMM of gemm8.hip, 0 = hipBLASLt
 * (pure host logic; nsets = 2: the gate/up pair, trans = 1: the backward product dx[M, K] = dy[M, N] . W^T) */
int gptq_prefill_route_for(int M, int K, int N, int nsets, int trans) {
    if (M <= 0 || K <= 0 || N <= 0 || nsets < 1 || nsets > 2) return GPTQ_E_SHAPE;
    if (trans) return (own_dense_route() && N % 128 == 0) ? 1 : 0;
    return (own_dense_route() && K % 128 == 0) ? 1 : 0;
}
int gptq_set_library_enabled(int on) { return dense_gemm_set_enabled(on); }
int gptq_set_gemm8_mfma(int shape) { return gemm8_set_mfma(shape); }
int gptq_set_gemm8_tile(int rows) { return gemm8_set_tile(rows); }
int gptq_prefill_plan_count(void) { return dense_gemm_plan_count(); }

int gptq_set_prefill_route(int route) {
    if (route < 0 || route > 2) return GPTQ_E_VARIANT;
    return g_prefill_route.exchange(route);
}

int gptq_set_stripe_gemm_max_rows(int rows) {
    if (rows < 0) return GPTQ_E_VARIANT;
    return g_stripe_gemm_max_rows.exchange(rows);
}

int gptq_set_stripe_mm_pass_rows(int rows) {
    if (rows != 64 && rows != 128) return GPTQ_E_VARIANT;
    return g_stripe_mm_pass_rows.exchange(rows);
}

int gptq_silu_mul_f16(const void *gate, int64_t ldg, const void *up, int64_t ldu, void *c, int64_t ldc, int M, int N, gptq_stream_t stream) {
    if (M < 0 || N <= 0 || N % 8 != 0 || ldg < N || ldu < N || ldc < N || ldg % 8 != 0 || ldu % 8 != 0 || ldc % 8 != 0) return GPTQ_E_SHAPE;
    if (M == 0) return GPTQ_OK;
    if (!gate || !up || !c) return GPTQ_E_NULL;
    if (((uintptr_t)gate | (uintptr_t)up | (uintptr_t)c) % 16 != 0) return GPTQ_E_ALIGN;
    return silu_mul_launch((const half_t *)gate, ldg, (const half_t *)up, ldu, (half_t *)c, ldc, M, N, (hipStream_t)stream);
}

int gptq_act_order_repack(const int32_t *qweight, const int32_t *perm, int K, int N, int bits, int32_t *qweight_sorted,
                          gptq_stream_t stream) {
    if (!qweight || !perm || !qweight_sorted) return GPTQ_E_NULL;
    if (bits != 2 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (K <= 0 || N <= 0 || K % 32 != 0 || N % 32 != 0) return GPTQ_E_SHAPE;
    return act_order_repack_launch((const uint32_t *)qweight, perm, K, N, bits, (uint32_t *)qweight_sorted, (hipStream_t)stream);
}

int gptq_matmul248_sorted_f16(const void *x, int64_t ldx, const int32_t *perm, const int32_t *qweight_sorted, const void *scales,
                              const int32_t *qzeros, const void *bias, void *y, int64_t ldy, int M, int K, int N, int bits,
                              int groupsize, void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    if (!perm) return GPTQ_E_NULL;
    Problem q = make_problem(x, ldx, qweight_sorted, scales, qzeros, nullptr, bias, y, ldy, M, K, N, bits, groupsize, workspace,
                             workspace_bytes);
    q.xperm = perm;
    if (int rc = validate(q)) return rc;
    if (!fast_eligible(q, 32 / (bits == 3 ? 4 : bits))) return GPTQ_E_VARIANT;
    return run_rowwave(q, (hipStream_t)stream);   // one launch per row of x
}

int gptq_fused_mlp_sorted_f16(const void *x, int64_t ldx, const int32_t *perm, const int32_t *qweight_gate_sorted, const void *scales_gate,
                              const int32_t *qzeros_gate, const int32_t *qweight_up_sorted, const void *scales_up,
                              const int32_t *qzeros_up, void *c, int64_t ldc, int M, int K, int N, int bits, int groupsize,
                              void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    if (!perm) return GPTQ_E_NULL;
    Problem q = make_problem(x, ldx, qweight_gate_sorted, scales_gate, qzeros_gate, nullptr, nullptr, c, ldc, M, K, N, bits, groupsize,
                             workspace, workspace_bytes);
    q.fused2 = true;
    q.qw[1] = qweight_up_sorted;
    q.sc[1] = scales_up;
    q.qz[1] = qzeros_up;
    q.gi[1] = nullptr;
    q.xperm = perm;
    if (int rc = validate(q)) return rc;
    if (!fast_eligible(q, 32 / (bits == 3 ? 4 : bits))) return GPTQ_E_VARIANT;
    return run_rowwave(q, (hipStream_t)stream);   // one launch per row of x
}

int gptq_rmsnorm_sorted_f16(const void *x, const void *norm_weight, float eps, const int32_t *perm, const int32_t *qweight_sorted,
                            const void *scales, const int32_t *qzeros, const int32_t *qweight_up_sorted, const void *scales_up,
                            const int32_t *qzeros_up, const void *bias, void *y, int K, int N, int bits, int groupsize, void *workspace,
                            size_t workspace_bytes, gptq_stream_t stream) {
    if (!norm_weight || !perm) return GPTQ_E_NULL;
    Problem q = make_problem(x, K, qweight_sorted, scales, qzeros, nullptr, bias, y, N, 1, K, N, bits, groupsize, workspace, workspace_bytes);
    if (qweight_up_sorted) {
        if (bias) return GPTQ_E_VARIANT;
        q.fused2 = true;
        q.qw[1] = qweight_up_sorted;
        q.sc[1] = scales_up;
        q.qz[1] = qzeros_up;
        q.gi[1] = nullptr;
    }
    q.xperm = perm;
    q.norm_w = norm_weight;
    q.norm_eps = eps;
    if (int rc = validate(q)) return rc;
    if (!aligned(norm_weight, 2) || !fast_eligible(q, 32 / (bits == 3 ? 4 : bits))) return GPTQ_E_VARIANT;
    return run_rowwave(q, (hipStream_t)stream);
}

int gptq_decode_rope_kv_f16(void *qkv, const int64_t *position, void *k_cache, void *v_cache, int heads, int head_dim, int t_max,
                            float base, gptq_stream_t stream) {
    if (!qkv || !position || !k_cache || !v_cache) return GPTQ_E_NULL;
    if (heads <= 0 || head_dim <= 0 || head_dim % 2 != 0 || head_dim > 512 || t_max <= 0) return GPTQ_E_SHAPE;
    return decode_rope_kv_launch((half_t *)qkv, position, (half_t *)k_cache, (half_t *)v_cache, heads, head_dim, t_max, base,
                                 (hipStream_t)stream);
}

size_t gptq_decode_attn_workspace_bytes(int heads, int head_dim, int t_max) {
    if (heads <= 0 || head_dim != 128 || t_max <= 0) return 0;
    return decode_attn_ws_bytes(heads, t_max);
}

int gptq_decode_attn_f16(const void *q, const void *k_cache, const void *v_cache, const int64_t *position, void *out, void *workspace,
                         size_t workspace_bytes, int heads, int head_dim, int t_max, float scale, gptq_stream_t stream) {
    if (!q || !k_cache || !v_cache || !position || !out || !workspace) return GPTQ_E_NULL;
    if (heads <= 0 || head_dim != 128 || t_max <= 0) return GPTQ_E_SHAPE;
    if (!aligned(q, 16) || !aligned(k_cache, 16) || !aligned(v_cache, 16) || !aligned(workspace, 16)) return GPTQ_E_ALIGN;
    if (workspace_bytes < decode_attn_ws_bytes(heads, t_max)) return GPTQ_E_WORKSPACE;
    return decode_attn_launch((const half_t *)q, (const half_t *)k_cache, (const half_t *)v_cache, position, (half_t *)out,
                              (float *)workspace, heads, t_max, scale, (hipStream_t)stream);
}

int gptq_decode_attn_fused_f16(const void *qkv, const int64_t *position, void *k_cache, void *v_cache, void *out, void *workspace,
                               size_t workspace_bytes, int heads, int head_dim, int t_max, float base, float scale,
                               gptq_stream_t stream) {
    if (!qkv || !k_cache || !v_cache || !position || !out || !workspace) return GPTQ_E_NULL;
    if (heads <= 0 || head_dim != 128 || t_max <= 0) return GPTQ_E_SHAPE;
    if (!aligned(qkv, 16) || !aligned(k_cache, 16) || !aligned(v_cache, 16) || !aligned(workspace, 16)) return GPTQ_E_ALIGN;
    if (workspace_bytes < decode_attn_ws_bytes(heads, t_max)) return GPTQ_E_WORKSPACE;
    return decode_attn_fused_launch((const half_t *)qkv, position, (half_t *)k_cache, (half_t *)v_cache, (half_t *)out,
                                    (float *)workspace, heads, t_max, base, scale, nullptr, (hipStream_t)stream, 1, 3 * (int64_t)heads * head_dim,
                                    (int64_t)heads * head_dim);
}

int gptq_rope_table_f32(float *table, int t_max, int head_dim, float base, gptq_stream_t stream) {
    if (!table) return GPTQ_E_NULL;
    if (t_max <= 0 || head_dim != 128) return GPTQ_E_SHAPE;
    if (!aligned(table, 8)) return GPTQ_E_ALIGN;
    return rope_table_launch(table, t_max, head_dim, base, (hipStream_t)stream);
}

int gptq_decode_attn_fused_table_f16(const void *qkv, const int64_t *position, void *k_cache, void *v_cache, void *out, void *workspace,
                                     size_t workspace_bytes, int heads, int head_dim, int t_max, float base, float scale,
                                     const float *rope_table, gptq_stream_t stream) {
    if (!qkv || !k_cache || !v_cache || !position || !out || !workspace || !rope_table) return GPTQ_E_NULL;
    if (heads <= 0 || head_dim != 128 || t_max <= 0) return GPTQ_E_SHAPE;
    if (!aligned(qkv, 16) || !aligned(k_cache, 16) || !aligned(v_cache, 16) || !aligned(workspace, 16) || !aligned(rope_table, 8)) return GPTQ_E_ALIGN;
    if (workspace_bytes < decode_attn_ws_bytes(heads, t_max)) return GPTQ_E_WORKSPACE;
    return decode_attn_fused_launch((const half_t *)qkv, position, (half_t *)k_cache, (half_t *)v_cache, (half_t *)out,
                                    (float *)workspace, heads, t_max, base, scale, rope_table, (hipStream_t)stream, 1, 3 * (int64_t)heads * head_dim,
                                    (int64_t)heads * head_dim);
}

/* round 5: a decode BATCH -- row b has its own position (positions[b]; negative = idle row), qkv row (ldq apart), output row (ldo apart) and
 * [t_max][heads * head_dim] slice of k_cache / v_cache.  rope_table may be NULL (trig in the kernel). */
size_t gptq_decode_attn_batch_workspace_bytes(int batch, int heads, int head_dim, int t_max) {
    if (batch <= 0 || heads <= 0 || head_dim != 128 || t_max <= 0) return 0;
    return decode_attn_ws_bytes(heads, t_max, batch);
}

int gptq_decode_attn_batch_f16(const void *qkv, int64_t ldq, const int64_t *positions, void *k_cache, void *v_cache, void *out, int64_t ldo,
                               void *workspace, size_t workspace_bytes, int batch, int heads, int head_dim, int t_max, float base, float scale,
                               const float *rope_table, const int32_t *out_perm, gptq_stream_t stream) {
    if (!qkv || !k_cache || !v_cache || !positions || !out || !workspace) return GPTQ_E_NULL;
    if (out_perm && !aligned(out_perm, 4)) return GPTQ_E_ALIGN;
    if (batch <= 0 || batch > 65535 || heads <= 0 || head_dim != 128 || t_max <= 0 || ldq < 3 * (int64_t)heads * head_dim || ldo < (int64_t)heads * head_dim)
        return GPTQ_E_SHAPE;
    if (!aligned(qkv, 16) || !aligned(k_cache, 16) || !aligned(v_cache, 16) || !aligned(workspace, 16) || (rope_table && !aligned(rope_table, 8)) ||
        ldq % 8 != 0 || ldo % 8 != 0 || !aligned(out, 2))
        return GPTQ_E_ALIGN;
    if (workspace_bytes < decode_attn_ws_bytes(heads, t_max, batch)) return GPTQ_E_WORKSPACE;
    return decode_attn_fused_launch((const half_t *)qkv, positions, (half_t *)k_cache, (half_t *)v_cache, (half_t *)out, (float *)workspace, heads, t_max,
                                    base, scale, rope_table, (hipStream_t)stream, batch, ldq, ldo, out_perm);
}

/* round 6: the same launch, but every active split of a row leaves its record {M, den, num[128]} (fp32) in the workspace and NOTHING merges them here:
 * the consumer is gptq_layer_decode_attn_f16 (o_proj's decode kernel stages x from the records).  tokens_per_split <= 0: the default. */
int gptq_decode_attn_split_f16(const void *qkv, int64_t ldq, const int64_t *positions, void *k_cache, void *v_cache, void *workspace,
                               size_t workspace_bytes, int batch, int heads, int head_dim, int t_max, float base, float scale, const float *rope_table,
                               int tokens_per_split, gptq_stream_t stream) {
    if (!qkv || !k_cache || !v_cache || !positions || !workspace) return GPTQ_E_NULL;
    if (batch <= 0 || batch > 65535 || heads <= 0 || head_dim != 128 || t_max <= 0 || ldq < 3 * (int64_t)heads * head_dim) return GPTQ_E_SHAPE;
    if (!aligned(qkv, 16) || !aligned(k_cache, 16) || !aligned(v_cache, 16) || !aligned(workspace, 16) || (rope_table && !aligned(rope_table, 8)) ||
        ldq % 8 != 0)
        return GPTQ_E_ALIGN;
    if (workspace_bytes < decode_attn_ws_bytes(heads, t_max, batch)) return GPTQ_E_WORKSPACE;
    return decode_attn_fused_launch((const half_t *)qkv, positions, (half_t *)k_cache, (half_t *)v_cache, nullptr, (float *)workspace, heads, t_max, base,
                                    scale, rope_table, (hipStream_t)stream, batch, ldq, 0, nullptr, true, tokens_per_split);
}

/* splits per (row, head) of a launch's grid -- also the S of the workspace layout [batch][S][heads * 128] | [batch][S][heads][2] | tickets */
int gptq_decode_attn_splits(int batch, int heads, int head_dim, int t_max) {
    if (batch <= 0 || heads <= 0 || head_dim != 128 || t_max <= 0) return GPTQ_E_SHAPE;
    return decode_attn_grid_splits(heads, t_max, batch);
}

// ---- stripe16: no-split-K decode GEMV on a load-time repacked copy (stripe*.hip) ----
size_t gptq_stripe_bytes(int K, int N, int bits, int groupsize, int nsets) { return stripe_total_bytes(K, N, bits, groupsize, nsets); }

int gptq_stripe_repack(const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *qweight_up, const void *scales_up,
                       const int32_t *qzeros_up, void *stripes, size_t stripes_bytes, int K, int N, int bits, int groupsize,
                       gptq_stream_t stream) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0) return GPTQ_E_SHAPE;
    if (!qweight || !scales || !qzeros || !stripes) return GPTQ_E_NULL;
    const bool fused = qweight_up != nullptr;
    if (fused && (!scales_up || !qzeros_up)) return GPTQ_E_NULL;
    const size_t need = stripe_total_bytes(K, N, bits, groupsize, fused ? 2 : 1);
    if (need == 0) return GPTQ_E_VARIANT;
    if (stripes_bytes < need) return GPTQ_E_WORKSPACE;
    if (!aligned(qweight, 4) || !aligned(scales, 2) || !aligned(qzeros, 4) || !aligned(stripes, 16)) return GPTQ_E_ALIGN;
    return stripe_repack_launch((const uint32_t *)qweight, (const half_t *)scales, qzeros, (const uint32_t *)qweight_up,
                                (const half_t *)scales_up, qzeros_up, stripes, K, N, bits, groupsize, (hipStream_t)stream);
}

// round 6: "also write h = rmsnorm(y) * w" (gptq_layer_decode_next_norm_f16); *done is set by the route that did
struct NextNorm {
    const void *w;
    float eps;
    void *h;
    int64_t ldh;
    int *done;
};

static int stripe_matvec(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, const void *bias, void *y, int64_t ldy, float *y32,
                         int M, int K, int N, int bits, int groupsize, int nsets, const void *norm_weight, float norm_eps, const uint16_t *perm,
                         gptq_stream_t stream, void *mm_ws = nullptr, size_t mm_ws_bytes = 0, bool gemm_only_above_128 = false, int64_t ldb = 0,
                         const int32_t *yperm = nullptr, const AttnMerge *att = nullptr, const NextNorm *nn = nullptr) {
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (M < 0 || K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0 || nsets < 1 || nsets > 2) return GPTQ_E_SHAPE;
    if (!x || !stripes || (!y && !y32)) return GPTQ_E_NULL;
    const int gq = stripe_gq_shift(K, N, bits, groupsize);
    const int gemm_max = g_stripe_gemm_max_rows.load();
    const bool gemm_rows = mm_ws && M > 128 && M <= gemm_max;
    if (gq == -2 || (M > (mm_ws ? 256 : 16) && !gemm_rows) || (nsets == 2 && bias) || (y32 && bias)) return GPTQ_E_VARIANT;
    if (M > 1 && (perm || (y32 && (M > 4 || norm_weight)))) return GPTQ_E_VARIANT;
    if (norm_weight && mm_ws) return GPTQ_E_VARIANT;        // the fused norm lives in the decode kernel only (every row its own rstd)
    if (ldb != 0 && (!bias || ldb < N || y32)) return GPTQ_E_SHAPE;
    if (yperm && (mm_ws || y32 || !aligned(yperm, 4))) return GPTQ_E_VARIANT;          // the scattered store lives in the decode kernel only
    if (stripes_bytes < stripe_total_bytes(K, N, bits, groupsize, nsets)) return GPTQ_E_WORKSPACE;
    if (!aligned(x, 16) || !aligned(stripes, 16) || !aligned(y, 2) || !aligned(y32, 4) || (norm_weight && !aligned(norm_weight, 16)) ||
        (perm && !aligned(perm, 16)) || (M > 1 && (ldx % 8 != 0 || ldy < N)))
        return GPTQ_E_ALIGN;
    if (M == 0) return 0;
    StripeParams p{};
    p.x = (const half_t *)x;
    p.ldx = ldx;
    p.ldy = ldy;
    p.R = (const uint32_t *)stripes;
    p.tab = (const uint32_t *)((const char *)stripes + stripe_tab_offset(K, N, bits, nsets));
    p.y = (half_t *)y;
    p.y32 = y32;
    p.bias = (const half_t *)bias;
    p.ldb = ldb;
    p.yperm = yperm;
    p.norm_w = (const half_t *)norm_weight;
    p.norm_eps = norm_eps;
    p.xperm = perm;
    p.M = M;
    p.K = K;
    p.N = N;
    p.G = groupsize >= K ? 1 : K / groupsize;
    p.NS = nsets;
    p.gq_shift = gq;
    p.bits = bits;
    p.progress = M == 1 ? stripe_progress_counter() : nullptr;
    if (nn && nn->w && mm_ws && M <= 16) {   // (16-row tiles: the slices' combine launch may write the next norm's rows too)
        p.next_norm_w = (const half_t *)nn->w; p.next_norm_eps = nn->eps; p.h = (half_t *)nn->h; p.ldh = nn->ldh; p.next_norm_done = nn->done;
    }
    if (att && att->o16) {
        if (M != 1 || mm_ws || y32 || yperm || perm || norm_weight || nsets != 1) return GPTQ_E_VARIANT;
        p.att = *att;
    }
    if (gemm_rows) {   // 2-D tiles, weights kept packed (stripe_gemm_kernel)
        int rc;
        switch (bits) {
            case 2: rc = stripe_gemm_dispatch_b2(p, mm_ws, mm_ws_bytes, (hipStream_t)stream); break;
            case 3: rc = stripe_gemm_dispatch_b3(p, mm_ws, mm_ws_bytes, (hipStream_t)stream); break;
            case 4: rc = stripe_gemm_dispatch_b4(p, mm_ws, mm_ws_bytes, (hipStream_t)stream); break;
            default: rc = stripe_gemm_dispatch_b8(p, mm_ws, mm_ws_bytes, (hipStream_t)stream); break;
        }
        if (rc != GPTQ_E_VARIANT) return rc;
    }
    if (mm_ws && M > 128 && (gemm_only_above_128 || M > 256)) return GPTQ_E_VARIANT;
    if (mm_ws) {   // row tiles on the matrix core (stripe_mm.inc): passes of up to 128 rows (64 when the partial tiles of a 128-row pass
                   // do not fit the scratch, or with gptq_set_stripe_mm_pass_rows(64)), each streams the weights once
        const int forced = g_force_split_k.load();
        auto pass = [&](int m0, int rows) {
            p.x = (const half_t *)x + (size_t)m0 * ldx;
            p.y = (half_t *)y + (size_t)m0 * ldy;
            p.bias = (const half_t *)bias + (size_t)m0 * ldb;
            p.M = rows;
            switch (bits) {
                case 2: return stripe_mm_dispatch_b2(p, mm_ws, mm_ws_bytes, forced, (hipStream_t)stream);
                case 3: return stripe_mm_dispatch_b3(p, mm_ws, mm_ws_bytes, forced, (hipStream_t)stream);
                case 4: return stripe_mm_dispatch_b4(p, mm_ws, mm_ws_bytes, forced, (hipStream_t)stream);
                default: return stripe_mm_dispatch_b8(p, mm_ws, mm_ws_bytes, forced, (hipStream_t)stream);
            }
        };
        const int pass_rows = g_stripe_mm_pass_rows.load();
        for (int m0 = 0; m0 < M;) {
            int rows = std::min(pass_rows, M - m0);
            int rc = pass(m0, rows);
            if (rc == GPTQ_E_WORKSPACE && rows > 64) {   // nothing was launched: the same rows in 64-row passes
                rows = 64;
                rc = pass(m0, rows);
            }
            if (rc != 0) return rc;
            m0 += rows;
        }
        return 0;
    }
    return stripe_gemv_dispatch(p, (hipStream_t)stream);
}

int gptq_stripe_matmul_partial_f32(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, float *y_partial, int M, int K, int N,
                                   int bits, int groupsize, int nsets, gptq_stream_t stream) {
    if (!y_partial) return GPTQ_E_NULL;
    if (M < 1 || M > 4) return GPTQ_E_VARIANT;
    return stripe_matvec(x, ldx, stripes, stripes_bytes, nullptr, nullptr, (int64_t)nsets * N, y_partial, M, K, N, bits, groupsize, nsets, nullptr, 0.f, nullptr,
                         stream);
}

int gptq_stripe_matvec_f16(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, const void *bias, void *y, int64_t ldy, int M,
                           int K, int N, int bits, int groupsize, int nsets, const void *norm_weight, float norm_eps, const uint16_t *perm,
                           gptq_stream_t stream) {
    if (!y) return GPTQ_E_NULL;
    return stripe_matvec(x, ldx, stripes, stripes_bytes, bias, y, ldy, nullptr, M, K, N, bits, groupsize, nsets, norm_weight, norm_eps, perm, stream);
}

/* the same launch with the columns of y stored through a permutation: y[m][y_perm[n]] = result column n (round 5: the consumer of y is an act-order
 * layer whose image holds group-sorted rows -- written in ITS order, it runs the trivial kernel instead of gathering x per launch).  M <= 4 (8 / 16
 * where the row groups run); y_perm: int32 [N], a permutation of 0 .. N - 1. */
int gptq_stripe_matvec_perm_out_f16(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, const void *bias, void *y, int64_t ldy, int M,
                                    int K, int N, int bits, int groupsize, int nsets, const void *norm_weight, float norm_eps, const uint16_t *perm,
                                    const int32_t *y_perm, gptq_stream_t stream) {
    if (!y) return GPTQ_E_NULL;
    return stripe_matvec(x, ldx, stripes, stripes_bytes, bias, y, ldy, nullptr, M, K, N, bits, groupsize, nsets, norm_weight, norm_eps, perm, stream, nullptr, 0,
                         false, 0, y_perm);
}

int gptq_stripe_matmul_f16(const void *x, int64_t ldx, const void *stripes, size_t stripes_bytes, const void *bias, void *y, int64_t ldy, int M,
                           int K, int N, int bits, int groupsize, int nsets, void *workspace, size_t workspace_bytes, gptq_stream_t stream) {
    if (!y || !workspace) return GPTQ_E_NULL;
    if (!aligned(workspace, 256)) return GPTQ_E_ALIGN;
    return stripe_matvec(x, ldx, stripes, stripes_bytes, bias, y, ldy, nullptr, M, K, N, bits, groupsize, nsets, nullptr, 0.f, nullptr, stream, workspace,
                         workspace_bytes);
}

int gptq_stripe_matvec_partial_f32(const void *x, const void *stripes, size_t stripes_bytes, float *y_partial, int K, int N, int bits,
                                   int groupsize, int nsets, const uint16_t *perm, gptq_stream_t stream) {
    if (!y_partial) return GPTQ_E_NULL;
    return stripe_matvec(x, K, stripes, stripes_bytes, nullptr, nullptr, N, y_partial, 1, K, N, bits, groupsize, nsets, nullptr, 0.f, perm, stream);
}

// =================================================================================================================================
// Prepared layers: ONE handle per QuantLinear (or gate/up pair) that owns every derived copy and contains the WHOLE M dispatch.
// The reference has a single call site -- matmul248(input, qweight, scales, qzeros, g_idx, bits, maxq), quant_linear.py:263-269,
// behind QuantLinear.forward (:373-377), and fusedmatmul_248 behind QuantLlamaMLP (fused_mlp.py:203-218) -- and an autotuner that
// picks a kernel per M.  Here: gptq_layer_prepare() inspects g_idx ONCE (trivial / regular act-order / irregular), builds the
// stripe16 image (of the group-sorted rows for a regular act-order layer, with the permutation) into the caller's image buffer,
// and gptq_layer_forward() is the static M -> kernel table (decode matvec, row groups, 16-row MFMA tiles, prefill tile GEMM).
// Memory stays the caller's (image / workspace / scratch are plain device pointers); the handle itself is a small host object.
// =================================================================================================================================
}  // extern "C" (re-opened below)

struct gptq_layer {
    int K, N, bits, groupsize, nsets;
    int kind;                      // 0 trivial g_idx, 1 regular act-order (group-sorted image + permutation), 2 irregular (generic kernels)
    const int32_t *qw[2], *qz[2], *gi[2];
    const void *sc[2];
    const void *bias;
    void *image;
    size_t image_bytes;
    void *stripe;                  // stripe16 image (of the sorted rows when kind == 1) or nullptr
    size_t stripe_bytes;
    int32_t *perm32;               // kind == 1: sorted position -> original k
    uint16_t *perm16;
    int32_t *invperm32;            // kind == 1: original k -> sorted position (rebuilds the checkpoint rows out of the image)
    bool released;                 // the caller freed the checkpoint buffers: qw / sc / qz are gone, the image is the only copy
};

namespace {

inline size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr int LAYER_STRIPE_MM_MAX_M = 128;   // rows the 16-row MFMA tiles serve before the prefill route wins (profiles/r2c_mm/mid_m.txt)
constexpr int LAYER_PREFILL_MIN_M = 65;      // the dense route starts above the weight-streaming kernels

// host-side verdict on a g_idx vector: 0 trivial, 1 regular act-order (every group exactly `groupsize` members, as gptq.py:210-216
// produces; perm = stable argsort), 2 anything else
int classify_g_idx(const std::vector<int32_t> &g, int K, int groupsize, std::vector<int32_t> *perm) {
    bool trivial = true;
    for (int k = 0; k < K && trivial; k++) trivial = g[k] == k / groupsize;
    if (trivial) return 0;
    if (K % groupsize != 0) return 2;
    const int G = K / groupsize;
    std::vector<int> count(G, 0);
    for (int k = 0; k < K; k++) {
        if (g[k] < 0 || g[k] >= G) return 2;
        count[g[k]]++;
    }
    for (int c : count)
        if (c != groupsize) return 2;
    if (perm) {   // stable counting sort by group
        std::vector<int> next(G);
        for (int i = 0; i < G; i++) next[i] = i * groupsize;
        perm->assign(K, 0);
        for (int k = 0; k < K; k++) (*perm)[next[g[k]]++] = k;
    }
    return 1;
}

int fetch_g_idx(const int32_t *g_idx, int K, hipStream_t s, std::vector<int32_t> *out) {
    out->resize(K);
    hipError_t e = hipMemcpyAsync(out->data(), g_idx, (size_t)K * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return (int)e;
}

// (3-bit: the fields of the 96-bit blocks are gathered one by one while the image is written; a group must not split a 32-k block)
bool sorted_supported(int K, int bits, int groupsize) {
    const int unit = bits == 3 ? 32 : 32 / bits;
    return groupsize % unit == 0 && K % groupsize == 0 && K <= 65535;
}

}  // namespace

extern "C" {

/* 0 trivial, 1 regular act-order, 2 irregular; negative = GPTQ_E_*, > 2 never.  Synchronises the stream (load time). */
int gptq_layer_inspect(const int32_t *g_idx, int K, int groupsize, gptq_stream_t stream) {
    if (K <= 0 || groupsize <= 0) return GPTQ_E_SHAPE;
    if (!g_idx) return 0;
    std::vector<int32_t> g;
    if (int rc = fetch_g_idx(g_idx, K, (hipStream_t)stream, &g)) return rc > 0 ? -1000 - rc : rc;
    return classify_g_idx(g, K, groupsize, nullptr);
}

size_t gptq_layer_image_bytes(int K, int N, int bits, int groupsize, int nsets, int kind) {
    if (K <= 0 || N <= 0 || groupsize <= 0 || nsets < 1 || nsets > 2 || kind < 0 || kind > 2) return 0;
    if (kind == 2 || (kind == 1 && !sorted_supported(K, bits, groupsize))) return 0;
    const size_t st = stripe_total_bytes(K, N, bits, groupsize, nsets);
    size_t b = a256(st);
    if (kind == 1) b += 2 * a256((size_t)K * 4) + a256((size_t)K * 2);   // perm32, invperm32, perm16 (round 4: no group-sorted qweight copy any more)
    return b;
}

int gptq_layer_prepare(gptq_layer_t **out, const int32_t *qweight, const void *scales, const int32_t *qzeros, const int32_t *g_idx, const void *bias,
                       const int32_t *qweight_up, const void *scales_up, const int32_t *qzeros_up, const int32_t *g_idx_up, int K, int N, int bits,
                       int groupsize, void *image, size_t image_bytes, gptq_stream_t stream) {
    if (!out) return GPTQ_E_NULL;
    *out = nullptr;
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    if (K <= 0 || N <= 0 || groupsize <= 0 || K % 32 != 0 || N % 32 != 0) return GPTQ_E_SHAPE;
    if (!qweight || !scales || !qzeros) return GPTQ_E_NULL;
    const int nsets = qweight_up ? 2 : 1;
    if (nsets == 2 && (!scales_up || !qzeros_up || bias)) return GPTQ_E_NULL;
    if (!aligned(qweight, 16) || !aligned(scales, 8) || !aligned(qzeros, 4) || (bias && !aligned(bias, 2)) || (image && !aligned(image, 256))) return GPTQ_E_ALIGN;
    if (nsets == 2 && (!aligned(qweight_up, 16) || !aligned(scales_up, 8) || !aligned(qzeros_up, 4))) return GPTQ_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    // ---- g_idx verdict (once per layer, host side) ----
    // single set: its own kind.  pair: trivial + trivial -> 0; act-order + act-order with the SAME permutation (gate and up share
    // their input, hence their Hessian diagonal, as q/k/v do) -> 1; anything else -> 2 (generic kernels, g_idx per set)
    int kind = 0;
    std::vector<int32_t> perm;
    const int32_t *gis[2] = {g_idx, g_idx_up};
    int kinds[2] = {0, 0};
    for (int i = 0; i < nsets; i++) {
        if (!gis[i]) continue;
        std::vector<int32_t> g, pm;
        if (int rc = fetch_g_idx(gis[i], K, s, &g)) return rc;
        kinds[i] = classify_g_idx(g, K, groupsize, &pm);
        if (kinds[i] == 1) {
            if (!perm.empty() && perm != pm) kinds[i] = 2;
            else perm = pm;
        }
    }
    if (nsets == 1) kind = kinds[0];
    else if (kinds[0] == kinds[1] && kinds[0] != 2) kind = kinds[0];
    else kind = 2;
    if (kind == 1 && !sorted_supported(K, bits, groupsize)) kind = 2;
    gptq_layer *L = new (std::nothrow) gptq_layer();
    if (!L) return (int)hipErrorOutOfMemory;
    L->K = K; L->N = N; L->bits = bits; L->groupsize = groupsize; L->nsets = nsets; L->kind = kind;
    L->qw[0] = qweight; L->sc[0] = scales; L->qz[0] = qzeros; L->gi[0] = kinds[0] == 0 ? nullptr : g_idx;
    L->qw[1] = qweight_up; L->sc[1] = scales_up; L->qz[1] = qzeros_up; L->gi[1] = kinds[1] == 0 ? nullptr : g_idx_up;
    L->bias = bias;
    L->image = image; L->image_bytes = image_bytes;
    // ---- derived copies ----
    const size_t need = gptq_layer_image_bytes(K, N, bits, groupsize, nsets, kind);
    const size_t st_bytes = kind == 2 ? 0 : stripe_total_bytes(K, N, bits, groupsize, nsets);
    if (need && image && image_bytes >= need) {
        char *p = (char *)image;
        void *stripe = st_bytes ? p : nullptr;
        p += a256(st_bytes);
        if (kind == 1) {
            L->perm32 = (int32_t *)p; p += a256((size_t)K * 4);
            L->invperm32 = (int32_t *)p; p += a256((size_t)K * 4);
            L->perm16 = (uint16_t *)p; p += a256((size_t)K * 2);
            std::vector<uint16_t> p16(K);
            std::vector<int32_t> inv(K);
            for (int k = 0; k < K; k++) { p16[k] = (uint16_t)perm[k]; inv[perm[k]] = k; }
            hipError_t e = hipMemcpyAsync(L->perm32, perm.data(), (size_t)K * 4, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(L->invperm32, inv.data(), (size_t)K * 4, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(L->perm16, p16.data(), (size_t)K * 2, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);   // the host vectors go out of scope
            if (e != hipSuccess) { delete L; return (int)e; }
        }
        if (stripe) {   // (kind 1: the rows are gathered through the permutation while the image is written -- no sorted copy of qweight)
            if (int rc = stripe_repack_launch((const uint32_t *)qweight, (const half_t *)scales, qzeros, nsets == 2 ? (const uint32_t *)qweight_up : nullptr,
                                              (const half_t *)scales_up, qzeros_up, stripe, K, N, bits, groupsize, s, kind == 1 ? L->perm32 : nullptr)) {
                delete L;
                return rc;
            }
            L->stripe = stripe; L->stripe_bytes = st_bytes;
        }
    } else if (need && image) {
        delete L;
        return GPTQ_E_WORKSPACE;
    }   // (no image given: the layer runs on the checkpoint-layout kernels)
    *out = L;
    return GPTQ_OK;
}

void gptq_layer_destroy(gptq_layer_t *layer) { delete layer; }

/* what the layer was classified as / owns (tests, engines that drive the stripe kernels directly) */
int gptq_layer_kind(const gptq_layer_t *layer) { return layer ? layer->kind : GPTQ_E_NULL; }
int gptq_layer_stripe_image(const gptq_layer_t *layer, const void **stripe, size_t *stripe_bytes, const uint16_t **perm16) {
    if (!layer) return GPTQ_E_NULL;
    if (stripe) *stripe = layer->stripe;
    if (stripe_bytes) *stripe_bytes = layer->stripe_bytes;
    if (perm16) *perm16 = layer->perm16;
    return GPTQ_OK;
}
/* regular act-order layer: original k -> position in the group-sorted order of its image (int32 [K]; NULL for other layers).  What a PRODUCER of
 * this layer's input stores through (gptq_stripe_matvec_perm_out_f16, gptq_decode_attn_batch_f16 out_perm) so that the layer needs no gather. */
int gptq_layer_inverse_perm(const gptq_layer_t *layer, const int32_t **invperm32) {
    if (!layer || !invperm32) return GPTQ_E_NULL;
    *invperm32 = layer->kind == 1 ? layer->invperm32 : nullptr;
    return GPTQ_OK;
}

/* Memory mode: the caller is about to FREE the checkpoint buffers (the stripe16 image is a bijection of them: one copy of the packed
 * weights per layer instead of two).  Only for layers whose every decode / small-batch route runs on the image: trivial or regular
 * act-order g_idx, an image present (else GPTQ_E_VARIANT and nothing changes).  Afterwards the routes that read the checkpoint layout
 * (prefill, fall-backs) first unpack it from the image into `scratch` -- gptq_layer_scratch_bytes() accounts for that. */
static size_t layer_unpacked_bytes(const gptq_layer &L) {
    const size_t G = L.groupsize >= L.K ? 1 : (size_t)L.K / L.groupsize;
    return (size_t)L.nsets * (a256((size_t)(L.K / 32 * L.bits) * L.N * 4) + a256(G * L.N * 2) + a256(G * (size_t)(L.N / 32 * L.bits) * 4));
}
int gptq_layer_release_checkpoint(gptq_layer_t *layer) {
    if (!layer) return GPTQ_E_NULL;
    if (layer->kind == 2 || !layer->stripe) return GPTQ_E_VARIANT;
    layer->released = true;
    // (a regular act-order layer keeps borrowing g_idx -- K ints; qweight / scales / qzeros come back out of the image on demand)
    for (int i = 0; i < 2; i++) layer->qw[i] = nullptr, layer->sc[i] = nullptr, layer->qz[i] = nullptr;
    return GPTQ_OK;
}
/* the checkpoint buffers of weight set `set` reproduced from the image (bit-exact): for state_dict() of a released layer */
int gptq_layer_unpack_checkpoint(const gptq_layer_t *layer, int set, int32_t *qweight, void *scales, int32_t *qzeros, gptq_stream_t stream) {
    if (!layer || !qweight || !scales || !qzeros) return GPTQ_E_NULL;
    if (!layer->stripe || layer->kind == 2) return GPTQ_E_VARIANT;
    return stripe_unpack_launch(layer->stripe, layer->K, layer->N, layer->bits, layer->groupsize, layer->nsets, set, (uint32_t *)qweight, (half_t *)scales,
                                qzeros, (hipStream_t)stream, layer->kind == 1 ? layer->invperm32 : nullptr);
}

// rows of x the decode kernel (its 4x4x4 row groups) takes before the 16-row MFMA tiles do: 8 while ONE round of workgroups covers N, else 4
//  -- and 8 for a wide gate | up pair the C-stripes kernel serves (round 5, stripe_kernel.inc stripe_launch_c: more than 512 stripes,
// at most five row blocks per wave, not 2-bit).  A/B knobs (read once): GPTQ_DECODE_ROWS_WIDE = rows on wider single layers, GPTQ_DECODE_ROWS_PAIR =
// rows of a wide gate/up pair.
static int decode_rows_max(int K, int N, int nsets, int bits, int groupsize) {
    static const int wide = [] { const char *e = getenv("GPTQ_DECODE_ROWS_WIDE"); return e ? atoi(e) : 4; }();
    static const int pair = [] { const char *e = getenv("GPTQ_DECODE_ROWS_PAIR"); return e ? atoi(e) : -1; }();
    static const int mf8 = [] { const char *e = getenv("GPTQ_DECODE_MF8"); return e ? atoi(e) : 1; }();
    const int bk = bits == 8 ? 64 : 128;     // k per row block of the image (stripe_unpack.inc BK)
    // round 6: 5 .. 8 rows through the 16x16x16 inner product of the decode kernel (stripe_kernel.inc, MF) cost what four rows cost -- every
    // shape it serves keeps eight rows in the decode launch: groups that span a row block, not 2-bit, eight rows of x in LDS (K <= 9216; at
    // most five row blocks per wave when a workgroup walks several stripes), or a single set on a K up to 12288 (x in two halves)
    if (mf8 && bits != 2) {
        const int gq = stripe_gq_shift(K, N, bits, groupsize);
        const int nu = (K / bk + 7) / 8;
        if ((gq == -1 || gq >= 2) && ((N / 16 <= 256 && nu * 8 * bk <= 9216) || (N / 16 > 256 && nu <= 5) || (nsets == 1 && nu * 8 * bk > 9216 && nu * 8 * bk <= 12288)))
            return 8;
    }
    if (N <= 4608) return 8;
    if (nsets != 2) return wide;
    if (pair >= 0) return pair;
    return (bits != 2 && N / 16 > 512 && K <= 5 * 8 * bk) ? 8 : 4;
}

// round 6: 9 .. 16 rows in the decode launch (stripe_gemvc_kernel, MF with sixteen distinct A rows): K <= 4096, groups that span a row block, not
// 2-bit; the launch itself declines when the tables do not fit LDS next to sixteen rows of x.  Returns a mask: 1 = launches with the fused norm,
// 2 = plain launches.  Measured (profiles/r6h_mf16/, us per launch at 16 rows, this kernel / 16-row tiles): gate | up pair plain 13.5 / 15.3 -- but
// with the norm fused 17.3 / 17.6 (tiles + the stand-alone norm launch): every one of the 230 workgroups normalises all 16 x 4096 elements itself,
// ~330 VALU instructions per thread, more than the launch boundary of the 16-workgroup norm kernel costs; qkv plain 9.4 / 9.5, fused norm 13.2 / 12.1;
// o_proj plain 6.5 / 5.4.  So: the pair takes this kernel PLAIN behind the norm launch (16.0 against 17.6), single sets stay on the tiles.
// GPTQ_DECODE_MF16 = mask pins it for every shape (A/B).
static int decode_rows_mf16(int K, int N, int nsets, int bits, int groupsize) {
    static const int mode = [] { const char *e = getenv("GPTQ_DECODE_MF16"); return e ? atoi(e) : -1; }();
    static const int mf8 = [] { const char *e = getenv("GPTQ_DECODE_MF8"); return e ? atoi(e) : 1; }();
    if (!mode || !mf8 || bits == 2) return 0;
    const int bk = bits == 8 ? 64 : 128;
    const int gq = stripe_gq_shift(K, N, bits, groupsize);
    const int nu = (K / bk + 7) / 8;
    if (!(gq == -1 || gq >= 2) || nu * 8 * bk > 4096 || (N / 16 > 256 && nu > 5) || N / 16 > 1024) return 0;
    if (mode > 0) return mode;
    return nsets == 2 ? 2 : 0;
}

// Round 5: 129 .. gptq_set_stripe_gemm_max_rows() rows -- the fused tile GEMM on the image, or dequantise + the tile GEMM of gemm8.hip?  Once gemm8 gave
// every XCD the same number of tiles (gemm8.hip, round 5) the dense route became the faster OWN kernel where the image route's 128 x 128 tiles no
// longer fit the chip at once (two workgroups per CU = 512 tiles; beyond that its time steps up: 4096 x 12288 at 640 / 768 rows 84.6 / 118.9 us
// against 103.7 / 105.9) AND the product is long enough to pay for the dequantise pass (2.5 bytes per weight against 2 M flops: the ratio depends on
// M alone).  Measured on the four LLaMA-7B shapes, 192 .. 2560 rows (profiles/r5e_gemm8_tile/image_vs_dense_192_2560_rows.txt): single sets -- dense
// wins from 768 rows on 4096 x 12288 / 4096 x 11008 (0.84-0.96 of the image route's time; tie at 2048), never on the N = 4096 shapes (at most 512
// tiles up to 2048 rows); gate | up pair (A fragments shared by both sets: the image kernel's best case) -- dense wins from 1280 rows (0.85-0.96).
// Act-order layers keep the image route (their dense route rebuilds the checkpoint order first; not measured).
static bool image_gemm_wanted(int M, int K, int N, int nsets, int kind) {
    if (M > g_stripe_gemm_max_rows.load()) return false;
    if (kind != 0 || K % 128 != 0 || !own_dense_route()) return true;   // the alternative would not be the tile GEMM
    const long tiles = (long)((M + 127) / 128) * (((long)N * nsets + 127) / 128);
    return tiles <= 512 || M <= (nsets == 2 ? 1152 : 640);
}

/* persistent workspace every forward takes: [split-K words, zero on first use and left zero][scratch of the 16-row MFMA tiles] */
size_t gptq_layer_workspace_bytes(void) { return WS_BYTES + STRIPE_MM_WS_BYTES; }

// a released layer whose dense route needs no checkpoint layout: trivial g_idx, an image, a product the tile GEMM of gemm8.hip takes
static bool layer_dense_from_image(const gptq_layer &L, int M) {
    return L.released && L.kind == 0 && L.stripe && L.K % 128 == 0 && own_dense_route() && (L.nsets == 2 || !L.bias || aligned(L.bias, 8));
}

/* transient scratch that gives forward(M) its fast route: the gathered x of an act-order batch, the per-call dequantised weight */
size_t gptq_layer_scratch_bytes(const gptq_layer_t *layer, int M) {
    if (!layer || M <= 0) return 0;
    // the fused tile GEMM on the image (129 .. gptq_set_stripe_gemm_max_rows() rows) needs no scratch beyond the gather of x for an
    // act-order layer: no K * N dequantised copy per call on the prompt path.  Should that kernel decline at launch, the ladder goes on
    // with what it was given (library-free kernels, or GPTQ_E_WORKSPACE for a released layer).
    if (gptq_layer_route_for(layer, M) == GPTQ_ROUTE_STRIPE_GEMM) return layer->kind == 1 ? a256((size_t)M * layer->K * 2) : 0;
    size_t unpack = (layer->released && M > 1) ? layer_unpacked_bytes(*layer) : 0;   // (M == 1 always runs on the image)
    if (M >= LAYER_PREFILL_MIN_M && (M > LAYER_STRIPE_MM_MAX_M || !layer->stripe)) {
        // round 5: a released layer with a trivial g_idx feeds the tile GEMM straight from its image (stripe_dequant_t_kernel): no checkpoint
        // layout is rebuilt.  Should that GEMM decline at launch the call returns GPTQ_E_WORKSPACE: gptq_layer_fallback_scratch_bytes() is the retry.
        if (layer_dense_from_image(*layer, M)) unpack = 0;
        return unpack + gptq_prefill_workspace_bytes(M, layer->K, layer->N, layer->nsets);
    }
    return unpack + (layer->kind == 1 && M > 1 ? a256((size_t)M * layer->K * 2) : 0);
}

/* the scratch with which forward(M) cannot fail for lack of memory: every fall-back's needs (the rebuilt checkpoint layout of a released layer,
 * the dense route's workspace, the gathered x of an act-order batch).  What a caller retries with after GPTQ_E_WORKSPACE. */
size_t gptq_layer_fallback_scratch_bytes(const gptq_layer_t *layer, int M) {
    if (!layer || M <= 0) return 0;
    return (layer->released ? layer_unpacked_bytes(*layer) : 0) + gptq_prefill_workspace_bytes(M, layer->K, layer->N, layer->nsets) +
           a256((size_t)M * layer->K * 2);
}

static int layer_forward_checkpoint(const gptq_layer &L, const void *x, int64_t ldx, void *y, int64_t ldy, int M, void *workspace, void *scratch,
                                    size_t scratch_bytes, gptq_stream_t stream);

// The table of gptq_layer_forward as a host-only query (no launch): which kernel family a batch of M rows takes for a layer of this
// shape, given that the caller supplies gptq_layer_scratch_bytes() of scratch.  It mirrors the ladder below; a kernel may still decline at
// launch (LDS limits of the row groups ...) and hand the batch to the next rung.
int gptq_layer_route_for_shape(int M, int K, int N, int bits, int groupsize, int nsets, int kind, int has_image) {
    if (M <= 0 || K <= 0 || N <= 0 || nsets < 1 || nsets > 2 || kind < 0 || kind > 2) return GPTQ_E_SHAPE;
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return GPTQ_E_BITS;
    const int gq = stripe_gq_shift(K, N, bits, groupsize);
    const bool image = has_image && gq != -2 && kind != 2;
    int rows_max = decode_rows_max(K, N, nsets, bits, groupsize);
    if (M > 8 && M <= 16 && (decode_rows_mf16(K, N, nsets, bits, groupsize) & 2)) rows_max = std::max(rows_max, 16);   // round 6: sixteen A rows (the pair)
    if (image && (kind == 0 || M == 1) && M <= rows_max && (M <= 4 || K <= 9216 || (K <= 12288 && nsets == 1))) return GPTQ_ROUTE_STRIPE_DECODE;
    if (image && M > 1) {
        if (M <= rows_max && kind == 1 && (M <= 4 || K <= 9216 || (K <= 12288 && nsets == 1))) return GPTQ_ROUTE_STRIPE_DECODE;       // after one gather of x
        if (M <= LAYER_STRIPE_MM_MAX_M) return GPTQ_ROUTE_STRIPE_TILES;
        if (image_gemm_wanted(M, K, N, nsets, kind) && bits != 2 && (gq == -1 || gq >= 2)) return GPTQ_ROUTE_STRIPE_GEMM;       // groups of at least a row block
    }
    if (M >= LAYER_PREFILL_MIN_M)
        return (own_dense_route() && K % 128 == 0) ? GPTQ_ROUTE_DENSE_TILE_GEMM : GPTQ_ROUTE_DENSE_LIBRARY;
    return GPTQ_ROUTE_CHECKPOINT_KERNELS;
}

int gptq_layer_route_for(const gptq_layer_t *layer, int M) {
    if (!layer) return GPTQ_E_NULL;
    return gptq_layer_route_for_shape(M, layer->K, layer->N, layer->bits, layer->groupsize, layer->nsets, layer->kind, layer->stripe != nullptr);
}

int gptq_layer_forward(const gptq_layer_t *layer, const void *x, int64_t ldx, void *y, int64_t ldy, int M, void *workspace, size_t workspace_bytes,
                       void *scratch, size_t scratch_bytes, gptq_stream_t stream) {
    if (!layer) return GPTQ_E_NULL;
    const gptq_layer &L = *layer;
    if (M < 0 || ldx < L.K || ldy < L.N) return GPTQ_E_SHAPE;
    if (M == 0) return GPTQ_OK;
    if (!x || !y) return GPTQ_E_NULL;
    if (!aligned(x, 16) || ldx % 8 != 0 || !aligned(y, 8) || ldy % 4 != 0) return GPTQ_E_ALIGN;
    if (!workspace || !aligned(workspace, 256) || workspace_bytes < gptq_layer_workspace_bytes()) return GPTQ_E_WORKSPACE;
    void *mm_ws = (char *)workspace + WS_BYTES;
    const int K = L.K, N = L.N, bits = L.bits, gs = L.groupsize, ns = L.nsets;
    int rows_max = decode_rows_max(K, N, ns, bits, gs);    // row groups only while ONE round of workgroups covers N
    if (M > 8 && M <= 16 && (decode_rows_mf16(K, N, ns, bits, gs) & 2)) rows_max = std::max(rows_max, 16);   // round 6: sixteen A rows (the pair)
    // ---- 1. decode and small batches on the stripe16 image ----
    if (L.stripe && L.kind == 0) {
        if (M <= rows_max) {
            const int rc = stripe_matvec(x, ldx, L.stripe, L.stripe_bytes, L.bias, y, ldy, nullptr, M, K, N, bits, gs, ns, nullptr, 0.f, nullptr, stream);
            if (rc != GPTQ_E_VARIANT) return rc;
        }
        // ... 128 rows: 16-row tiles; above: the fused tile GEMM, unless the dense route is the faster own kernel for this batch (image_gemm_wanted)
        // AND the caller brought the scratch it needs (gptq_layer_scratch_bytes says so; without it the batch stays on the image)
        bool on_image = M <= LAYER_STRIPE_MM_MAX_M || M <= g_stripe_gemm_max_rows.load();
        if (on_image && M > LAYER_STRIPE_MM_MAX_M && !image_gemm_wanted(M, K, N, ns, 0)) {
            const size_t need = !L.released ? gptq_prefill_workspace_bytes(M, K, N, ns)
                                            : layer_dense_from_image(L, M) ? a256((size_t)K * N * ns * 2) : layer_unpacked_bytes(L) + gptq_prefill_workspace_bytes(M, K, N, ns);
            on_image = !(scratch && aligned(scratch, 256) && scratch_bytes >= need);
        }
        if (M > 4 && on_image) {
            const int rc = stripe_matvec(x, ldx, L.stripe, L.stripe_bytes, L.bias, y, ldy, nullptr, M, K, N, bits, gs, ns, nullptr, 0.f, nullptr, stream, mm_ws,
                                         STRIPE_MM_WS_BYTES, true);
            if (rc != GPTQ_E_VARIANT) return rc;
        }
    }
    if (L.stripe && L.kind == 1) {
        if (M == 1) {   // the decode kernel gathers x through the permutation itself
            const int rc = stripe_matvec(x, ldx, L.stripe, L.stripe_bytes, L.bias, y, ldy, nullptr, 1, K, N, bits, gs, ns, nullptr, 0.f, L.perm16, stream);
            if (rc != GPTQ_E_VARIANT) return rc;
        } else if (M <= std::max(LAYER_STRIPE_MM_MAX_M, g_stripe_gemm_max_rows.load()) && scratch && aligned(scratch, 16) && scratch_bytes >= (size_t)M * K * 2) {
            // batches: ONE gather of x, then the trivial-g_idx kernels on the image of the group-sorted rows
            if (int rc = gather_cols_launch((const half_t *)x, ldx, L.perm32, (half_t *)scratch, K, M, K, (hipStream_t)stream)) return rc;
            if (M <= rows_max) {
                const int rc = stripe_matvec(scratch, K, L.stripe, L.stripe_bytes, L.bias, y, ldy, nullptr, M, K, N, bits, gs, ns, nullptr, 0.f, nullptr, stream);
                if (rc != GPTQ_E_VARIANT) return rc;
            }
            const int rc = stripe_matvec(scratch, K, L.stripe, L.stripe_bytes, L.bias, y, ldy, nullptr, M, K, N, bits, gs, ns, nullptr, 0.f, nullptr, stream, mm_ws,
                                         STRIPE_MM_WS_BYTES, true);
            if (rc != GPTQ_E_VARIANT) return rc;
        }
    }
    if (M >= LAYER_PREFILL_MIN_M && layer_dense_from_image(L, M) && scratch && aligned(scratch, 256) && scratch_bytes >= a256((size_t)K * N * ns * 2)) {
        // memory mode, prompt sizes: image -> W^T in ONE pass (bit-identical to the weight the two-pass route dequantises), then the tile GEMM
        half_t *Wt = (half_t *)scratch;
        int rc = stripe_dequant_t_launch(L.stripe, K, N, bits, gs, ns, Wt, K, (hipStream_t)stream);
        if (rc == 0) rc = gemm8_dense_f16((const half_t *)x, ldx, Wt, K, ns == 2 ? nullptr : (const half_t *)L.bias, (half_t *)y, ldy, M, K, N, ns == 2,
                                          (hipStream_t)stream);
        if (rc != GPTQ_E_VARIANT) return rc;
    }
    gptq_layer Lr;   // a released layer: the checkpoint layout is rebuilt from the image into the head of `scratch` for the routes below
    const gptq_layer *Lp = &L;
    if (L.released) {
        const size_t ub = layer_unpacked_bytes(L);
        if (!scratch || !aligned(scratch, 256) || scratch_bytes < ub) return GPTQ_E_WORKSPACE;
        Lr = L;
        char *p = (char *)scratch;
        const size_t G = gs >= K ? 1 : (size_t)K / gs;
        for (int i = 0; i < ns; i++) {
            int32_t *qw_i = (int32_t *)p; p += a256((size_t)(K / 32 * bits) * N * 4);
            void *sc_i = p; p += a256(G * N * 2);
            int32_t *qz_i = (int32_t *)p; p += a256(G * (size_t)(N / 32 * bits) * 4);
            if (int rc = stripe_unpack_launch(L.stripe, K, N, bits, gs, ns, i, (uint32_t *)qw_i, (half_t *)sc_i, qz_i, (hipStream_t)stream,
                                              L.kind == 1 ? L.invperm32 : nullptr))
                return rc;
            Lr.qw[i] = qw_i; Lr.sc[i] = sc_i; Lr.qz[i] = qz_i;
        }
        scratch = p;
        scratch_bytes -= ub;
        Lp = &Lr;
    }
    return layer_forward_checkpoint(*Lp, x, ldx, y, ldy, M, workspace, scratch, scratch_bytes, stream);
}

// routes 2 and 3 of gptq_layer_forward: everything that reads the checkpoint layout
static int layer_forward_checkpoint(const gptq_layer &L, const void *x, int64_t ldx, void *y, int64_t ldy, int M, void *workspace, void *scratch,
                                    size_t scratch_bytes, gptq_stream_t stream) {
    const int K = L.K, N = L.N, bits = L.bits, gs = L.groupsize, ns = L.nsets;
    // ---- 2. the dense route: dequantise once per call + tile GEMM (any width, any g_idx; no gather of x) ----
    if (M >= LAYER_PREFILL_MIN_M && scratch && aligned(scratch, 256) && scratch_bytes >= gptq_prefill_workspace_bytes(M, K, N, ns)) {
        const int rc = ns == 2 ? gptq_prefill_fused_mlp_f16(x, ldx, L.qw[0], L.sc[0], L.qz[0], L.gi[0], L.qw[1], L.sc[1], L.qz[1], L.gi[1], y, ldy, M, K, N, bits, gs,
                                                            scratch, scratch_bytes, stream)
                               : gptq_prefill_matmul_f16(x, ldx, L.qw[0], L.sc[0], L.qz[0], L.gi[0], L.bias, y, ldy, M, K, N, bits, gs, scratch, scratch_bytes, stream);
        if (rc != GPTQ_E_LIBRARY && rc != GPTQ_E_VARIANT) return rc;   // (no hipBLASLt for a shape the tile GEMM does not serve: own kernels below)
        static std::atomic<bool> warned{false};
        if (rc == GPTQ_E_LIBRARY && !warned.exchange(true))
            fprintf(stderr, "libgptq_mi355x: %s -- falling back to the library-free kernels of the C ABI (slower at these sizes, same results)\n",
                    gptq_strerror(rc));
    }
    // ---- 3. kernels on the checkpoint layout (split-K rowwave / stream / tile GEMM; generic kernels for an irregular g_idx) ----
    Problem q = make_problem(x, ldx, L.qw[0], L.sc[0], L.qz[0], L.gi[0], L.bias, y, ldy, M, K, N, bits, gs, workspace, WS_BYTES);
    if (ns == 2) {
        q.fused2 = true;
        q.qw[1] = L.qw[1]; q.sc[1] = L.sc[1]; q.qz[1] = L.qz[1]; q.gi[1] = L.gi[1];
    }
    // (a regular act-order layer that gets here -- no scratch for the gather, a kernel declined -- runs the generic g_idx kernels on the
    // checkpoint rows: round 3's group-sorted qweight copy, one more qweight per layer kept for this corner, is gone)
    if (int rc = validate(q)) return rc;
    return run_auto(q, (hipStream_t)stream);
}

// ---- round 5: the operator of the batched decode engine -- y = residual + layer(rmsnorm(x)) for 1 .. 128 rows ----
// One decoder layer is four of these (qkv with the input norm, o_proj with the residual, gate/up with the post-attention norm, down_proj
// with the residual: fused_attn.py:117-161, fused_mlp.py:203-218 between HF's norms and adds).  The decode kernel takes norm AND residual
// into its launch for up to 4 (8 on one-round shapes) rows; the 16-row tiles take the residual into their epilogue and get the norm from a
// stand-alone launch into `scratch`; everything else is gptq_layer_forward + an add.
size_t gptq_layer_decode_scratch_bytes(const gptq_layer_t *layer, int M) {
    if (!layer || M <= 0) return 0;
    return a256((size_t)M * layer->K * 2) + gptq_layer_scratch_bytes(layer, M);
}

static int layer_decode(const gptq_layer_t *layer, const void *x, int64_t ldx, void *y, int64_t ldy, int M, const void *norm_weight, float norm_eps,
                        const void *residual, int64_t ldr, void *workspace, size_t workspace_bytes, void *scratch, size_t scratch_bytes,
                        gptq_stream_t stream, const NextNorm *nn) {
    if (!layer) return GPTQ_E_NULL;
    const gptq_layer &L = *layer;
    if (M < 0 || ldx < L.K || ldy < L.N || (residual && ldr < L.N)) return GPTQ_E_SHAPE;
    if (M == 0) return GPTQ_OK;
    if (M > LAYER_STRIPE_MM_MAX_M) return GPTQ_E_VARIANT;
    if (!x || !y) return GPTQ_E_NULL;
    if (L.nsets == 2 && residual) return GPTQ_E_VARIANT;            // (the SiLU pair has no residual in a LLaMA block)
    if (!aligned(x, 16) || ldx % 8 != 0 || !aligned(y, 16) || ldy % 8 != 0 || (norm_weight && !aligned(norm_weight, 16)) ||
        (residual && (!aligned(residual, 16) || ldr % 8 != 0)))
        return GPTQ_E_ALIGN;
    if (!workspace || !aligned(workspace, 256) || workspace_bytes < gptq_layer_workspace_bytes()) return GPTQ_E_WORKSPACE;
    if (scratch && !aligned(scratch, 256)) return GPTQ_E_ALIGN;
    void *mm_ws = (char *)workspace + WS_BYTES;
    const int K = L.K, N = L.N, bits = L.bits, gs = L.groupsize, ns = L.nsets;
    const int rows_max = decode_rows_max(K, N, ns, bits, gs);
    const bool add_fused = !(L.bias && residual);                   // one add slot per launch: bias OR residual
    const void *add = residual && add_fused ? residual : L.bias;
    const int64_t ldb = residual && add_fused ? ldr : 0;
    auto finish = [&](int rc) {                                     // the residual of a launch that carried the bias
        if (rc == 0 && residual && !add_fused) return add_rows_launch((half_t *)y, ldy, (const half_t *)residual, ldr, M, N, (hipStream_t)stream);
        return rc;
    };
    const bool image = L.stripe != nullptr && L.kind != 2;
    const int mf16 = M > 8 && M <= 16 ? decode_rows_mf16(K, N, ns, bits, gs) : 0;
    // 1. everything in the decode kernel's launch
    if (image && (M <= rows_max || (norm_weight ? (mf16 & 1) : (mf16 & 2))) && (L.kind == 0 || M == 1)) {
        const int rc = stripe_matvec(x, ldx, L.stripe, L.stripe_bytes, add, y, ldy, nullptr, M, K, N, bits, gs, ns, norm_weight, norm_eps,
                                     L.kind == 1 ? L.perm16 : nullptr, stream, nullptr, 0, false, ldb);
        if (rc != GPTQ_E_VARIANT) return finish(rc);
    }
    // 2. the norm on its own, into scratch
    const void *xin = x;
    int64_t ldin = ldx;
    char *sp = (char *)scratch;
    size_t left = scratch_bytes;
    if (norm_weight) {
        const size_t nb = a256((size_t)M * K * 2);
        if (!scratch || left < nb) return GPTQ_E_WORKSPACE;
        if ((size_t)K * 2 > 65536) return GPTQ_E_NORM_WIDTH;
        if (int rc = rmsnorm_launch((const half_t *)x, ldx, (const half_t *)norm_weight, (half_t *)sp, K, M, K, norm_eps, (hipStream_t)stream)) return rc;
        xin = sp; ldin = K;
        sp += nb; left -= nb;
    }
    if (image && L.kind == 0) {
        if ((M <= rows_max || (mf16 & 2)) && norm_weight) {   // (the decode kernel declined WITH the norm -- LDS -- but may take the rows without it)
            const int rc = stripe_matvec(xin, ldin, L.stripe, L.stripe_bytes, add, y, ldy, nullptr, M, K, N, bits, gs, ns, nullptr, 0.f, nullptr, stream, nullptr, 0,
                                         false, ldb);
            if (rc != GPTQ_E_VARIANT) return finish(rc);
        }
        if (M > 4) {                           // 16-row MFMA tiles, the residual in their epilogue (and, with K slices, the next norm: nn)
            const int rc = stripe_matvec(xin, ldin, L.stripe, L.stripe_bytes, add, y, ldy, nullptr, M, K, N, bits, gs, ns, nullptr, 0.f, nullptr, stream, mm_ws,
                                         STRIPE_MM_WS_BYTES, true, ldb, nullptr, nullptr, add_fused ? nn : nullptr);
            if (rc != GPTQ_E_VARIANT) return finish(rc);
        }
    }
    // 3. whatever gptq_layer_forward picks (act-order batches after a gather, layers without an image ...), then the add
    const int rc = gptq_layer_forward(layer, xin, ldin, y, ldy, M, workspace, workspace_bytes, left ? sp : nullptr, left, stream);
    if (rc == 0 && residual) return add_rows_launch((half_t *)y, ldy, (const half_t *)residual, ldr, M, N, (hipStream_t)stream);
    return rc;
}

int gptq_layer_decode_f16(const gptq_layer_t *layer, const void *x, int64_t ldx, void *y, int64_t ldy, int M, const void *norm_weight, float norm_eps,
                          const void *residual, int64_t ldr, void *workspace, size_t workspace_bytes, void *scratch, size_t scratch_bytes,
                          gptq_stream_t stream) {
    return layer_decode(layer, x, ldx, y, ldy, M, norm_weight, norm_eps, residual, ldr, workspace, workspace_bytes, scratch, scratch_bytes, stream, nullptr);
}

/* round 6: gptq_layer_decode_f16 + "leave the NEXT RMSNorm's rows behind if that is free": when this batch's route combines K slices in a launch of
 * its own (9 .. 16 rows of a long-K layer: LLaMA's down_proj), that launch owns whole rows of y and also writes h[M][ldh] = rmsnorm(y) * next_norm_weight
 * -- bit for bit what gptq_rmsnorm_f16 would write -- and *h_written = 1: the consumer (the next block's qkv) runs without its norm, one launch less
 * per decoder block.  Every other route leaves h untouched and *h_written = 0 (the consumer keeps its fused / stand-alone norm). */
int gptq_layer_decode_next_norm_f16(const gptq_layer_t *layer, const void *x, int64_t ldx, void *y, int64_t ldy, int M, const void *norm_weight, float norm_eps,
                                    const void *residual, int64_t ldr, const void *next_norm_weight, float next_norm_eps, void *h, int64_t ldh, int *h_written,
                                    void *workspace, size_t workspace_bytes, void *scratch, size_t scratch_bytes, gptq_stream_t stream) {
    if (!h_written) return GPTQ_E_NULL;
    *h_written = 0;
    if (!layer) return GPTQ_E_NULL;
    static const int on = [] { const char *e = getenv("GPTQ_NEXT_NORM"); return e ? atoi(e) : 1; }();
    if (!on || !next_norm_weight || !h || ldh < layer->N || M < 1 || M > 16)
        return layer_decode(layer, x, ldx, y, ldy, M, norm_weight, norm_eps, residual, ldr, workspace, workspace_bytes, scratch, scratch_bytes, stream, nullptr);
    NextNorm nn{next_norm_weight, next_norm_eps, h, ldh, h_written};
    return layer_decode(layer, x, ldx, y, ldy, M, norm_weight, norm_eps, residual, ldr, workspace, workspace_bytes, scratch, scratch_bytes, stream, &nn);
}

// ---- round 6: o_proj of a decode step whose attention left split records instead of an fp16 row ----
// y = residual + layer(x), x = the merge of the records gptq_decode_attn_split_f16 wrote for this row (attn_split.h): the decode kernel stages x
// from them, so the attention launch needs no cross-workgroup merge of its own.  One row (a batch-1 decode step), a layer with a trivial g_idx
// and a stripe16 image, K = heads x 128; anything else: GPTQ_E_VARIANT (the caller then runs gptq_decode_attn_batch_f16 + gptq_layer_decode_f16).
int gptq_layer_decode_attn_supported(const gptq_layer_t *layer, int batch, int heads, int head_dim) {
    if (!layer) return GPTQ_E_NULL;
    const gptq_layer &L = *layer;
    if (batch != 1 || head_dim != 128 || heads <= 0 || L.K != heads * 128) return 0;
    if (!L.stripe || L.kind != 0 || L.nsets != 1) return 0;
    if (stripe_gq_shift(L.K, L.N, L.bits, L.groupsize) == -2) return 0;
    const int bk = L.bits == 8 ? 64 : 128;
    if ((L.K / bk + 7) / 8 > stripe_max_nu(L.bits)) return 0;
    return 1;
}

int gptq_layer_decode_attn_f16(const gptq_layer_t *layer, const void *attn_workspace, size_t attn_workspace_bytes, const int64_t *positions, int batch,
                               int heads, int head_dim, int t_max, int tokens_per_split, void *y, int64_t ldy, const void *residual, int64_t ldr,
                               gptq_stream_t stream) {
    if (!layer || !attn_workspace || !positions || !y) return GPTQ_E_NULL;
    const gptq_layer &L = *layer;
    if (batch <= 0 || heads <= 0 || head_dim != 128 || t_max <= 0 || ldy < L.N || (residual && ldr < L.N)) return GPTQ_E_SHAPE;
    if (gptq_layer_decode_attn_supported(layer, batch, heads, head_dim) != 1) return GPTQ_E_VARIANT;
    if (L.bias && residual) return GPTQ_E_VARIANT;                  // one add slot per launch
    if (!aligned(attn_workspace, 16) || !aligned(y, 16) || ldy % 8 != 0 || (residual && (!aligned(residual, 16) || ldr % 8 != 0))) return GPTQ_E_ALIGN;
    if (attn_workspace_bytes < decode_attn_ws_bytes(heads, t_max, batch)) return GPTQ_E_WORKSPACE;
    const int S = decode_attn_grid_splits(heads, t_max, batch);
    AttnMerge am{};
    am.o16 = (const half_t *)attn_workspace;
    am.md = (const float *)(am.o16 + (size_t)batch * S * heads * 128);
    am.pos = positions;
    am.S = S;
    am.tps = tokens_per_split > 0 ? tokens_per_split : decode_attn_tps(true);
    am.heads = heads;
    am.t_max = t_max;
    const void *add = residual ? residual : L.bias;
    return stripe_matvec(attn_workspace, L.K, L.stripe, L.stripe_bytes, add, y, ldy, nullptr, 1, L.K, L.N, L.bits, L.groupsize, 1, nullptr, 0.f, nullptr, stream,
                         nullptr, 0, false, residual ? ldr : 0, nullptr, &am);
}

// ---- GPTQ solver: the sequential loop of one column block (gptq_solver.hip) ----
int gptq_solver_block_f32(const float *W, int64_t ldw, const float *Hinv, int64_t ldh, int rows, int cols, int i1, int count, int groupsize,
                          int maxq, const float *scale, const float *zero, int64_t ldg, float *Q, int64_t ldq, float *Err, int64_t lde,
                          float *loss_rows, gptq_stream_t stream) {
    if (!W || !Hinv || !scale || !zero || !Q || !Err || !loss_rows) return GPTQ_E_NULL;
    if (rows <= 0 || cols <= 0 || i1 < 0 || count <= 0 || i1 + count > cols || groupsize <= 0 || maxq <= 0) return GPTQ_E_SHAPE;
    if (count > 128) return GPTQ_E_VARIANT;   // two columns per lane
    if (ldw < cols || ldh < cols || ldq < cols || lde < count || ldg < (cols + groupsize - 1) / groupsize) return GPTQ_E_SHAPE;
    return gptq_block_launch(W, ldw, Hinv, ldh, rows, i1, count, groupsize, maxq, scale, zero, ldg, Q, ldq, Err, lde, loss_rows,
                             (hipStream_t)stream);
}

}  // extern "C"
