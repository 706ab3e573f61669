// gptq_device.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define GPTQ_DEV static __device__ __forceinline__

GPTQ_DEV half2_t as_half2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
GPTQ_DEV uint32_t as_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

// Number of XCDs on MI355X; block b is observed to run on XCD b % 8 (speed only, never
// correctness).  Bijective remap that hands every XCD a contiguous range of logical ids so
// neighbouring tiles (which share 128-B lines / x) hit the same L2.
GPTQ_DEV int xcd_remap(int bid, int nwg) {
    constexpr int NXCD = 8;
    if (nwg < 2 * NXCD) return bid;
    int xcd = bid % NXCD, idx = bid / NXCD;
    int q = nwg / NXCD, r = nwg % NXCD;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---------------------------------------------------------------------------------------
// Field extraction.  A packed word holds KPW = 32/BITS consecutive k of ONE column.
// Unpack<BITS>::pairs() turns a word into NP = KPW/2 half2 values t[p] = {OFF + q_p,
// OFF + q_{p+NP}} using the fp16 "magic exponent" trick: OR-ing the field into the mantissa
// of a constant whose ulp equals the field's bit weight, so no int->float convert is needed.
// The constant offset OFF is removed exactly (fp16 subtract) or folded into the zero point.
// ---------------------------------------------------------------------------------------
template <int BITS>
struct Unpack;

template <>
struct Unpack<4> {
    static constexpr int KPW = 8, NP = 4;
    static constexpr float OFF = 64.0f;  // 0x5400 = 64.0, ulp 1/16: field at bits [7:4] -> +q
    GPTQ_DEV void pairs(uint32_t w, half2_t (&t)[NP]) {
        constexpr uint32_t MSK = 0x00F000F0u, MAG = 0x54005400u;
        t[0] = as_half2(((w << 4) & MSK) | MAG);  // fields 0,4
        t[1] = as_half2((w & MSK) | MAG);          // fields 1,5
        t[2] = as_half2(((w >> 4) & MSK) | MAG);  // fields 2,6
        t[3] = as_half2(((w >> 8) & MSK) | MAG);  // fields 3,7
    }
    // same with the mask in an SGPR and the magic in a VGPR: 3 shifts + 4 v_and_or_b32
    static constexpr uint32_t MSK_C = 0x00F000F0u, MAG_C = 0x54005400u;
    GPTQ_DEV void pairs_rc(uint32_t w, half2_t (&t)[NP], uint32_t msk, uint32_t mag) {
        t[0] = as_half2(((w << 4) & msk) | mag);
        t[1] = as_half2((w & msk) | mag);
        t[2] = as_half2(((w >> 4) & msk) | mag);
        t[3] = as_half2(((w >> 8) & msk) | mag);
    }
};

template <>
struct Unpack<2> {
    static constexpr int KPW = 16, NP = 8;
    static constexpr float OFF = 64.0f;  // field at bits [5:4] of 0x5400 -> +q
    GPTQ_DEV void pairs(uint32_t w, half2_t (&t)[NP]) {
        constexpr uint32_t MSK = 0x00300030u, MAG = 0x54005400u;
        t[0] = as_half2(((w << 4) & MSK) | MAG);   // fields 0,8
        t[1] = as_half2(((w << 2) & MSK) | MAG);   // 1,9
        t[2] = as_half2((w & MSK) | MAG);           // 2,10
        t[3] = as_half2(((w >> 2) & MSK) | MAG);   // 3,11
        t[4] = as_half2(((w >> 4) & MSK) | MAG);   // 4,12
        t[5] = as_half2(((w >> 6) & MSK) | MAG);   // 5,13
        t[6] = as_half2(((w >> 8) & MSK) | MAG);   // 6,14
        t[7] = as_half2(((w >> 10) & MSK) | MAG);  // 7,15
    }
    static constexpr uint32_t MSK_C = 0x00300030u, MAG_C = 0x54005400u;
    GPTQ_DEV void pairs_rc(uint32_t w, half2_t (&t)[NP], uint32_t msk, uint32_t mag) {
        t[0] = as_half2(((w << 4) & msk) | mag);
        t[1] = as_half2(((w << 2) & msk) | mag);
        t[2] = as_half2((w & msk) | mag);
        t[3] = as_half2(((w >> 2) & msk) | mag);
        t[4] = as_half2(((w >> 4) & msk) | mag);
        t[5] = as_half2(((w >> 6) & msk) | mag);
        t[6] = as_half2(((w >> 8) & msk) | mag);
        t[7] = as_half2(((w >> 10) & msk) | mag);
    }
};

template <>
struct Unpack<8> {
    static constexpr int KPW = 4, NP = 2;
    static constexpr float OFF = 256.0f;  // 0x5C00 = 256.0, ulp 1/4: byte at bits [9:2] -> +q
    GPTQ_DEV void pairs(uint32_t w, half2_t (&t)[NP]) {
        constexpr uint32_t MSK = 0x03FC03FCu, MAG = 0x5C005C00u;
        t[0] = as_half2(((w << 2) & MSK) | MAG);  // bytes 0,2
        t[1] = as_half2(((w >> 6) & MSK) | MAG);  // bytes 1,3
    }
    static constexpr uint32_t MSK_C = 0x03FC03FCu, MAG_C = 0x5C005C00u;
    GPTQ_DEV void pairs_rc(uint32_t w, half2_t (&t)[NP], uint32_t msk, uint32_t mag) {
        t[0] = as_half2(((w << 2) & msk) | mag);
        t[1] = as_half2(((w >> 6) & msk) | mag);
    }
};

// Staged-x order for one word-row of KPW values: position 2p <- field p, 2p+1 <- field p+NP.
template <int BITS>
GPTQ_DEV int staged_pos(int field) {
    constexpr int NP = Unpack<BITS>::NP;
    return (field < NP) ? 2 * field : 2 * (field - NP) + 1;
}

// Generic integer field j (0..31) of a 32-k block for one column; rows w[0..CH) are the
// BITS consecutive qweight rows of that block.  Used by the fallback kernels and by 3-bit.
template <int BITS>
GPTQ_DEV int field_of_block(const uint32_t *w, int j) {
    if constexpr (BITS == 3) {
        int bit = 3 * j, wi = bit >> 5, o = bit & 31;
        uint64_t lo = w[wi];
        uint64_t hi = (wi < 2) ? w[wi + 1] : 0;
        return (int)(((lo | (hi << 32)) >> o) & 7u);
    } else {
        constexpr int KPW = 32 / BITS;
        return (int)((w[j / KPW] >> (BITS * (j % KPW))) & ((1u << BITS) - 1u));
    }
}

// zero point (stored + 1, not re-masked) of column n from a qzeros row pointer.
template <int BITS>
GPTQ_DEV int zero_of(const int32_t *zrow, int n) {
    if constexpr (BITS == 3) {
        const uint32_t *p = (const uint32_t *)zrow + 3 * (n >> 5);
        int bit = 3 * (n & 31), wi = bit >> 5, o = bit & 31;
        uint64_t lo = p[wi];
        uint64_t hi = (wi < 2) ? p[wi + 1] : 0;
        return (int)(((lo | (hi << 32)) >> o) & 7u) + 1;
    } else {
        constexpr int KPW = 32 / BITS;
        uint32_t w = (uint32_t)zrow[n / KPW];
        return (int)((w >> (BITS * (n % KPW))) & ((1u << BITS) - 1u)) + 1;
    }
}

// sum over the xor butterfly offsets from, 2 from, ... 32.  __shfl_xor lowers to ds_bpermute_b32 (an LDS round trip, ~120 cycles per
// dependent step); the offsets inside a 16-lane row run as DPP row operations and 16 / 32 as gfx950 permlane swaps instead -- same
// operand pairs, bit-identical sums.  (from = 2, 4, 8 keep the shuffle for the in-row steps: the mirror patterns only equal the
// xor pairs when the finer steps ran first.)
template <int CTRL>
GPTQ_DEV float dpp_row_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
GPTQ_DEV float wave_sum_xor(float v, int from) {
    if (from == 1) {
        v += dpp_row_f32<0xB1>(v);    // quad_perm [1,0,3,2]
        v += dpp_row_f32<0x4E>(v);    // quad_perm [2,3,0,1]
        v += dpp_row_f32<0x141>(v);   // row_half_mirror
        v += dpp_row_f32<0x140>(v);   // row_mirror
    } else {
#pragma unroll
        for (int off = from; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
    }
    if (from <= 16) {
        const uint32_t u = __builtin_bit_cast(uint32_t, v);
        auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    }
    if (from <= 32) {
        const uint32_t u = __builtin_bit_cast(uint32_t, v);
        auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
    }
    return v;
}

// ---------------------------------------------------------------------------------------
// Split-K combine in ONE memory round trip, order-independent (bit-reproducible):
// every K-slice adds its partial sum as a biased fixed-point integer together with an arrival
// count in the top bits of the SAME 64-bit word, with one returning agent-scope atomic.  The
// slice whose returned value completes the count owns the total: it decodes it, stores zero
// back (the workspace is all-zero between launches) and writes the output.  No ticket, no
// second pass, no fence; integer addition makes the sum independent of arrival order.
//   word : [63:56] count (S <= 64) | [55:0] sum of (trunc(v * 2^24) + 2^49), |v| <= 2^24
//          64 slices add at most 64 * (2^49 + 2^48) = 1.5 * 2^55 < 2^56: the sum never carries into the count
//          (with S = 128 it would: 128 * 2^49 = 2^56 -- the arrival count would read S + 1 and no slice would own it).
// The fused gate/up kernel keeps TWO such words per column (gate, up): same range and resolution as the single
// combine -- no clamp on the partial sums (the reference accumulates in fp32 without range limits,
// fused_mlp.py:128-160).
// ---------------------------------------------------------------------------------------
typedef unsigned long long u64_t;
constexpr int SPLITK_MAX_SINGLE = 64;
constexpr int SPLITK_MAX_PAIR = 64;

GPTQ_DEV u64_t splitk_encode(float v) {
    const float c = fminf(fmaxf(v, -16777216.0f), 16777216.0f);               // |v| <= 2^24 (fp16 max is 65504)
    const long long fx = (long long)(c * 16777216.0f) + (1LL << 49);            // exact: power-of-two scale
    return (u64_t)fx + (1ULL << 56);
}
GPTQ_DEV float splitk_decode(u64_t now, int S) {
    const long long sum = (long long)(now & ((1ULL << 56) - 1)) - (long long)S * (1LL << 49);
    return (float)sum * (1.0f / 16777216.0f);
}

GPTQ_DEV bool splitk_add1(u64_t *word, float v, int S, float &total) {
    const u64_t add = splitk_encode(v);
    const u64_t old = __hip_atomic_fetch_add(word, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64_t now = old + add;
    if ((int)(now >> 56) != S) return false;
    __hip_atomic_store(word, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    total = splitk_decode(now, S);
    return true;
}

// words[0] = gate, words[1] = up.  Both atomics are in flight together; the slice that completes the count of word 0
// owns the column.  Atomics to different addresses may retire out of order, so in the rare case that another
// slice's addend to word 1 is still in flight the owner re-reads word 1 until its count is complete (the other
// slice issued that atomic before the one the owner has already seen: it needs nothing from the owner to land;
// the spin is bounded anyway).
GPTQ_DEV bool splitk_add2(u64_t *words, float a, float b, int S, float &ta, float &tb) {
    const u64_t addb = splitk_encode(b), adda = splitk_encode(a);
    const u64_t oldb = __hip_atomic_fetch_add(words + 1, addb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64_t olda = __hip_atomic_fetch_add(words, adda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64_t nowa = olda + adda;
    if ((int)(nowa >> 56) != S) return false;
    u64_t nowb = oldb + addb;
    for (int spin = 0; (int)(nowb >> 56) != S && spin < (1 << 22); spin++)
        nowb = __hip_atomic_load(words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(words, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(words + 1, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ta = splitk_decode(nowa, S);
    tb = splitk_decode(nowb, S);
    return true;
}

// Opaque register-resident constants: (a & mask) | magic is ONE v_and_or_b32 only when the mask
// sits in an SGPR and the magic in a VGPR (VOP3 on gfx9 takes no literals); with literal operands
// hipcc emits v_and_b32 + v_or_b32.
GPTQ_DEV uint32_t vreg_const(uint32_t c) {
    uint32_t v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(c));
    return v;
}
GPTQ_DEV uint32_t sreg_const(uint32_t c) {
    uint32_t v;
    asm("s_mov_b32 %0, %1" : "=s"(v) : "i"(c));
    return v;
}

// 16-byte system-scope (sc0 sc1: write-through / cache-bypassing) accesses for hand-offs between
// workgroups (MI355X_MICROARCH.md "Valid forms").  The loads are issued without a wait; call
// wait_sys_loads() on the whole batch before using any of them.
// The s_nop is REQUIRED: a global store of more than 8 bytes reads its data VGPRs one wait state after
// issue ("12-dword store" hazard).  The compiler's hazard recogniser does not look inside inline asm, so
// without it the next VALU write to v's first register was stored instead of v (found by the fused-MLP
// fuzz test: zeros at every 4th column of the partial tiles).
GPTQ_DEV void store_sys16(float *p, float4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// Eight independent 16-byte system-scope loads and their wait as ONE asm statement: the compiler
// cannot see that the destination registers are not valid until the s_waitcnt, so issue and wait
// must not be separable (a register move scheduled in between would copy stale data).  Early-clobber outputs: the data
// lands asynchronously, so no destination may share registers with any of the address operands.
GPTQ_DEV void load_sys16_x8(float4_t (&v)[8], const float *const (&p)[8]) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
        "global_load_dwordx4 %1, %9, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc0 sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc0 sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc0 sc1\n\t"
        "global_load_dwordx4 %5, %13, off sc0 sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc0 sc1\n\t"
        "global_load_dwordx4 %7, %15, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "memory");
}

// Development-only s_memtime / s_memrealtime checkpoints (tools/timeline.py).
GPTQ_DEV u64_t stamp_cycles(uint32_t dep) {
    u64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return t;
}
GPTQ_DEV u64_t stamp_realtime() {
    u64_t t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
