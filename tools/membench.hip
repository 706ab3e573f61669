// membench.hip -- what read pattern reaches HBM speed on a matrix that is read ONCE?
// Reads a row-major [R][C16] matrix of 16-byte elements with a 2-D decomposition:
//   tile  = NL consecutive 16-B elements of a row (NL*16 contiguous bytes)
//   block = tile x a K-slice of rows; T threads = NL column lanes x KL row lanes
//   every lane keeps U loads in flight; rows are taken interleaved (r = r0 + kl + i*KL).
// Cold protocol: NBUF distinct matrices (> 256 MiB total) rotated inside one hipGraph.
// build: hipcc --offload-arch=gfx950 -O3 -o membench tools/membench.hip ; run: ./membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(1024) read_kernel(const u32x4* __restrict__ base, int R, int C16, int NL, int splitk, int rows_per_slice, unsigned* out) {
    const int T = blockDim.x, KL = T / NL;
    const int tid = threadIdx.x;
    const int cg = tid % NL, kl = tid / NL;
    const int tile = blockIdx.x / splitk, slice = blockIdx.x % splitk;
    const int c = tile * NL + cg;
    const int r0 = slice * rows_per_slice, r1 = min(R, r0 + rows_per_slice);
    u32x4 acc = {0, 0, 0, 0};
    if (c < C16) {
        for (int r = r0 + kl; r < r1; r += KL * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int rr = r + u * KL;
                if (rr < r1) {
                    const u32x4* p = base + (size_t)rr * C16 + c;
                    v[u] = NT ? __builtin_nontemporal_load(p) : *p;
                } else v[u] = u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= v[u];
        }
    }
    unsigned x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (x == 0x9e3779b9u) out[blockIdx.x] = x;   // keeps the loads alive, (almost) never stores
}

__global__ void empty_kernel(unsigned* out) { if (threadIdx.x == 1025) out[0] = 1; }

struct Cfg { const char* name; int NL, T, splitk; };

template <int U, bool NT>
static float run(const std::vector<u32x4*>& bufs, int R, int C16, Cfg c, unsigned* out, hipStream_t s, int reps) {
    const int ntiles = (C16 + c.NL - 1) / c.NL;
    const int rps = (R + c.splitk - 1) / c.splitk;
    dim3 grid(ntiles * c.splitk), block(c.T);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (auto b : bufs) hipLaunchKernelGGL((read_kernel<U, NT>), grid, block, 0, s, b, R, C16, c.NL, c.splitk, rps, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (reps * bufs.size());
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned* out; CK(hipMalloc(&out, 1 << 20));
    // kernel boundary floor
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 64; i++) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, out);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel chain: %.3f us per launch\n", ms * 1e3 / (20 * 64));
    }
    struct Shape { const char* name; int R, C16; } shapes[] = {{"4096x4096 (512 x 16KB)", 512, 1024}, {"4096x12288 (512 x 48KB)", 512, 3072}, {"11008x4096 (1376 x 16KB)", 1376, 1024}, {"4096x11008 (512 x 43KB)", 512, 2752}};
    Cfg cfgs[] = {
        {"stripe64B  T512 s1", 4, 512, 1}, {"stripe64B  T256 s1", 4, 256, 1}, {"stripe64B T1024 s1", 4, 1024, 1},
        {"stripe128B T512 s1", 8, 512, 1}, {"stripe128B T256 s2", 8, 256, 2}, {"stripe256B T256 s1", 16, 256, 1}, {"stripe256B T256 s4", 16, 256, 4},
        {"stripe256B T512 s4", 16, 512, 4}, {"stripe512B T256 s4", 32, 256, 4}, {"stripe512B T256 s8", 32, 256, 8},
        {"row1KB     T256 s8", 64, 256, 8}, {"row1KB     T256 s16", 64, 256, 16}, {"row1KB     T256 s32", 64, 256, 32}, {"row1KB     T512 s16", 64, 512, 16},
        {"row4KB     T256 s32", 256, 256, 32}, {"row4KB     T256 s64", 256, 256, 64}, {"row4KB     T256 s128", 256, 256, 128}, {"row4KB    T1024 s32", 256, 1024, 32},
        {"row16KB   T1024 s64", 1024, 1024, 64}, {"row16KB   T1024 s128", 1024, 1024, 128}, {"row16KB   T1024 s256", 1024, 1024, 256},
    };
    for (auto& sh : shapes) {
        const size_t bytes = (size_t)sh.R * sh.C16 * 16;
        const int nbuf = (int)((400ull << 20) / bytes) + 1;
        std::vector<u32x4*> bufs(nbuf);
        for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemsetAsync(b, 0x5a, bytes, s)); }
        CK(hipStreamSynchronize(s));
        printf("== %s: %.1f MB x %d buffers\n", sh.name, bytes / 1e6, nbuf);
        for (auto& c : cfgs) {
            if (c.NL > sh.C16) continue;
            float u8 = run<8, true>(bufs, sh.R, sh.C16, c, out, s, 5);
            float u4 = run<4, true>(bufs, sh.R, sh.C16, c, out, s, 5);
            float u8p = run<8, false>(bufs, sh.R, sh.C16, c, out, s, 5);
            const int ntiles = (sh.C16 + c.NL - 1) / c.NL;
            printf("  %-22s wgs %5d | U8 nt %6.2f us %6.0f GB/s | U4 nt %6.2f us %6.0f GB/s | U8 plain %6.2f us %6.0f GB/s\n", c.name, ntiles * c.splitk,
                   u8, bytes / u8 / 1e3, u4, bytes / u4 / 1e3, u8p, bytes / u8p / 1e3);
        }
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
