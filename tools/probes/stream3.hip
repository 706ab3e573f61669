// stream3.hip -- does the 3-bit image stream faster with full-width loads?  ("a lane owns 128 consecutive k of one column =
// three dwordx4 loads" against today's one dwordx3 per lane and row block.)  A bare streaming kernel with the decode kernel's launch shape for
// 4096 x 12288 at 3 bits -- 768 workgroups (one 16-column stripe each) x 512 threads, every wave requests its 3072 bytes up front, XORs them and
// stores one word per lane -- in two load shapes over the SAME bytes:
//   mode 0: four global_load_dwordx3 per lane  (wave instruction = 768 contiguous bytes; today's image: row blocks wave, wave + 8, ...)
//   mode 1: three global_load_dwordx4 per lane (wave instruction = 1024 contiguous bytes; the proposed layout)
//   mode 2: mode 0 with the wave's four row blocks adjacent (3072 contiguous bytes per wave, as in mode 1, but dwordx3 instructions)
// 40 launches in a hipGraph, hipEvents, over one buffer (cache-warm) and over 16 rotating buffers (302 MB: past the 256 MB of MALL).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/stream3.hip -o tools/probes/stream3      (test infrastructure, not product)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

constexpr int NW = 8, WAVE_BYTES = 3072, WG_BYTES = NW * WAVE_BYTES;

template <int MODE>
__global__ void __launch_bounds__(512) stream_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *base = (const char *)src + (size_t)blockIdx.x * WG_BYTES;
    uint32_t acc = 0;
    if constexpr (MODE == 1) {
        u32x4 v[3];
#pragma unroll
        for (int i = 0; i < 3; i++) v[i] = *(const u32x4 *)(base + wave * WAVE_BYTES + i * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < 3; i++) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    } else {
        u32x3 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int rb = MODE == 0 ? wave + NW * u : wave * 4 + u;
            v[u] = *(const u32x3 *)(base + rb * 768 + lane * 12);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc ^= v[u][0] ^ v[u][1] ^ v[u][2];
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

template <int MODE>
int run(const char *name, uint32_t *buf, int nbuf, size_t words, uint32_t *out, int grid, hipStream_t s) {
    const int calls = 40;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < calls; i++) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(grid), dim3(512), 0, s, buf + (size_t)(i % nbuf) * words, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, med[7];
    for (int r = 0; r < 7; r++) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        med[r] = ms * 1e3f / calls; if (med[r] < best) best = med[r];
    }
    for (int i = 0; i < 7; i++) for (int j = i + 1; j < 7; j++) if (med[j] < med[i]) { float t = med[i]; med[i] = med[j]; med[j] = t; }
    const double bytes = (double)grid * WG_BYTES;
    printf("%-34s %2d buffer(s): median %6.2f us  best %6.2f us  %7.1f GB/s\n", name, nbuf, med[3], best, bytes / med[3] / 1e3);
    return 0;
}

int main() {
    const int grid = 768, nbuf = 16;
    const size_t words = (size_t)grid * WG_BYTES / 4;
    uint32_t *buf, *out;
    CK(hipMalloc(&buf, words * 4 * nbuf)); CK(hipMalloc(&out, (size_t)grid * 512 * 4));
    std::vector<uint32_t> h(words * nbuf);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    CK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("3-bit image of 4096 x 12288: %.2f MB per launch, %d workgroups x 512 threads, %d B per wave\n", grid * (double)WG_BYTES / 1e6, grid, WAVE_BYTES);
    for (int nb : {1, nbuf}) {
        if (run<0>("4 x dwordx3, row blocks w + 8 u", buf, nb, words, out, grid, s)) return 1;
        if (run<2>("4 x dwordx3, adjacent row blocks", buf, nb, words, out, grid, s)) return 1;
        if (run<1>("3 x dwordx4 (proposed layout)", buf, nb, words, out, grid, s)) return 1;
    }
    return 0;
}
