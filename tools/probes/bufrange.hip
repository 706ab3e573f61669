// bufrange.hip -- does the raw-buffer range check include the SGPR offset (soffset) on gfx950?  (LLVM documents soffset as "excluded from bounds
// checking"; stripe_mm.inc relied on rows past M - 1 reading as zero with the row offset in soffset.)  Prints what four loads return.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/bufrange.hip -o tools/probes/bufrange && tools/probes/bufrange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t *buf, uint32_t *out, int nrec, int soff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, nrec, 0x00020000);
    u32x4 a = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0));          // in range
    u32x4 b = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, 1024, 0, 0));       // voffset past num_records
    u32x4 c = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, 0, soff, 0));       // soffset past num_records
    u32x4 d = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, 240, 0, 0));        // last in-range 16 bytes (nrec = 256)
    u32x4 e = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, 244, 0, 0));        // straddles the end
    if (threadIdx.x == 0) { out[0] = a[0]; out[1] = b[0]; out[2] = c[0]; out[3] = d[0]; out[4] = e[0]; out[5] = e[3]; }
}
int main() {
    uint32_t h[1024], *d, *o, r[6];
    for (int i = 0; i < 1024; i++) h[i] = 0x1000 + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 256, 1024);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("in-range %x | voffset 1024 past nrec 256: %x | soffset 1024 past nrec: %x (memory holds %x) | last in-range: %x | straddling: %x %x\n", r[0], r[1], r[2], h[256], r[3], r[4],
           r[5]);
    printf("soffset %s the range check\n", r[2] == 0 ? "IS INCLUDED in" : "is EXCLUDED from");
    return 0;
}
