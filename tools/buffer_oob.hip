// What does a raw buffer_load_dwordx4 return when only PART of the 16 bytes lies inside num_records?  (gfx950)
// hipcc --offload-arch=gfx950 -O2 -o /tmp/buffer_oob tools/buffer_oob.hip && /tmp/buffer_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int rsrc_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, unsigned records, float* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, records, 0x00020000);
    const unsigned off = threadIdx.x * 4u;                       // dword-aligned offsets 0, 4, 8, ...
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = __uint_as_float(v[e]);
}
int main() {
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * 4 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    const unsigned records = 10 * 4;                             // 10 floats in range
    hipLaunchKernelGGL(probe, dim3(1), dim3(16), 0, 0, d, records, o);
    float r[64]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    for (int t = 0; t < 12; ++t) printf("offset %2d floats: %6.1f %6.1f %6.1f %6.1f\n", t, r[4 * t], r[4 * t + 1], r[4 * t + 2], r[4 * t + 3]);
    return 0;
}
