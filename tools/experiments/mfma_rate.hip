// Experiment (not product code): issue rate of the MFMA forms the towers use, one wave per SIMD, two alternating accumulators
// (the towers' pattern).  Prints cycles per instruction (s_memtime ticks).
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef long long2_ __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int n, int seed) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = 0x38303430 + l * 7 + j * seed; b[j] = 0x34383038 + l * 3 + j; }
    bf16x8 ab = __builtin_bit_cast(bf16x8, (typeof(__builtin_shufflevector(a, a, 0, 1, 2, 3))){a[0], a[1], a[2], a[3]});
    bf16x8 bb = __builtin_bit_cast(bf16x8, (typeof(__builtin_shufflevector(b, b, 0, 1, 2, 3))){b[0], b[1], b[2], b[3]});
    f32x16 c0, c1;
    for (int i = 0; i < 16; i++) { c0[i] = 0.f; c1[i] = 0.f; }
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
        } else if (MODE == 1) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(((long*)&a)[0], ((long*)&b)[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(((long*)&a)[0], ((long*)&b)[0], c1, 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* o; long long* c;
    hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&c, 1024 * 8);
    const int n = 2000;
    for (int grid : {1, 256, 1024}) {
        long long h[1024];
        k<0><<<grid, 256>>>(o, c, n, 1); hipMemcpy(h, c, grid * 8, hipMemcpyDeviceToHost);
        printf("grid %4d: bf16 32x32x16 %.1f cyc/MFMA", grid, (double)h[grid / 2] / (2.0 * n));
        k<1><<<grid, 256>>>(o, c, n, 1); hipMemcpy(h, c, grid * 8, hipMemcpyDeviceToHost);
        printf("   scaled e4m3 32x32x64 %.1f cyc/MFMA", (double)h[grid / 2] / (2.0 * n));
        k<2><<<grid, 256>>>(o, c, n, 1); hipMemcpy(h, c, grid * 8, hipMemcpyDeviceToHost);
        printf("   plain fp8 32x32x16 %.1f cyc/MFMA\n", (double)h[grid / 2] / (2.0 * n));
    }
    return 0;
}
