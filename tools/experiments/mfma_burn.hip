// Experiment (not product code): what matrix rate, clock and socket power does the chip HOLD when every SIMD issues MFMAs
// back to back for about a second?  The towers' roofline fraction is priced against the nominal peak (every pipe busy at
// 2.4 GHz); this prints the rate a register-resident MFMA loop sustains on this device, with random (non-zero) operands
// as in the towers, or all-zero operands (data-dependent power).
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/mfma_burn.hip -o gpurun_out/mfma_burn && gpurun_out/mfma_burn [seconds] [waves_per_simd] [zero]
// rocm-smi is sampled by the caller (tools/power_trace.sh style); this program prints TFLOP/s per ~0.1 s slice.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(256) void burn(float* out, int n, int zero) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = zero ? 0 : (0x38303430 + l * 7919 + j * 104729) & 0x3f7f3f7f;   // bf16 pairs / e4m3 bytes of moderate magnitude
        b[j] = zero ? 0 : (0x34383038 + l * 31337 + j * 7) & 0x3f7f3f7f;
    }
    const bf16x8 ab = __builtin_bit_cast(bf16x8, (i32x4){a[0], a[1], a[2], a[3]});
    const bf16x8 bb = __builtin_bit_cast(bf16x8, (i32x4){b[0], b[1], b[2], b[3]});
    f32x16 c[4];
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 16; i++) c[k][i] = 0.f;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (MODE == 0) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c[k], 0, 0, 0);
            else c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[k], 0, 0, 0, 127, 0, 127);
        }
    }
    float s = 0;
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 16; i++) s += c[k][i];
    out[blockIdx.x * blockDim.x + l] = s;
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    const int wps = argc > 2 ? atoi(argv[2]) : 1;
    const int zero = argc > 3 ? atoi(argv[3]) : 0;
    float* o;
    hipMalloc(&o, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wps, n = 20000;   // per launch: grid * 4 waves * n * 4 MFMAs
    for (int mode = 0; mode < 2; mode++) {
        const double flop = (double)grid * 4 * n * 4 * (mode == 0 ? 32768.0 : 131072.0);
        const auto t0 = std::chrono::steady_clock::now();
        int it = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
            hipEventRecord(e0);
            for (int r = 0; r < 4; r++) {
                if (mode == 0) burn<0><<<grid, 256>>>(o, n, zero);
                else burn<1><<<grid, 256>>>(o, n, zero);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (it++ % 4 == 0)
                printf("%s waves/SIMD %d zero %d: %.0f TFLOP/s  (%.1f %% of %s)\n", mode == 0 ? "bf16 32x32x16" : "e4m3 32x32x64", wps, zero, 4 * flop / (ms * 1e-3) / 1e12,
                       100.0 * 4 * flop / (ms * 1e-3) / (mode == 0 ? 2.5e15 : 5.0e15), mode == 0 ? "2.5 PF" : "5 PF");
            fflush(stdout);
        }
    }
    return 0;
}
