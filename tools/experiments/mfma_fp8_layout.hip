// Experiment (not product code): pins the operand / scale lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 with exact
// small-integer e4m3 data, and the rounding / saturation of v_cvt_pk_fp8_f32, before the fp8 tower relies on them
// (cdna_hip_programming.md section 3: "check the map with exact integer data before relying on it").
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/mfma_fp8_layout.hip -o /tmp/mfma_fp8_layout && /tmp/mfma_fp8_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// A: [32 rows][64 k] bytes row-major; B: [64 k][32 cols] stored as Bt[32 cols][64 k]
__global__ void k(const uint8_t* A, const uint8_t* Bt, float* D, int sa, int sb, int mode) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    if (mode == 1) sa = 127 + (r % 3);            // per-row scale, same for both K blocks
    if (mode == 2) sa = 127 + h;                  // per-K-block scale (lanes 32..63 own block 1)
    if (mode == 3) sb = 127 + (r % 2);            // per-column scale of B
    i32x8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = *(const int*)(A + r * 64 + 32 * h + 4 * j);
        b[j] = *(const int*)(Bt + r * 64 + 32 * h + 4 * j);
    }
    f32x16 c;
    for (int i = 0; i < 16; i++) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; i++) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D[row * 32 + r] = c[i];
    }
}
__global__ void kcvt(const float* x, uint32_t* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[i], 0, false);
}
static float e4m3(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : (e == 15 && m == 7) ? NAN : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
int main() {
    std::vector<uint8_t> A(32 * 64), Bt(32 * 64);
    for (int r = 0; r < 32; r++)
        for (int k2 = 0; k2 < 64; k2++) {
            A[r * 64 + k2] = (uint8_t)(0x30 + ((r * 7 + k2 * 3) % 23));          // small positive values
            Bt[r * 64 + k2] = (uint8_t)(0x28 + ((r * 5 + k2 * 11) % 19) + ((k2 & 1) ? 0x80 : 0));
        }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bt.data(), 2048, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; mode++)
    for (int sa : {127, 128, 0x7f7f7f80}) for (int sb : {127, 126}) {
        if (mode && (sa != 127 || sb != 127)) continue;
        k<<<1, 64>>>(dA, dB, dD, sa, sb, mode);
        std::vector<float> D(1024);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0; double maxerr = 0;
        float scale = ldexpf(1.0f, ((sa & 255) - 127) + ((sb & 255) - 127));
        for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
            double ref = 0;
            for (int k2 = 0; k2 < 64; k2++) ref += (double)e4m3(A[i * 64 + k2]) * e4m3(Bt[j * 64 + k2]);
            ref *= scale;
            if (mode == 1) ref *= (double)(1 << (i % 3));
            if (mode == 3) ref *= (double)(1 << (j % 2));
            if (mode == 2) {
                ref = 0;
                for (int k2 = 0; k2 < 64; k2++) ref += (double)e4m3(A[i * 64 + k2]) * e4m3(Bt[j * 64 + k2]) * (k2 >= 32 ? 2.0 : 1.0);
            }
            double e = fabs(ref - D[i * 32 + j]);
            if (e > 1e-3 * fabs(ref) + 1e-4) bad++;
            if (e > maxerr) maxerr = e;
        }
        printf("mode=%d sa=%x sb=%x bad=%d maxerr=%g D[0][0]=%g D[1][0]=%g D[0][1]=%g\n", mode, sa, sb, bad, maxerr, D[0], D[32], D[1]);
    }
    float xs[16] = {0.f, 1.f, 1.0625f, 1.125f, 1.1875f, 447.f, 448.f, 464.f, 480.f, 1000.f, 1e30f, -1000.f, 0.001f, 0.0009765625f, 0.00146f, NAN};
    float* dx; uint32_t* dout; hipMalloc(&dx, 64); hipMalloc(&dout, 64);
    hipMemcpy(dx, xs, 64, hipMemcpyHostToDevice);
    kcvt<<<1, 64>>>(dx, dout, 16);
    uint32_t o[16]; hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++) printf("cvt %g -> %02x (%g)\n", xs[i], o[i] & 255, e4m3(o[i] & 255));
    return 0;
}
