// pk_micro.hip -- issue cost of VALU instructions, one wave or two per SIMD (generated body: one asm block of 32 instructions per
// repetition, so the compiler inserts nothing between them).  
//   hipcc --offload-arch=gfx950 -O3 -o pk_micro.bin pk_micro.hip && ./pk_micro.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 64
#define NITEM 12
template <int NT>
__global__ __launch_bounds__(NT, 1) void k_micro(long long* out, float* sink, int zero) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long t[NITEM + 1];
    f32x2 a[8], s[8];
    for (int i = 0; i < 8; i++) {
        a[i] = f32x2{(float)(tid + i) * 1e-3f, (float)(tid - i) * 1e-3f};
        s[i] = f32x2{(float)i, (float)-i};
    }
    float sf[8], af[8];
    for (int i = 0; i < 8; i++) {
        sf[i] = s[i].x;
        af[i] = a[i].y;
    }
    __syncthreads();
    int k = 0;
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %4, %12, %12, %4\nv_pk_fma_f32 %5, %13, %13, %5\nv_pk_fma_f32 %6, %14, %14, %6\nv_pk_fma_f32 %7, %15, %15, %7\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %4, %12, %12, %4\nv_pk_fma_f32 %5, %13, %13, %5\nv_pk_fma_f32 %6, %14, %14, %6\nv_pk_fma_f32 %7, %15, %15, %7\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %4, %12, %12, %4\nv_pk_fma_f32 %5, %13, %13, %5\nv_pk_fma_f32 %6, %14, %14, %6\nv_pk_fma_f32 %7, %15, %15, %7\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %2, %10, %10, %2\nv_pk_fma_f32 %3, %11, %11, %3\nv_pk_fma_f32 %4, %12, %12, %4\nv_pk_fma_f32 %5, %13, %13, %5\nv_pk_fma_f32 %6, %14, %14, %6\nv_pk_fma_f32 %7, %15, %15, %7" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %1, %9, %9, %1" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0\nv_pk_fma_f32 %0, %8, %8, %0" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_pk_add_f32 %0, %8, %0\nv_pk_add_f32 %1, %9, %1\nv_pk_add_f32 %2, %10, %2\nv_pk_add_f32 %3, %11, %3\nv_pk_add_f32 %4, %12, %4\nv_pk_add_f32 %5, %13, %5\nv_pk_add_f32 %6, %14, %6\nv_pk_add_f32 %7, %15, %7\nv_pk_add_f32 %0, %8, %0\nv_pk_add_f32 %1, %9, %1\nv_pk_add_f32 %2, %10, %2\nv_pk_add_f32 %3, %11, %3\nv_pk_add_f32 %4, %12, %4\nv_pk_add_f32 %5, %13, %5\nv_pk_add_f32 %6, %14, %6\nv_pk_add_f32 %7, %15, %7\nv_pk_add_f32 %0, %8, %0\nv_pk_add_f32 %1, %9, %1\nv_pk_add_f32 %2, %10, %2\nv_pk_add_f32 %3, %11, %3\nv_pk_add_f32 %4, %12, %4\nv_pk_add_f32 %5, %13, %5\nv_pk_add_f32 %6, %14, %6\nv_pk_add_f32 %7, %15, %7\nv_pk_add_f32 %0, %8, %0\nv_pk_add_f32 %1, %9, %1\nv_pk_add_f32 %2, %10, %2\nv_pk_add_f32 %3, %11, %3\nv_pk_add_f32 %4, %12, %4\nv_pk_add_f32 %5, %13, %5\nv_pk_add_f32 %6, %14, %6\nv_pk_add_f32 %7, %15, %7" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_pk_mul_f32 %0, %8, %0\nv_pk_mul_f32 %1, %9, %1\nv_pk_mul_f32 %2, %10, %2\nv_pk_mul_f32 %3, %11, %3\nv_pk_mul_f32 %4, %12, %4\nv_pk_mul_f32 %5, %13, %5\nv_pk_mul_f32 %6, %14, %6\nv_pk_mul_f32 %7, %15, %7\nv_pk_mul_f32 %0, %8, %0\nv_pk_mul_f32 %1, %9, %1\nv_pk_mul_f32 %2, %10, %2\nv_pk_mul_f32 %3, %11, %3\nv_pk_mul_f32 %4, %12, %4\nv_pk_mul_f32 %5, %13, %5\nv_pk_mul_f32 %6, %14, %6\nv_pk_mul_f32 %7, %15, %7\nv_pk_mul_f32 %0, %8, %0\nv_pk_mul_f32 %1, %9, %1\nv_pk_mul_f32 %2, %10, %2\nv_pk_mul_f32 %3, %11, %3\nv_pk_mul_f32 %4, %12, %4\nv_pk_mul_f32 %5, %13, %5\nv_pk_mul_f32 %6, %14, %6\nv_pk_mul_f32 %7, %15, %7\nv_pk_mul_f32 %0, %8, %0\nv_pk_mul_f32 %1, %9, %1\nv_pk_mul_f32 %2, %10, %2\nv_pk_mul_f32 %3, %11, %3\nv_pk_mul_f32 %4, %12, %4\nv_pk_mul_f32 %5, %13, %5\nv_pk_mul_f32 %6, %14, %6\nv_pk_mul_f32 %7, %15, %7" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_fma_f32 %0, %8, %8, %0\nv_fma_f32 %1, %9, %9, %1\nv_fma_f32 %2, %10, %10, %2\nv_fma_f32 %3, %11, %11, %3\nv_fma_f32 %4, %12, %12, %4\nv_fma_f32 %5, %13, %13, %5\nv_fma_f32 %6, %14, %14, %6\nv_fma_f32 %7, %15, %15, %7\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %1, %9, %9, %1\nv_fma_f32 %2, %10, %10, %2\nv_fma_f32 %3, %11, %11, %3\nv_fma_f32 %4, %12, %12, %4\nv_fma_f32 %5, %13, %13, %5\nv_fma_f32 %6, %14, %14, %6\nv_fma_f32 %7, %15, %15, %7\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %1, %9, %9, %1\nv_fma_f32 %2, %10, %10, %2\nv_fma_f32 %3, %11, %11, %3\nv_fma_f32 %4, %12, %12, %4\nv_fma_f32 %5, %13, %13, %5\nv_fma_f32 %6, %14, %14, %6\nv_fma_f32 %7, %15, %15, %7\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %1, %9, %9, %1\nv_fma_f32 %2, %10, %10, %2\nv_fma_f32 %3, %11, %11, %3\nv_fma_f32 %4, %12, %12, %4\nv_fma_f32 %5, %13, %13, %5\nv_fma_f32 %6, %14, %14, %6\nv_fma_f32 %7, %15, %15, %7" : "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(sf[3]), "+v"(sf[4]), "+v"(sf[5]), "+v"(sf[6]), "+v"(sf[7]) : "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(af[4]), "v"(af[5]), "v"(af[6]), "v"(af[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0\nv_fma_f32 %0, %8, %8, %0" : "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(sf[3]), "+v"(sf[4]), "+v"(sf[5]), "+v"(sf[6]), "+v"(sf[7]) : "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(af[4]), "v"(af[5]), "v"(af[6]), "v"(af[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_add_f32 %0, %8, %0\nv_add_f32 %1, %9, %1\nv_add_f32 %2, %10, %2\nv_add_f32 %3, %11, %3\nv_add_f32 %4, %12, %4\nv_add_f32 %5, %13, %5\nv_add_f32 %6, %14, %6\nv_add_f32 %7, %15, %7\nv_add_f32 %0, %8, %0\nv_add_f32 %1, %9, %1\nv_add_f32 %2, %10, %2\nv_add_f32 %3, %11, %3\nv_add_f32 %4, %12, %4\nv_add_f32 %5, %13, %5\nv_add_f32 %6, %14, %6\nv_add_f32 %7, %15, %7\nv_add_f32 %0, %8, %0\nv_add_f32 %1, %9, %1\nv_add_f32 %2, %10, %2\nv_add_f32 %3, %11, %3\nv_add_f32 %4, %12, %4\nv_add_f32 %5, %13, %5\nv_add_f32 %6, %14, %6\nv_add_f32 %7, %15, %7\nv_add_f32 %0, %8, %0\nv_add_f32 %1, %9, %1\nv_add_f32 %2, %10, %2\nv_add_f32 %3, %11, %3\nv_add_f32 %4, %12, %4\nv_add_f32 %5, %13, %5\nv_add_f32 %6, %14, %6\nv_add_f32 %7, %15, %7" : "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(sf[3]), "+v"(sf[4]), "+v"(sf[5]), "+v"(sf[6]), "+v"(sf[7]) : "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(af[4]), "v"(af[5]), "v"(af[6]), "v"(af[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_max_f32 %0, %8, %0\nv_max_f32 %1, %9, %1\nv_max_f32 %2, %10, %2\nv_max_f32 %3, %11, %3\nv_max_f32 %4, %12, %4\nv_max_f32 %5, %13, %5\nv_max_f32 %6, %14, %6\nv_max_f32 %7, %15, %7\nv_max_f32 %0, %8, %0\nv_max_f32 %1, %9, %1\nv_max_f32 %2, %10, %2\nv_max_f32 %3, %11, %3\nv_max_f32 %4, %12, %4\nv_max_f32 %5, %13, %5\nv_max_f32 %6, %14, %6\nv_max_f32 %7, %15, %7\nv_max_f32 %0, %8, %0\nv_max_f32 %1, %9, %1\nv_max_f32 %2, %10, %2\nv_max_f32 %3, %11, %3\nv_max_f32 %4, %12, %4\nv_max_f32 %5, %13, %5\nv_max_f32 %6, %14, %6\nv_max_f32 %7, %15, %7\nv_max_f32 %0, %8, %0\nv_max_f32 %1, %9, %1\nv_max_f32 %2, %10, %2\nv_max_f32 %3, %11, %3\nv_max_f32 %4, %12, %4\nv_max_f32 %5, %13, %5\nv_max_f32 %6, %14, %6\nv_max_f32 %7, %15, %7" : "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(sf[3]), "+v"(sf[4]), "+v"(sf[5]), "+v"(sf[6]), "+v"(sf[7]) : "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(af[4]), "v"(af[5]), "v"(af[6]), "v"(af[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_pk_max_i16 %0, %8, %0\nv_pk_max_i16 %1, %9, %1\nv_pk_max_i16 %2, %10, %2\nv_pk_max_i16 %3, %11, %3\nv_pk_max_i16 %4, %12, %4\nv_pk_max_i16 %5, %13, %5\nv_pk_max_i16 %6, %14, %6\nv_pk_max_i16 %7, %15, %7\nv_pk_max_i16 %0, %8, %0\nv_pk_max_i16 %1, %9, %1\nv_pk_max_i16 %2, %10, %2\nv_pk_max_i16 %3, %11, %3\nv_pk_max_i16 %4, %12, %4\nv_pk_max_i16 %5, %13, %5\nv_pk_max_i16 %6, %14, %6\nv_pk_max_i16 %7, %15, %7\nv_pk_max_i16 %0, %8, %0\nv_pk_max_i16 %1, %9, %1\nv_pk_max_i16 %2, %10, %2\nv_pk_max_i16 %3, %11, %3\nv_pk_max_i16 %4, %12, %4\nv_pk_max_i16 %5, %13, %5\nv_pk_max_i16 %6, %14, %6\nv_pk_max_i16 %7, %15, %7\nv_pk_max_i16 %0, %8, %0\nv_pk_max_i16 %1, %9, %1\nv_pk_max_i16 %2, %10, %2\nv_pk_max_i16 %3, %11, %3\nv_pk_max_i16 %4, %12, %4\nv_pk_max_i16 %5, %13, %5\nv_pk_max_i16 %6, %14, %6\nv_pk_max_i16 %7, %15, %7" : "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(sf[3]), "+v"(sf[4]), "+v"(sf[5]), "+v"(sf[6]), "+v"(sf[7]) : "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(af[4]), "v"(af[5]), "v"(af[6]), "v"(af[7]));
    t[k++] = clock64();
    for (int r = 0; r < REP; r++) asm volatile("v_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %10\nv_mov_b32 %3, %11\nv_mov_b32 %4, %12\nv_mov_b32 %5, %13\nv_mov_b32 %6, %14\nv_mov_b32 %7, %15\nv_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %10\nv_mov_b32 %3, %11\nv_mov_b32 %4, %12\nv_mov_b32 %5, %13\nv_mov_b32 %6, %14\nv_mov_b32 %7, %15\nv_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %10\nv_mov_b32 %3, %11\nv_mov_b32 %4, %12\nv_mov_b32 %5, %13\nv_mov_b32 %6, %14\nv_mov_b32 %7, %15\nv_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %10\nv_mov_b32 %3, %11\nv_mov_b32 %4, %12\nv_mov_b32 %5, %13\nv_mov_b32 %6, %14\nv_mov_b32 %7, %15" : "+v"(sf[0]), "+v"(sf[1]), "+v"(sf[2]), "+v"(sf[3]), "+v"(sf[4]), "+v"(sf[5]), "+v"(sf[6]), "+v"(sf[7]) : "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(af[4]), "v"(af[5]), "v"(af[6]), "v"(af[7]));
    t[k++] = clock64();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += s[i].x + s[i].y + sf[i];
    if (zero) sink[tid] = acc;
    if (lane == 0)
        for (int i = 0; i < NITEM; i++) out[((size_t)blockIdx.x * (NT / 64) + wave) * NITEM + i] = t[i + 1] - t[i];
}
template <int NT>
static void run(const char* title) {
    long long* d_out;
    float* d_sink;
    const int NB = 256, NW = NT / 64;
    (void)hipMalloc(&d_out, sizeof(long long) * NB * NW * NITEM);
    (void)hipMalloc(&d_sink, 4 * NT);
    (void)hipMemset(d_out, 0, sizeof(long long) * NB * NW * NITEM);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_micro<NT>, dim3(NB), dim3(NT), 0, 0, d_out, d_sink, 0);
    (void)hipDeviceSynchronize();
    std::vector<long long> h((size_t)NB * NW * NITEM);
    (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"v_pk_fma_f32, 4 chains", "v_pk_fma_f32, 8 chains", "v_pk_fma_f32, 2 chains", "v_pk_fma_f32, 1 chain", "v_pk_add_f32, 8 chains", "v_pk_mul_f32, 8 chains", "v_fma_f32, 8 chains", "v_fma_f32, 1 chain", "v_add_f32, 8 chains", "v_max_f32, 8 chains", "v_pk_max_i16, 8 chains", "v_mov_b32, 8 independent"};
    printf("== %s\n", title);
    for (int i = 0; i < NITEM; i++) {
        std::vector<long long> v;
        for (int b = 0; b < NB * NW; b++) v.push_back(h[(size_t)b * NITEM + i]);
        std::sort(v.begin(), v.end());
        printf("%-36s %6.2f cycles per instruction and wave (median over waves; 32 per repetition)\n", names[i], (double)v[v.size() / 2] / REP / 32);
    }
    (void)hipFree(d_out);
    (void)hipFree(d_sink);
}
int main() {
    run<256>("one wave per SIMD (256 threads per workgroup, one workgroup per CU)");
    run<512>("two waves per SIMD (512 threads)");
    run<1024>("four waves per SIMD (1024 threads)");
    return 0;
}
