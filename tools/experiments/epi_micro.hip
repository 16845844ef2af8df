// epi_micro.hip -- developer microbenchmarks behind the tower's epilogue work: what do the building blocks of
// a LayerNorm / squeeze-excitation epilogue cost on one wave per SIMD?  256 workgroups x 256 threads (one workgroup per CU,
// like the tower), cycles from s_memtime around REP repetitions, median over workgroups printed per item.
//   hipcc --offload-arch=gfx950 -O3 -o epi_micro.bin epi_micro.hip && ./epi_micro.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP 64
#define NITEM 16

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

__global__ __launch_bounds__(256, 1) void k_micro(long long* out, float* sink, int zero) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long t[NITEM + 1];
    f32x2 a[16];
    for (int i = 0; i < 16; i++) a[i] = f32x2{(float)(tid + i) * 1e-3f, (float)(tid - i) * 1e-3f};
    f32x2 s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0}, s3 = {0, 0};
    __syncthreads();
    int k = 0;
#define T() t[k++] = clock64()
    T();
    // 0: 32 independent-ish pk_fma per rep (4 chains of 8)
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            s0 = a[i] * a[i] + s0;
            s1 = a[i + 1] * a[i + 1] + s1;
            s2 = a[i + 2] * a[i + 2] + s2;
            s3 = a[i + 3] * a[i + 3] + s3;
        }
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            s0 = a[i] + s0;
            s1 = a[i + 1] + s1;
            s2 = a[i + 2] + s2;
            s3 = a[i + 3] + s3;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    T();
    // 1: 32 pk ops per rep in ONE dependent chain
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s0 = a[i] * s0 + a[i];
#pragma unroll
        for (int i = 0; i < 16; i++) s0 = a[i] + s0;
        __builtin_amdgcn_sched_barrier(0);
    }
    T();
    // 2: 32 scalar v_fma_f32 in 4 chains
    float x0 = s0.x, x1 = s1.x, x2 = s2.x, x3 = s3.x;
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            x0 = a[i].x * x0 + a[i].y;
            x1 = a[i + 1].x * x1 + a[i].y;
            x2 = a[i + 2].x * x2 + a[i].y;
            x3 = a[i + 3].x * x3 + a[i].y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    T();
    // 3: LDS write (b64 per lane) -> barrier -> 4 reads (b64) + sum  (the LayerNorm statistics exchange)
    float2* st = reinterpret_cast<float2*>(lds);
    for (int r = 0; r < REP; r++) {
        st[(r & 1) * 256 + wave * 64 + lane] = make_float2(x0, x1);
        __syncthreads();
        float2 q0 = st[(r & 1) * 256 + lane], q1 = st[(r & 1) * 256 + 64 + lane], q2 = st[(r & 1) * 256 + 128 + lane], q3 = st[(r & 1) * 256 + 192 + lane];
        x0 = (q0.x + q1.x) + (q2.x + q3.x);
        x1 = (q0.y + q1.y) + (q2.y + q3.y);
    }
    T();
    // 4: barrier alone
    for (int r = 0; r < REP; r++) {
        __syncthreads();
        x2 += 1.0f;
    }
    T();
    // 5: wave-private LDS round trip: write b32, read b128 (no barrier)
    for (int r = 0; r < REP; r++) {
        lds[4096 + wave * 256 + lane] = x2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        f32x4 v = *reinterpret_cast<const f32x4*>(lds + 4096 + wave * 256 + 4 * (lane & 15));
        x2 = v.x + v.y + v.z + v.w;
    }
    T();
    // 6: v_exp_f32 + v_rcp_f32 dependent pair (sigmoid) x 4 per rep
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) x3 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x3));
    }
    T();
    // 7: 16 independent v_exp_f32 per rep
    {
        float e[16];
        for (int i = 0; i < 16; i++) e[i] = a[i].x;
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) e[i] = __builtin_amdgcn_exp2f(e[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 16; i++) x3 += e[i];
    }
    T();
    // 8: permlane32_swap + add (one per rep, dependent)
    for (int r = 0; r < REP; r++) {
        const auto rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
        x0 = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
        x1 = x0 * 0.5f;
    }
    T();
    // 9: the five-level halving butterfly on 16 values (the SE average pool), as in nn_tower32.hpp
    {
        float v[16];
        for (int i = 0; i < 16; i++) v[i] = a[i].x + x0;
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int tt = 0; tt < 8; tt++) {
                const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[tt]), __float_as_uint(v[tt + 8]), false, false);
                v[tt] = __uint_as_float(q[0]) + __uint_as_float(q[1]);
            }
            {
                const bool hi = (lane & 8) != 0;
#pragma unroll
                for (int tt = 0; tt < 4; tt++) {
                    float send = hi ? v[tt] : v[tt + 4], keep = hi ? v[tt + 4] : v[tt];
                    v[tt] = keep + dpp_f<0x140>(send);
                }
            }
            {
                const bool hi = (lane & 4) != 0;
#pragma unroll
                for (int tt = 0; tt < 2; tt++) {
                    float send = hi ? v[tt] : v[tt + 2], keep = hi ? v[tt + 2] : v[tt];
                    v[tt] = keep + dpp_f<0x141>(send);
                }
            }
            {
                const bool hi = (lane & 2) != 0;
                float send = hi ? v[0] : v[1], keep = hi ? v[1] : v[0];
                v[0] = keep + dpp_f<0x4E>(send);
            }
            v[0] += dpp_f<0xB1>(v[0]);
#pragma unroll
            for (int i = 1; i < 16; i++) v[i] = v[0] + (float)i;
            __builtin_amdgcn_sched_barrier(0);
        }
        x1 += v[0];
    }
    T();
    // 10: 16 ds_write_b64 + waitcnt (image store of one epilogue: 2 tiles x 8 stores would be 16 b64)
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<float2*>(lds + ((lane & 31) * 68 + (lane >> 5) * 2 + i * 4)) = make_float2(x0, x1 + i);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        x0 += 1.f;
    }
    T();
    // 11: 8 ds_read_b128 (parameters) + wait
    for (int r = 0; r < REP; r++) {
        f32x4 acc4 = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) acc4 += *reinterpret_cast<const f32x4*>(lds + wave * 32 + i * 256 + 4 * (lane >> 5) + ((int)x0 & 3) * 8);
        x0 = acc4.x + acc4.y + acc4.z + acc4.w;
    }
    T();
    // 12: 4 MFMA 16x16x32 dependent chain + result use
    {
        typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
        bf16x8 av, bv;
        for (int i = 0; i < 8; i++) { av[i] = (__bf16)(a[i].x); bv[i] = (__bf16)(a[i].y); }
        f32x4 d = {0, 0, 0, 0};
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 4; i++) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, d, 0, 0, 0);
            d.x += 1.0f;
        }
        x2 += d.x;
    }
    T();
    // 13: 16 cvt_pk_bf16 + 16 pk_max_i16
    {
        typedef __bf16 bf16pair __attribute__((ext_vector_type(2)));
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        unsigned acc_u = 0;
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(a[i] + f32x2{x0, x0}, bf16pair));
                acc_u ^= __builtin_bit_cast(unsigned, __builtin_elementwise_max(s, s16x2{0, 0}));
            }
            x0 += 1.0f;
            __builtin_amdgcn_sched_barrier(0);
        }
        x1 += (float)acc_u;
    }
    T();
    if (zero) sink[tid] = x0 + x1 + x2 + x3 + s0.x + s1.x + s2.x + s3.x + s0.y;
    if (lane == 0)
        for (int i = 0; i < k - 1; i++) out[((size_t)blockIdx.x * 4 + wave) * NITEM + i] = t[i + 1] - t[i];
}

int main() {
    long long* d_out;
    float* d_sink;
    const int NB = 256;
    hipMalloc(&d_out, sizeof(long long) * NB * 4 * NITEM);
    hipMalloc(&d_sink, 4 * 256);
    hipMemset(d_out, 0, sizeof(long long) * NB * 4 * NITEM);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_micro, dim3(NB), dim3(256), 0, 0, d_out, d_sink, 0);
    hipDeviceSynchronize();
    std::vector<long long> h((size_t)NB * 4 * NITEM);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"32 pk ops, 4 chains", "32 pk ops, 1 chain", "32 v_fma_f32, 4 chains", "LN stat exchange: write+barrier+4 reads", "barrier alone",
                           "wave-private LDS round trip", "4 x (exp2 + rcp) dependent", "16 independent exp2", "permlane32_swap + add", "pool butterfly (16 values)",
                           "8 ds_write_b64 + fence", "8 ds_read_b128 + use", "4 MFMA 16x16x32 chain + use", "16 cvt_pk_bf16 + 16 pk_max_i16"};
    for (int i = 0; i < 14; i++) {
        std::vector<long long> v;
        for (int b = 0; b < NB * 4; b++) v.push_back(h[(size_t)b * NITEM + i]);
        std::sort(v.begin(), v.end());
        printf("%-44s %8.1f cycles per repetition (median over waves; p90 %.1f)\n", names[i], (double)v[v.size() / 2] / REP, (double)v[v.size() * 9 / 10] / REP);
    }
    return 0;
}
