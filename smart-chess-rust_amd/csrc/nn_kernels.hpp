// nn_kernels.hpp -- policy/value network forward for gfx950 (MI355X): shared device helpers and the value head's
// batched kernels.  The tower itself lives in nn_tower32.hpp.
//
// Reproduces ChessModule.forward (reference py/module.py:135-154; ResBlockSE :14-46, PolicyHead
// :65-80, ValueHead :83-106) with the numerics of the exported bf16 graph the Rust backends load
// (src/backends/torch.rs:115-125): bf16 GEMM operands, fp32 accumulate,
// fp32 LayerNorm and residual stream, fp32 log-softmax.
//
// Design (MI355X-first, not a port of any library graph):
//   * ONE workgroup (4 waves, one per SIMD) owns ONE position for the whole tower.  The position's
//     activation never leaves the CU: a zero-haloed bf16 image [10x10 pixels][C] in LDS feeds the
//     implicit GEMM of every 3x3 conv, the fp32 residual stream stays in registers, and LayerNorm / SE /
//     residual / ReLU are register epilogues.  No activation is written to HBM between layers; per position
//     the kernel reads 7 KB of input planes and writes 32 KB (value-head features) + <=19 KB (policy).
//   * Weights are streamed from L2 / Infinity Cache straight into VGPRs, pre-packed on the host in MFMA
//     fragment order (one contiguous 1 KiB per wave-load), behind a register prefetch ring that is carried
//     from layer to layer.  All 256 workgroups walk the layers roughly in step, so a layer's weights are
//     served from the XCD's L2 after the first toucher.
//   * Only the value head's Linear(16391->128) is batched ACROSS positions (its weight is 4 MB and
//     position-specific in K): k_value_fc1 is a split-K MFMA GEMM over the whole batch, reduced in
//     fixed order by k_value_finish -- or, in self-play, by the search kernel's fused tail
//     (bitwise reproducible, no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nn_types.hpp"
#include "value_tail.hpp"

namespace scnn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline bf16_t f2bf(float x) {
    __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
}
__device__ inline float bf2f(bf16_t u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }

// Wave-wide sum / max with the result in every lane: four DPP steps inside each row of 16 lanes (quad_perm,
// row_half_mirror, row_mirror -- plain VALU moves) and four readlanes, instead of six dependent trips through the LDS
// crossbar (ds_bpermute).  Fixed order, so results are reproducible.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
__device__ __forceinline__ float wave_sum64(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max64(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// The kernel arguments live in memory that is fresh at every launch, and the compiler reads them piece by piece where they are
// first needed, each piece behind its own wait: a chain of scalar-cache misses at the top of every kernel (stamped in the fused
// step kernel: 8.7 k -> 6.6 k cycles from the first instruction to the arrival of the search's control block).  One scalar load
// per 64-byte line of the argument block, all in flight together, makes the chain one miss deep (+0.5 % / +1.3 % simulations/s
// at bf16 / fp8; the stand-alone tower launch, whose prologue has more to hide it under, measured no change and does not use it).  (One asm statement with its
// own wait inside: the compiler does not count the loads of an asm statement, and the destination registers must not be reused
// before the loads have landed.)
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch() {
    constexpr int LINES = (BYTES + 63) / 64;
    static_assert(LINES == 11 && BYTES >= 0x284, "the statement below reads eleven lines: exactly the argument block of the fused step kernel");
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
    asm volatile(
        "s_load_dword %0, %11, 0x0\n\ts_load_dword %1, %11, 0x40\n\ts_load_dword %2, %11, 0x80\n\ts_load_dword %3, %11, 0xc0\n\t"
        "s_load_dword %4, %11, 0x100\n\ts_load_dword %5, %11, 0x140\n\ts_load_dword %6, %11, 0x180\n\ts_load_dword %7, %11, 0x1c0\n\t"
        "s_load_dword %8, %11, 0x200\n\ts_load_dword %9, %11, 0x240\n\ts_load_dword %10, %11, 0x280\n\ts_waitcnt lgkmcnt(0)"
        : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10)
        : "s"(ka)
        : "memory");
}

// haloed image index of pixel p (0..63)
__device__ inline int hidx(int p) { return ((p >> 3) + 1) * 10 + (p & 7) + 1; }

// --------------------------------------------------------------------------------------------
// Implicit GEMM: acc[mt][i] += A(64 x K) * B(K x 16*NTW*4) for this wave's NTW column tiles.
//   A: LDS image, pixel stride `CP` elements (haloed when TAPS==9, plain [64] rows when !HALO)
//   B: global, packed [kstep][ntile_total][lane][8]
// K = TAPS*CIN, k-step = 32.
// All LDS accesses of the conv loop go through this symbol so that the compiler keeps them in the LDS address
// space (ds_read_b128); pointers carried through the tap loop degrade to flat loads, which also poison vmcnt.
extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

__device__ __forceinline__ bf16x8 lds_frag(int byte_off) { return *reinterpret_cast<const bf16x8*>(g_smem + byte_off); }

// 16-byte weight fragment through a wave-uniform buffer descriptor: voffset = lane*16 (loop invariant),
// soffset = scalar cursor, imm = column-tile offset -> zero vector ALU per load.
#ifdef SC_EXP
__constant__ int g_exp_wand = -1;  // experiment builds only: confine the weight stream to a small window (L1/L2-hot)
#endif
__device__ __forceinline__ bf16x8 wload(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#ifdef SC_EXP
    soff &= g_exp_wand;
#endif
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return __builtin_bit_cast(bf16x8, v);
}

// 1 x K vector times packed B: the vector sits in row 0 of the A tile.  The weight fragments are fetched by
// vec_w_load (early, so that the L2 round trip hides under other epilogue work) and consumed by vec_mma.
// x: LDS floats (rounded to bf16 on load).  Result for column tile i: lanes 0..15, element 0.
template <int K, int NTW>
struct VecW {
    bf16x8 w[K / 32][NTW];
};
template <int K, int NTW, int NT_TOTAL>
__device__ __forceinline__ void vec_w_load(VecW<K, NTW>& v, const bf16_t* __restrict__ Wp, int tile0, int lane) {
    const bf16x8* __restrict__ Wv = reinterpret_cast<const bf16x8*>(Wp) + (size_t)tile0 * 64 + lane;
#pragma unroll
    for (int s = 0; s < K / 32; s++)
#pragma unroll
        for (int i = 0; i < NTW; i++) v.w[s][i] = Wv[((size_t)s * NT_TOTAL + i) * 64];
}
// x: packed bf16 vector in LDS (16-byte aligned): one ds_read_b128 per k-step instead of 8 scalar reads + converts
template <int K, int NTW>
__device__ __forceinline__ void vec_mma(const bf16_t* x, const VecW<K, NTW>& v, int lane, f32x4 (&acc)[NTW]) {
    const int kq = lane >> 4;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
    for (int s = 0; s < K / 32; s++) {
        // all 16 rows of the A tile read the same vector (an LDS broadcast): every output row is the product, row 0 is used
        const u32x4 raw = *reinterpret_cast<const u32x4*>(x + s * 32 + 8 * kq);
        const bf16x8 a = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
        for (int i = 0; i < NTW; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, v.w[s][i], acc[i], 0, 0, 0);
    }
}

// Accumulator layout helpers.  acc[mt][i][r] is pixel row = mt*16 + (lane>>4)*4 + r and output
// channel = wave*(16*NTW) + (lane&15)*NTW + i  (the host packs B columns in that order so that a
// lane owns NTW ADJACENT channels: per-channel parameters load as one vector, activations store
// as one 8-byte LDS write).
template <int NTW>
__device__ inline int chan0(int wave, int lane) { return wave * (16 * NTW) + (lane & 15) * NTW; }

// --------------------------------------------------------------------------------------------
// value_head.ffn.0 (Linear 16391->128) over the batch: out[b][j] = sum_k hval[b][k] W[k][j], one 64-position x K/ksplit
// tile per workgroup (256 threads = 4 waves, wave w owns column tiles 2w, 2w+1); partial sums go to
// vpart[ksplit][b][128] and the search kernel's value tail reduces them in fixed order.  The tile is three device
// functions because it has two callers: k_value_fc1 (its own launch) and the tail of the fused step kernel
// (step_kernels.hip), which requests the weights BEFORE it waits for the other workgroups' feature rows.
// Split-K >= 64 (the launchers check): a tile is at most 8 k-steps and its weights fit one register batch.
constexpr int AUX_SC1 = 16;   // cache-policy bits of the gfx942/gfx950 buffer builtins: 1 = sc0, 2 = nt, 16 = sc1
struct Fc1W {
    bf16x8 b[8][4];
};
constexpr int FC1_TILE_LDS = 64 * (FC1_K / 64 + 8) * 2;   // bytes of the staged feature rows (33 KB)
// Wave w computes rows 32 (w >> 1) .. +31 x column tiles 4 (w & 1) .. +3 of the 64 x 128 tile: with each wave on 64 rows x 2 column
// tiles the four waves read every staged row four times (128 KB of LDS reads against 1 024 cycles of MFMA); 2 x 2 halves that
// (each output element is still the same chain of MFMAs on the same operands: the partial sums do not change by a bit).
__device__ __forceinline__ void fc1_wload(Fc1W& w, const NetDev& net, int ks, int ksplit, int wave, int lane) {
    const int steps = FC1_K / ksplit / 32;
    const bf16x8* Wv = reinterpret_cast<const bf16x8*>(net.wb + net.o_fc1) + ((size_t)(ks * steps) * 8 + (wave & 1) * 4) * 64 + lane;
    // the tile is a short dependent chain: all its weight loads are in flight before the first MFMA
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int s = u < steps ? u : steps - 1;
#pragma unroll
        for (int i = 0; i < 4; i++) w.b[u][i] = Wv[((size_t)s * 8 + i) * 64];
    }
}
// The 64 feature rows of the tile are the same for all four waves (each owns 32 of the 128 output columns): they go
// through LDS once -- [row][kchunk + 8] bf16, the 16-byte pad keeps the 16 rows of a fragment read on distinct banks --
// instead of four times through the CU's 64 B/clk vector-memory path (two thirds of the tile's operand traffic).
__device__ __forceinline__ void fc1_stage_a(bf16_t* s_a, const bf16_t* hval, int n_pos, int mb, int ks, int ksplit, int tid) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const int kchunk = FC1_K / ksplit;  // multiple of 32
    const int apitch = kchunk + 8;
    const int pieces = 64 * kchunk / 8;               // 16-byte pieces of the tile
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(hval) + (size_t)mb * 64 * FC1_K, 0, 0x7fffffff, 0x00020000);
    const int last = n_pos - 1 - mb * 64;             // rows past the batch repeat the last one (their results are not stored)
    for (int c = tid; c < pieces; c += 256) {
        const int r = c / (kchunk / 8), q = c % (kchunk / 8);
        const int row = r < last ? r : last;
        *reinterpret_cast<u32x4*>(s_a + r * apitch + q * 8) =
            __builtin_amdgcn_raw_buffer_load_b128(rsrc, (row * FC1_K + ks * kchunk + q * 8) * 2, 0, 0);
    }
}
// the same in two halves for split-K 64 and a full block of 64 rows (the fused step kernel, where the rows were published by
// other workgroups of the SAME launch with write-through stores): sc1 loads, past this CU's L1, into registers ...
__device__ __forceinline__ void fc1_load_a(__attribute__((ext_vector_type(4))) unsigned int (&a)[8], const bf16_t* hval, int mb, int ks, int tid) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(hval) + (size_t)mb * 64 * FC1_K, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int c = tid + 256 * k, r = c >> 5, q = c & 31;
        a[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r * FC1_K + ks * 256 + q * 8) * 2, 0, AUX_SC1);
    }
}
// ... and their way into the staging tile
__device__ __forceinline__ void fc1_put_a(bf16_t* s_a, const __attribute__((ext_vector_type(4))) unsigned int (&a)[8], int tid) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int c = tid + 256 * k, r = c >> 5, q = c & 31;
        *reinterpret_cast<u32x4*>(s_a + r * (256 + 8) + q * 8) = a[k];
    }
}
__device__ __forceinline__ void fc1_mma_store(const Fc1W& w, const bf16_t* s_a, float* vpart, int n_pos, int mb, int ks, int ksplit,
                                              int wave, int lane) {
    const int kchunk = FC1_K / ksplit, steps = kchunk / 32, apitch = kchunk + 8;
    const int row16 = lane & 15, kq = lane >> 4;
    const int mt0 = 2 * (wave >> 1), ct0 = 4 * (wave & 1);
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (u < steps) {
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(s_a + ((mt0 + m) * 16 + row16) * apitch + u * 32 + 8 * kq);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w.b[u][i], acc[m][i], 0, 0, 0);
            }
        }
    }
    // column tile t, lane column c (the host's packing, chan0<2>): output channel (t >> 1) * 32 + c * 2 + (t & 1)
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = mb * 64 + (mt0 + m) * 16 + (lane >> 4) * 4 + r;
            if (row < n_pos) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int c0 = ((ct0 >> 1) + j) * 32 + (lane & 15) * 2;
                    *reinterpret_cast<float2*>(vpart + ((size_t)ks * n_pos + row) * FC1_N + c0) = make_float2(acc[m][2 * j][r], acc[m][2 * j + 1][r]);
                }
            }
        }
}

#ifndef SC_NO_KERNELS
// grid = (ceil(n/64), KSPLIT)
__global__ __launch_bounds__(256) void k_value_fc1(Fc1Args A) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mb = blockIdx.x, ks = blockIdx.y;
    __shared__ __attribute__((aligned(16))) bf16_t s_a[FC1_TILE_LDS / 2];
    Fc1W w;
    fc1_wload(w, A.net, ks, A.ksplit, wave, lane);
    fc1_stage_a(s_a, A.hval, A.n_pos, mb, ks, A.ksplit, tid);
    __syncthreads();   // A tile staged
    fc1_mma_store(w, s_a, A.vpart, A.n_pos, mb, ks, A.ksplit, wave, lane);
}

// value head tail (py/module.py:95-106,147-149), one wave per position: the arithmetic is value_tail.hpp's, shared with the
// search kernel's fused tail (mcts_kernels.hpp: value_tail_issue / value_tail_finish), so `predict` and the value the
// search backs up are the same bits.
__global__ __launch_bounds__(64) void k_value_finish(VfinArgs A) {
    const int pos = blockIdx.x, lane = threadIdx.x;
    if (pos >= A.n_pos) return;
    const float* wf = A.net.wf;
    scvt::ValueTail t;
#pragma unroll
    for (int k = 0; k < 7; k++) t.meta[k] = A.meta[(size_t)pos * A.meta_stride + k];
    const int j = 2 * lane;   // lane owns output columns 2*lane, 2*lane+1
    const float* vp = A.vpart + (size_t)pos * FC1_N + j;
    const size_t vstride = (size_t)A.n_pos * FC1_N;
#pragma unroll
    for (int ks = 0; ks < 32; ks++) t.acc[ks] = *reinterpret_cast<const float2*>(vp + (size_t)ks * vstride);
    if (A.ksplit > 32) {
#pragma unroll
        for (int ks = 32; ks < 64; ks++) t.acc[ks] = *reinterpret_cast<const float2*>(vp + (size_t)ks * vstride);
    } else {
#pragma unroll
        for (int ks = 32; ks < 64; ks++) t.acc[ks] = make_float2(0.f, 0.f);
    }
    t.bias = *reinterpret_cast<const float2*>(wf + A.net.f_fc1b + j);
    t.w2 = *reinterpret_cast<const float2*>(wf + A.net.f_fc2w + j);
#pragma unroll
    for (int k = 0; k < 7; k++) t.wm[k] = *reinterpret_cast<const float2*>(wf + A.net.f_fc1m + k * FC1_N + j);
    t.fc2b = wf[A.net.f_fc2b];
    const float v = scvt::value_tail_compute(t, A.ksplit);
    if (lane == 0) A.value[pos] = v;
}

#endif  // SC_NO_KERNELS

}  // namespace scnn
