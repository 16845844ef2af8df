// nn_tower32.hpp -- the policy/value tower with CHANNELS on the MFMA row axis (v_mfma_f32_32x32x16_bf16).
//
// Same network, numerics contract and one-workgroup-per-position design as described in nn_kernels.hpp; what
// changes is the GEMM orientation:   D[channel][pixel] += W^T[channel][k] * X[k][pixel]
//   * A operand = weights, streamed L2 -> VGPR in 32x32x16 A-fragment order [k16 step][32-channel tile][lane][8]
//     (natural channel order, 1 KiB per wave-load);  B operand = the LDS image, one ds_read_b128 per 32 pixels.
//   * A k-step is 2*CT MFMAs of 32 cycles (CT = C/128 channel tiles per wave): compared with a 16x16x32 tiling the same
//     bytes and the same number of memory instructions ride under half as many, twice as long matrix instructions, so
//     the wave has 24 free issue cycles per MFMA (8 with 16x16x32) to place its weight loads and LDS reads.
//   * Accumulator layout: lane = pixel (lane&31 of a 32-pixel tile), registers = channels
//     (8*(r>>2) + 4*(lane>>5) + (r&3)).  LayerNorm over channels is therefore an in-register sum plus ONE
//     cross-lane add (lane ^ 32, v_permlane32_swap) and one LDS exchange between the 4 waves; a lane owns 4 ADJACENT
//     channels per register quad, so activations go to LDS as 8-byte bf16 stores.  Only the squeeze-excitation
//     average pool reduces across lanes (halving butterfly, once per block).
//   * Epilogues are VALU-issue-bound at one wave per SIMD: packed fp32 math on register pairs, pair-wise bf16
//     converts, permlane swaps / DPP instead of LDS-crossbar shuffles (see the helpers below).
//   * Per-channel parameters of a block are fetched cooperatively and staged in LDS (a per-lane fetch costs a full
//     1 KiB wave-load on the texture path); the fp32 residual stream stays in registers; the taps of a conv are
//     unrolled (all 9 at C = 128) around a weight ring that is carried from layer to layer.
//   * Pixel tiles are chosen for the LDS banks: tile t holds ranks {2t, 2t+4} in ds_read_b128 lane group A and
//     {2t+1, 2t+5} in group B; with a pixel stride of an odd multiple of 16 B the 16 haloed addresses of every
//     lane group fall on 16 distinct 16-byte bank slots for every 3x3 tap.
#pragma once
#include <type_traits>

#include "nn_kernels.hpp"

namespace scnn {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// --------------------------------------------------------------------------------------------
// Precision policies of the conv GEMMs (stem, residual blocks, the three head convs).  Everything else -- fp32
// accumulators, LayerNorm, residual stream, the bf16 squeeze-excitation and value FC layers, the fp32 softmax -- is shared.
//   PrecBF16  operands bf16, v_mfma_f32_32x32x16_bf16 (K = 16 per instruction, 32 cycles)
//   PrecFP8   operands OCP e4m3 (BASELINE configs[4]), v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per instruction, 64
//             cycles: twice the MACs per cycle).  The block scales of the instruction carry the per-output-channel
//             power-of-two weight scales (E8M0, one byte per lane = per row of the A tile) for free; the image operand's
//             scale is 2^0.  Lane maps pinned with exact data by tools/experiments/mfma_fp8_layout.hip: lane l holds
//             row/column l & 31, k = 32 (l >> 5) .. + 31, one scale byte per lane applies to its row.
//             Activations are clamped to +-448 before v_cvt_pk_fp8_f32 (which overflows to NaN, not to the maximum).
struct PrecBF16 {
    static constexpr bool FP8 = false;
    static constexpr int EB = 2;     // bytes per image / weight element
    static constexpr int KS = 16;    // k per MFMA
    static constexpr int FB = 16;    // fragment bytes per lane (one ds_read_b128 / one 16-byte weight load)
    typedef bf16x8 frag;
};
struct PrecFP8 {
    static constexpr bool FP8 = true;
    static constexpr int EB = 1;
    static constexpr int KS = 64;
    static constexpr int FB = 32;    // two 16-byte halves: image bytes [32h, 32h + 32) of the k-step, weights [half][lane][16]
    typedef i32x8 frag;
};
// haloed image: pixel stride in bytes -- an odd multiple of 16 B at both precisions (272 / 528 B bf16, 144 / 272 B fp8)
template <class P>
constexpr int pix_stride(int channels) { return channels * P::EB + 16; }

template <class P>
__device__ __forceinline__ typename P::frag lds_frag_p(int byte_off) {
    if constexpr (P::FP8) {
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 a = *reinterpret_cast<const i32x4*>(g_smem + byte_off), b = *reinterpret_cast<const i32x4*>(g_smem + byte_off + 16);
        return i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    } else {
        return lds_frag(byte_off);
    }
}
// one weight fragment of channel tile ct (voff = lane * 16, soff = scalar cursor): 64 * FB bytes per tile
template <class P>
__device__ __forceinline__ typename P::frag wload_p(__amdgpu_buffer_rsrc_t rsrc, int voff, int ct, int soff) {
    if constexpr (P::FP8) {
        const bf16x8 a = wload(rsrc, voff + ct * 2048, soff), b = wload(rsrc, voff + ct * 2048 + 1024, soff);
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 x = __builtin_bit_cast(i32x4, a), y = __builtin_bit_cast(i32x4, b);
        return i32x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    } else {
        return wload(rsrc, voff + ct * 1024, soff);
    }
}
template <class P>
__device__ __forceinline__ f32x16 mma_p(const typename P::frag& a, const typename P::frag& b, const f32x16& c, int scale_a) {
    if constexpr (P::FP8) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, 127);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// four adjacent channels of one pixel -> the image (8 bytes bf16 / 4 bytes e4m3), optionally through ReLU
__device__ __forceinline__ uint32_t pk_fp8x4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}

// experiment builds (-DSC_EXP, tools/build_exp.sh): per-wave cycle stamps of the block phases, dumped with dbg_stage 2000
#ifdef SC_EXP
struct Stamp {
    bool on;
    long long prev;
    long long t[40];
    __device__ __forceinline__ void start() { if (on) prev = clock64(); }
    __device__ __forceinline__ void mark(int k) {
        if (on && k >= 0) {
            long long now = clock64();
            t[k] += now - prev;
            prev = now;
        }
    }
};
#define SC_STAMP_ARG , Stamp &stampv, int sk0
#define SC_STAMP_PASS(k) , stampv, (k)
#define SC_MARK(k) stampv.mark(k)
#else
#define SC_STAMP_ARG
#define SC_STAMP_PASS(k)
#define SC_MARK(k)
#endif

// GEMM pixel (tile pt, lane-in-tile i) -> board pixel (rank*8 + file); see the header comment
__device__ inline int gpix2board(int pt, int i) {
    const bool inA = (i < 4) || (i >= 12 && i < 16) || (i >= 20 && i < 28);
    const int a = inA ? (i < 4 ? i : (i < 16 ? i - 8 : i - 12)) : (i < 12 ? i - 4 : (i < 20 ? i - 8 : i - 16));
    return ((2 * pt + (inA ? 0 : 1) + 4 * (a >> 3)) << 3) | (a & 7);
}

// Packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth of work per issue slot) and packed bf16 conversion.
// With one wave per SIMD the epilogues are bound by VALU issue, and the compiler only forms these from explicit
// 2-vectors (scalar code got v_fma_f32 per element and a shuffle after every v_cvt_pk_bf16_f32).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16pair __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <typename V>
__device__ __forceinline__ f32x2 pair(const V& v, int k) { return f32x2{v[k], v[k + 1]}; }
__device__ __forceinline__ f32x2 splat(float x) { return f32x2{x, x}; }
__device__ __forceinline__ uint32_t pk_bf16(f32x2 y) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(y, bf16pair)); }
// ReLU after rounding, on the bf16 pair as two int16 (negative floats are negative integers; -0 becomes +0)
__device__ __forceinline__ uint32_t pk_bf16_relu(f32x2 y) {
    s16x2 s = __builtin_bit_cast(s16x2, __builtin_convertvector(y, bf16pair));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(s, s16x2{0, 0}));
}

template <class P, bool RELU>
__device__ __forceinline__ void store4(int byte_off_row, int chan, f32x2 lo, f32x2 hi) {
    if constexpr (P::FP8) {
        const float l = RELU ? 0.f : -448.f;   // med3 = clamp (and ReLU): e4m3 has no infinity, the convert does not saturate
        *reinterpret_cast<uint32_t*>(g_smem + byte_off_row + chan) =
            pk_fp8x4(__builtin_amdgcn_fmed3f(lo.x, l, 448.f), __builtin_amdgcn_fmed3f(lo.y, l, 448.f),
                     __builtin_amdgcn_fmed3f(hi.x, l, 448.f), __builtin_amdgcn_fmed3f(hi.y, l, 448.f));
    } else {
        *reinterpret_cast<uint2*>(g_smem + byte_off_row + chan * 2) =
            RELU ? make_uint2(pk_bf16_relu(lo), pk_bf16_relu(hi)) : make_uint2(pk_bf16(lo), pk_bf16(hi));
    }
}

// first channel of register quad g (registers 4g..4g+3) of channel tile ct
template <int CT>
__device__ __forceinline__ int chan32(int wave, int ct, int g, int h) { return wave * (32 * CT) + ct * 32 + 8 * g + 4 * h; }

// per-channel parameters in accumulator layout (a lane's 16*CT channels)
template <int CT>
struct ChP {
    f32x4 v[CT][4];
};
template <int CT>
__device__ __forceinline__ void ch_load(ChP<CT>& P, const float* __restrict__ p, int wave, int h) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int g = 0; g < 4; g++) P.v[ct][g] = *reinterpret_cast<const f32x4*>(p + chan32<CT>(wave, ct, g, h));
}
// same, from a parameter array staged in LDS (byte offset inside g_smem)
template <int CT>
__device__ __forceinline__ void ch_load_lds(ChP<CT>& P, int byte_off, int wave, int h) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int g = 0; g < 4; g++) P.v[ct][g] = *reinterpret_cast<const f32x4*>(g_smem + byte_off + chan32<CT>(wave, ct, g, h) * 4);
}
// the conv bias is the accumulator's initial value
template <int CT>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[CT][2], const ChP<CT>& B) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ct][pt][r] = B.v[ct][r >> 2][r & 3];
}

// --------------------------------------------------------------------------------------------
// Implicit GEMM, k-step 16.  acc[ct][pt] += W^T(32 channels x K) * X(K x 32 pixels).
//   px[pt]: byte offset (inside the image at xoff) of this lane's pixel of tile pt, including its 16-byte k half
//   weights: [K/16][TILES][64][8] bf16; this wave's tiles are wave*CT .. wave*CT+CT-1
// Ring / carry semantics as conv_mma (nn_kernels.hpp): RS slots of one k-step each, loads run RS-1 steps ahead,
// with PRE the first RS-1 slots were filled by the previous layer's loop (its last prefetches go to `next_first`,
// a byte offset relative to ITS weights).  AB image-fragment buffers, reads AB-1 steps ahead.
#ifndef SC_T32_AB
#define SC_T32_AB 4
#endif
//   PSB: pixel stride of the image in bytes; sa: E8M0 weight scales of this wave's channel tiles (fp8 only)
template <class P, int CIN, int TAPS, int CT, int TILES, int PSB, int RS, int TPI, bool PRE, int AB = SC_T32_AB>
__device__ __forceinline__ void conv_mma32(int xoff, const bf16_t* __restrict__ Wp, int wave_u, int lane, const int (&px)[2],
                                           f32x16 (&acc)[CT][2], typename P::frag (&bq)[RS][CT], int next_first, const int (&sa)[CT]) {
    typedef typename P::frag frag;
    constexpr int KPT = CIN / P::KS;       // k-steps per tap
    constexpr int SPG = KPT * TPI;         // k-steps per loop iteration (tap group)
    constexpr int NG = TAPS / TPI;
    constexpr int TB = 64 * P::FB;         // bytes of one tile's fragment (a wave-load, or two at fp8)
    constexpr int SBB = TILES * TB;        // bytes per k-step of packed weights
    constexpr int KSB = P::KS * P::EB;     // image bytes per k-step
    constexpr int PD = RS - 1, AD = AB - 1;
    static_assert(TAPS % TPI == 0 && SPG % RS == 0 && PD < SPG, "bad ring / tap-group geometry");
    static_assert(SPG % AB == 0 && AD <= SPG, "bad image buffer geometry");
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(Wp) + (size_t)wave_u * CT * (TB / 2), 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    auto toffb = [](int t) { return (TAPS == 9) ? ((((t * 11) >> 5) - 1) * 10 + (t - 3 * ((t * 11) >> 5)) - 1) * PSB : 0; };
    frag xq[AB][2];
    int wcur = 0;
    int pc[TPI][2];
#pragma unroll
    for (int tl = 0; tl < TPI; tl++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++) pc[tl][pt] = xoff + px[pt] + toffb(tl);
    if (!PRE) {
#pragma unroll
        for (int st = 0; st < PD; st++)
#pragma unroll
            for (int ct = 0; ct < CT; ct++) bq[st][ct] = wload_p<P>(rsrc, voff, ct, st * SBB);
    }
#pragma unroll
    for (int v = 0; v < AD; v++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++) xq[v][pt] = lds_frag_p<P>(pc[v / KPT][pt] + (v % KPT) * KSB);
#pragma unroll 1
    for (int j = 0; j < NG; j++) {
        const int tn = (j + 1 == NG) ? 0 : j + 1;
        const int wnext = (j == NG - 1) ? next_first : tn * (SPG * SBB);
        int pn[TPI][2];
#pragma unroll
        for (int tl = 0; tl < TPI; tl++)
#pragma unroll
            for (int pt = 0; pt < 2; pt++) pn[tl][pt] = xoff + px[pt] + toffb(tn * TPI + tl);
#pragma unroll
        for (int u = 0; u < SPG; u++) {
            const int slot = u % RS;
#ifndef SC_EXP_NOWLOAD   // experiment builds: the loop without its weight stream / without its image reads
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
                bq[(slot + PD) % RS][ct] = (u + PD < SPG) ? wload_p<P>(rsrc, voff, ct, wcur + (u + PD) * SBB)
                                                          : wload_p<P>(rsrc, voff, ct, wnext + (u + PD - SPG) * SBB);
#endif
#ifndef SC_EXP_NOLDS
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int v = u + AD;
                xq[v % AB][pt] = (v < SPG) ? lds_frag_p<P>(pc[v / KPT][pt] + (v % KPT) * KSB)
                                           : lds_frag_p<P>(pn[(v - SPG) / KPT][pt] + ((v - SPG) % KPT) * KSB);
            }
#endif
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int pt = 0; pt < 2; pt++)
                    acc[ct][pt] = mma_p<P>(bq[slot][ct], xq[u % AB][pt], acc[ct][pt], sa[ct]);
            // one memory instruction behind each MFMA (two at fp8: its fragments are two 16-byte halves); the fence
            // keeps every prefetch in the step it was written in
#pragma unroll
            for (int m = 0; m < 2 * CT; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (P::FP8) {
                    // per k-step: 2 CT weight loads and 4 image reads ride under 2 CT MFMAs of 64 cycles
                    __builtin_amdgcn_sched_group_barrier(0x100, CT == 1 ? 2 : 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                } else if (CT == 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (m == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                } else {
                    if ((m & 1) == 0)
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    else
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wcur = wnext;
#pragma unroll
        for (int tl = 0; tl < TPI; tl++)
#pragma unroll
            for (int pt = 0; pt < 2; pt++) pc[tl][pt] = pn[tl][pt];
    }
}

// LayerNorm over `count` channels (eps 1e-6, timm LayerNorm2d) + optional ReLU, in place; the bias is already in
// the accumulators.  Two halves: ln_reduce (per-pixel statistics: in-register sums, one lane^32 exchange, one LDS
// exchange between the 4 waves, ONE barrier) and ln_apply.  s_stat2: two alternating [4 waves][64 pixels] float2
// buffers: the buffer written here was last read two LayerNorms ago, and every wave has passed the barrier of the
// LayerNorm in between since then.  Pixels are indexed by GEMM pixel (pt*32 + lane&31).
struct LnStat {
    float rstd[2], nm[2];   // per pixel tile: 1/sigma and -mean/sigma
};
// ReLU is a single v_max_f32 (which maps NaN to 0, unlike torch.relu): a NaN anywhere in the network reaches some
// LayerNorm's variance first, so ln_reduce records it in `bad` and the kernel poisons its outputs at the end --
// non-finite results stay visible to the caller (reference: warning at src/backends/torch.rs:129-135).
template <int CT>
__device__ inline void ln_reduce(const f32x16 (&acc)[CT][2], LnStat& L, int count, int wave, int lane, float* s_stat2,
                                 int& parity, int& bad SC_STAMP_ARG) {
    float2* st = reinterpret_cast<float2*>(s_stat2) + (parity & 1) * 256;
    parity ^= 1;
    const int i = lane & 31;
    float s[2], q[2];
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        f32x2 a2 = {0.f, 0.f}, b2 = {0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v = pair(acc[ct][pt], r);
                a2 += v;
                b2 = v * v + b2;
            }
        s[pt] = a2.x + a2.y;
        q[pt] = b2.x + b2.y;
    }
    // The two channel halves (lane, lane^32) are added with v_permlane32_swap (gfx950): swapping the upper half of
    // tile 0's sums with the lower half of tile 1's leaves, after one add, tile 0's pixel sums in lanes 0..31 and
    // tile 1's in lanes 32..63 -- GEMM pixel = lane, one 8-byte store per lane, no trip through the LDS crossbar.
    {
        const auto rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(s[0]), __float_as_uint(s[1]), false, false);
        const auto rq = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[0]), __float_as_uint(q[1]), false, false);
        st[wave * 64 + lane] = make_float2(__uint_as_float(rs[0]) + __uint_as_float(rs[1]), __uint_as_float(rq[0]) + __uint_as_float(rq[1]));
    }
    SC_MARK(sk0);
    __syncthreads();
    SC_MARK(sk0 + 1);
    const float inv = 1.0f / (float)count;
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        const int gp = pt * 32 + i;
        float2 a0 = st[gp], a1 = st[64 + gp], a2 = st[128 + gp], a3 = st[192 + gp];
        float S = (a0.x + a1.x) + (a2.x + a3.x);
        float Q = (a0.y + a1.y) + (a2.y + a3.y);
        float mean = S * inv;
        float var = Q * inv - mean * mean;
        bad |= (var != var) ? 1 : 0;
        var = fmaxf(var, 0.f);
        L.rstd[pt] = __frsqrt_rn(var + 1e-6f);
        L.nm[pt] = -mean * L.rstd[pt];
    }
}
template <int CT>
__device__ __forceinline__ void ln_apply(f32x16 (&acc)[CT][2], const LnStat& L, const ChP<CT>& G, const ChP<CT>& E, bool relu) {
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 t = pair(acc[ct][pt], r) * splat(L.rstd[pt]) + splat(L.nm[pt]);
                const f32x2 y = t * pair(G.v[ct][r >> 2], r & 3) + pair(E.v[ct][r >> 2], r & 3);
                acc[ct][pt][r] = relu ? fmaxf(y.x, 0.f) : y.x;
                acc[ct][pt][r + 1] = relu ? fmaxf(y.y, 0.f) : y.y;
            }
}
// LayerNorm scale/shift + ReLU straight into the bf16 image (the fp32 values are not needed again)
template <class P, int CT>
__device__ __forceinline__ void ln_apply_relu_store(const f32x16 (&acc)[CT][2], const LnStat& L, const ChP<CT>& G, const ChP<CT>& E,
                                                    const int (&pixbase)[2], int wave, int h) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x2 y[2];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const f32x2 t = pair(acc[ct][pt], 4 * g + 2 * hf) * splat(L.rstd[pt]) + splat(L.nm[pt]);
                    y[hf] = t * pair(G.v[ct][g], 2 * hf) + pair(E.v[ct][g], 2 * hf);
                }
                store4<P, true>(pixbase[pt], chan32<CT>(wave, ct, g, h), y[0], y[1]);
            }
}
// parameters already in registers (stem, heads)
template <int CT>
__device__ inline void layernorm32(f32x16 (&acc)[CT][2], const ChP<CT>& G, const ChP<CT>& E, int count, bool relu, int wave,
                                   int lane, float* s_stat2, int& parity, int& bad SC_STAMP_ARG) {
    LnStat L;
    ln_reduce<CT>(acc, L, count, wave, lane, s_stat2, parity, bad SC_STAMP_PASS(sk0));
    ln_apply<CT>(acc, L, G, E, relu);
}

// accumulators -> bf16 image: pixbase[pt] = byte offset of the lane's pixel row inside g_smem, 4 adjacent channels
// per 8-byte store
template <class P, int CT>
__device__ inline void store_image32(const f32x16 (&acc)[CT][2], const int (&pixbase)[2], int wave, int h) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                store4<P, false>(pixbase[pt], chan32<CT>(wave, ct, g, h), pair(acc[ct][pt], 4 * g), pair(acc[ct][pt], 4 * g + 2));
}

// Halving butterfly over the 32 lanes of a half-wave (squeeze-excitation average pool).  Level LVL pairs every lane
// with a partner whose bit (16 >> LVL) differs: lane^16 through the LDS crossbar, then lane^15 (row_mirror),
// lane^7 (row_half_mirror), lane^2 and lane^1 (quad_perm) as DPP moves -- five independent XOR masks, so every
// lane's contribution reaches every sum exactly once.  Lanes with the bit set keep the upper W values (W = 0: a
// single value is left, plain exchange-add).  Lane j of a half ends with value index j (32 values) or j>>1 (16).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) { return dpp_f<CTRL>(x); }
template <int LVL>
__device__ __forceinline__ float pool_xchg(float x) {
    if constexpr (LVL == 1) return dpp_mov<0x140>(x);   // row_mirror
    else if constexpr (LVL == 2) return dpp_mov<0x141>(x);   // row_half_mirror
    else if constexpr (LVL == 3) return dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]
    else return dpp_mov<0xB1>(x);                            // quad_perm [1,0,3,2]
}
template <int NV, int W, int LVL>
__device__ __forceinline__ void pool_level(float (&v)[NV], int lane) {
    if constexpr (LVL < 5) {
        if constexpr (W >= 1) {
            if constexpr (LVL == 0) {
                // lane^16: v_permlane16_swap (gfx950) swaps the odd 16-lane rows of v[t] with the even rows of v[t+W];
                // the sum of the two results is value t (both lanes' parts) in even rows and value t+W in odd rows
#pragma unroll
                for (int t = 0; t < W; t++) {
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[t]), __float_as_uint(v[t + W]), false, false);
                    v[t] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                }
            } else {
                const bool hi = (lane & (16 >> LVL)) != 0;
#pragma unroll
                for (int t = 0; t < W; t++) {
                    float send = hi ? v[t] : v[t + W], keep = hi ? v[t + W] : v[t];
                    v[t] = keep + pool_xchg<LVL>(send);
                }
            }
            pool_level<NV, W / 2, LVL + 1>(v, lane);
        } else {
            v[0] += pool_xchg<LVL>(v[0]);
            pool_level<NV, 0, LVL + 1>(v, lane);
        }
    }
}

// first RS-1 k-steps of a layer's weights into its ring, ahead of the conv_mma32<.., PRE = true> that consumes them
template <class P, int CT, int TILES, int RS>
__device__ __forceinline__ void ring_fill(typename P::frag (&bq)[RS][CT], const bf16_t* __restrict__ Wp, int wave_u, int lane) {
    constexpr int TB = 64 * P::FB;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(Wp) + (size_t)wave_u * CT * (TB / 2), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int st = 0; st < RS - 1; st++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) bq[st][ct] = wload_p<P>(rsrc, lane * 16, ct, st * TILES * TB);
}
// E8M0 scale bytes of this wave's channel tiles for conv `conv_idx` (fp8): [conv][8 tiles][64 lanes] ints behind the
// fp32 parameters (weights.hpp); requested one conv ahead of their use
template <class P, int CT>
__device__ __forceinline__ void scale_load(int (&sa)[CT], const NetDev& net, int conv_idx, int wave, int lane) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
        sa[ct] = P::FP8 ? __builtin_bit_cast(int, net.wf[net.f_scales + (size_t)((conv_idx * 8 + wave * CT + ct) * 64 + lane)]) : 127;
}

// dynamic LDS of a tower workgroup: the haloed image (later the 4672 policy logits), the policy head's image, LayerNorm /
// SE scratch and the staged per-block parameters.  The e4m3 images are half the size: 49 KB instead of 73 KB at C = 128,
// which is what lets two fused step workgroups (10 KB of search scratch each) share a CU.
constexpr int tower32_lds_bytes(int C, bool fp8 = false) {
    const int psb = C * (fp8 ? 1 : 2) + 16, hpsb = HEAD * (fp8 ? 1 : 2) + 16;
    const int xa = 100 * psb > 4864 * 4 ? 100 * psb : 4864 * 4;
    return xa + 64 * hpsb + 4096 + 3072 + 1024 + 64 + 15 * C * 2;
}

// tower_body: the network for position `pos`, run by the 256 threads of one workgroup.  planes_lds: the position's input
// planes int8[64][112] in LDS, or nullptr: read them from A.boards.
// Pre (fused step kernel, step_kernels.hip): work that waves 0 and 1 do BEFORE the network -- pre(0) is the game's tree
// search and says whether the leaf needs the network at all, pre(1) its helper; between them they leave the planes in
// planes_lds.  Called after every wave has requested its first weights and the stem parameters, and while waves 2 and 3
// zero the image: the tower's cold prologue runs under the search instead of after it.
struct NoPre {
    __device__ __forceinline__ bool operator()(int) const { return true; }
};
// value_head.ffn.0 inside the fused step launch (step_kernels.hip: fc1_tail; HAND && A.fc1_arrive): what the tower requests
// ahead of the tail, under its softmax -- the tile's weights and a first reading of the block's arrival counter
struct Fc1Hand {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    Fc1W w;
    uint32_t early;   // lane 0 of wave 0: a reading of the block's arrival counter taken under the softmax
    int have_a;       // the first reading (taken under the last policy conv) already showed the block complete:
    u32x4 a[8];       // ... this thread's 8 pieces of the 64 feature rows, requested under the softmax too
};
struct NoHand {};
template <class P, int C, int RS, int TPI, int AB = SC_T32_AB, class Pre = NoPre, class Hand = NoHand>
__device__ __forceinline__ bool tower_body(const TowerArgs& A, const int pos, const int8_t* planes_lds, Pre pre, Hand& fhr) {
    constexpr bool FUSED = !std::is_same<Pre, NoPre>::value;
    constexpr bool HAND = std::is_same<Hand, Fc1Hand>::value;   // value_head.ffn.0 inside this launch is possible (A.fc1_arrive says)
    Hand* const fh = &fhr;
    typedef typename P::frag frag;
    constexpr int CT = C / 128;        // 32-channel tiles per wave in the trunk
    constexpr int TILES = C / 32;
    constexpr int PSB = pix_stride<P>(C);      // image pixel stride in bytes: an odd multiple of 16 B
    constexpr int HPSB = pix_stride<P>(HEAD);
    constexpr int EB = P::EB;
    constexpr int HRS = P::FP8 ? C / 64 : 8, HAB = P::FP8 ? 2 : SC_T32_AB;         // 256-wide head convs (K = C)
    constexpr int H2RS = P::FP8 ? HEAD / 64 : 16, H2AB = P::FP8 ? 2 : SC_T32_AB;   // the 73-wide one (K = 256)
    constexpr int NTW = C / 64;        // 16-column tiles per wave of the SE layers (16x16x32 vector products)
    constexpr int XA_BYTES = (100 * PSB > 4864 * 4) ? 100 * PSB : 4864 * 4;   // image; later the 4672 policy logits
    constexpr int RS_BYTES = 64 * HPSB;   // the policy head's image
    unsigned char* smem = g_smem;
    float* s_stat = reinterpret_cast<float*>(smem + XA_BYTES + RS_BYTES);   // 2 x [4][64] float2
    float* s_vec = s_stat + 1024;                                           // SE vectors (packed bf16)
    float* s_scl = s_vec + 768;                                             // [C] SE scales
    float* s_red = s_scl + 256;                                             // [8]
    constexpr int PAR_OFF = XA_BYTES + RS_BYTES + 4096 + 3072 + 1024 + 64;  // staged per-block parameters, 7.5*C floats
    bf16_t* s_xb = reinterpret_cast<bf16_t*>(s_vec);
    float* s_z = reinterpret_cast<float*>(smem);                            // policy logits [4672], aliases Xa (after the trunk)
    static_assert(4864 * 4 <= XA_BYTES, "policy logits (and the softmax pass that reads 19 x 256 of them) must fit in the image area");
    static_assert(XA_BYTES % 16 == 0 && RS_BYTES % 16 == 0, "LDS regions must stay 16-byte aligned");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const NetDev& net = A.net;
#ifdef SC_EXP
    // stage 2001 (tools/dbg_clock.py): shader-clock and wall-clock stamps around the whole tower of this workgroup
    const bool clk_on = A.dbg && A.dbg_stage == 2001 && tid == 0;
    const long long clk_t0 = clk_on ? clock64() : 0;
    const unsigned long long clk_r0 = clk_on ? __builtin_amdgcn_s_memrealtime() : 0;
    Stamp stampv;
    stampv.on = A.dbg && A.dbg_stage == 2000 && lane == 0;
    stampv.prev = 0;
    for (int k = 0; k < 40; k++) stampv.t[k] = 0;
    stampv.start();
#endif
    const int bp[2] = {gpix2board(0, i32), gpix2board(1, i32)};            // this lane's two board pixels
    const int pixbase[2] = {hidx(bp[0]) * PSB, hidx(bp[1]) * PSB};         // their rows in the haloed image (bytes)
    const int px[2] = {pixbase[0] + h * P::FB, pixbase[1] + h * P::FB};    // + this lane's k half

    frag ring[RS][CT];     // weight prefetch ring, carried from layer to layer
    int sa[CT], san[CT];   // fp8: E8M0 weight scales of the current / the next conv (conv index: 0 stem, 1 + 2b / 2 + 2b
                           // the convs of block b, then value conv, policy conv1, policy conv2)
    scale_load<P, CT>(sa, net, 0, wave, lane);
    scale_load<P, CT>(san, net, net.n_blocks > 0 ? 1 : 1 + 2 * net.n_blocks, wave, lane);
    ChP<CT> Bn;            // bias of the NEXT conv (requested one epilogue early)
    constexpr int PAR0 = XA_BYTES + RS_BYTES + 4096 + 3072 + 1024 + 64;   // = PAR_OFF below
    // ---- Every global read of the prologue is issued first: the input planes (7 dwords per thread) and, cooperatively,
    // the stem's bias / gamma / beta plus block 0's first bias (4*C floats, one 16-byte load per thread, staged in
    // LDS like the per-block parameters).  Their trip from HBM / L2 overlaps the zero fill instead of following it.
    const int p_in = tid >> 2, q_in = tid & 3;
    uint32_t win[7];
    if (!FUSED && !planes_lds) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(A.boards + (size_t)pos * 7168 + p_in * 112 + q_in * 28);
#pragma unroll
        for (int k = 0; k < 7; k++) win[k] = src[k];
    }
    static_assert(C <= 256, "one parameter vector per thread");
    f32x4 spv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 3 * C / 4) spv = *reinterpret_cast<const f32x4*>(net.wf + net.f_stem + tid * 4);
    else if (tid < C) spv = *reinterpret_cast<const f32x4*>(net.wf + net.f_blocks + (tid - 3 * C / 4) * 4);
    // the stem's first weights too -- except for the fused kernel's search wave: its first loads are the search's (a wave's
    // loads return in order: the control block would queue behind 12 KB of weights); it fills its ring after the search
#ifndef SC_FUSED_RING_LATE
#define SC_FUSED_RING_LATE 1   // experiment builds: 0 = the search wave also requests its weights first
#endif
    if (!FUSED || wave != 0 || !SC_FUSED_RING_LATE) ring_fill<P, CT, TILES, RS>(ring, net.wb + net.o_stem, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    // ---- zero the image (halo stays zero for the whole kernel), then write the 112 input planes
    if constexpr (FUSED) {
        __shared__ int s_go;
        if (wave <= 1) {   // wave 0: the search; wave 1: its helper (plane encoding)
            const bool go = pre(wave);
            if (wave == 0 && lane == 0) s_go = go ? 1 : 0;
            if (wave == 0 && SC_FUSED_RING_LATE) ring_fill<P, CT, TILES, RS>(ring, net.wb + net.o_stem, wave, lane);
        } else {
            uint4* z = reinterpret_cast<uint4*>(smem);
            for (int k = tid - 128; k < 100 * PSB / 16; k += 128) z[k] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();   // image zeroed; planes in planes_lds, legal moves / indices stored (the barrier waits for wave 0's stores)
        if (!s_go) return false; // terminal leaf or idle slot: no network evaluation this step
    } else {
        uint4* z = reinterpret_cast<uint4*>(smem);
        for (int k = tid; k < 100 * PSB / 16; k += 256) z[k] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    if (FUSED || planes_lds) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(planes_lds + p_in * 112 + q_in * 28);
#pragma unroll
        for (int k = 0; k < 7; k++) win[k] = src[k];
    }
    {
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem + hidx(p_in) * PSB + q_in * 28 * EB);   // 28-plane groups: 4-byte aligned
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const uint32_t w = win[k];
            float v[4];
#pragma unroll
            for (int b = 0; b < 4; b++) v[b] = (float)(int8_t)((w >> (8 * b)) & 0xff);
            if constexpr (P::FP8) {
                dst[k] = pk_fp8x4(v[0], v[1], v[2], v[3]);   // plane values are 0 / 1: exact
            } else {
                dst[2 * k] = pk_bf16(f32x2{v[0], v[1]});
                dst[2 * k + 1] = pk_bf16(f32x2{v[2], v[3]});
            }
        }
        if (tid < C) *reinterpret_cast<f32x4*>(g_smem + PAR0 + tid * 16) = spv;
    }
    __syncthreads();

    f32x16 acc[CT][2];
    int ln_parity = 0;
    int bad = 0;   // a LayerNorm saw a NaN variance
    // The fp32 residual stream stays in registers (16 values per 32-channel tile and pixel tile; at C = 256 the
    // compiler parks part of it in AGPRs): measured faster than a round trip through LDS at both widths.
    f32x16 res[CT][2];
    auto store_res = [&]() {
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int pt = 0; pt < 2; pt++) res[ct][pt] = acc[ct][pt];
    };
    auto dump = [&](int stage) {
        if constexpr (FUSED) return;   // (the fused step kernel has no debug output: sc_forward_debug runs the stand-alone tower)
        if (A.dbg && A.dbg_stage == stage) {
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int pt = 0; pt < 2; pt++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        A.dbg[((size_t)pos * 64 + bp[pt]) * C + chan32<CT>(wave, ct, r >> 2, h) + (r & 3)] = acc[ct][pt][r];
        }
    };

    SC_MARK(38);   // prologue: input planes, zero fill, parameter staging
    // ---- conv_block (py/module.py:120-126): conv3x3 112->C (K padded to 128/tap), LN, ReLU
    {
        ch_load_lds<CT>(Bn, PAR0, wave, h);
        acc_init<CT>(acc, Bn);
        const bf16_t* w0 = net.wb + net.o_stem;
        SC_MARK(33);
        conv_mma32<P, 128, 9, CT, TILES, PSB, RS, TPI, true, AB>(0, w0, wave, lane, px, acc, ring, (int)((net.wb + net.o_blocks) - w0) * 2, sa);
        ChP<CT> G, E;
        ch_load_lds<CT>(G, PAR0 + C * 4, wave, h);
        ch_load_lds<CT>(E, PAR0 + 2 * C * 4, wave, h);
        ch_load_lds<CT>(Bn, PAR0 + 3 * C * 4, wave, h);   // read before the barriers that let block 0 restage the area
        SC_MARK(34);
        layernorm32<CT>(acc, G, E, C, true, wave, lane, s_stat, ln_parity, bad SC_STAMP_PASS(35));
        SC_MARK(37);
    }
    store_res();
    store_image32<P, CT>(acc, pixbase, wave, h);  // every wave passed the LN barrier: the input image is dead
    __syncthreads();
    SC_MARK(14);
    dump(0);
    const int head_conv0 = 1 + 2 * net.n_blocks;   // conv index of the value conv

    // ---- residual tower (ResBlockSE.forward, py/module.py:38-46)
#pragma unroll 1
    for (int b = 0; b < net.n_blocks; b++) {
        const bf16_t* wb = net.wb + net.o_blocks + (size_t)b * net.blk_stride_b;
        const float* wf = net.wf + net.f_blocks + (size_t)b * net.blk_stride_f;
        constexpr int NT1 = C / 32;                       // 16-column tiles of SE fc1
        constexpr int NTW1 = NT1 / 4;
        // The block's per-channel parameters -- everything from conv1's LayerNorm to the NEXT block's conv1 bias,
        // 7.5*C contiguous floats -- are fetched cooperatively (one or two 16-byte loads per thread) under conv1 and
        // staged in LDS behind conv1's LayerNorm barrier: per-lane parameter fetches straight from L2 cost a full
        // 1 KiB wave-load through the texture path each, on the critical path of every epilogue.
        constexpr int PAR_V4 = 15 * C / 8;                // float4 count of the window
        constexpr int NPV = (PAR_V4 + 255) / 256;
        static_assert(PAR0 == PAR_OFF, "prologue staging area");
        constexpr int P_G1 = PAR_OFF, P_E1 = PAR_OFF + 4 * C, P_B2 = PAR_OFF + 8 * C, P_G2 = PAR_OFF + 12 * C, P_E2 = PAR_OFF + 16 * C;
        constexpr int P_SB1 = PAR_OFF + 20 * C, P_SB2 = PAR_OFF + 22 * C, P_BN = PAR_OFF + 26 * C;
#ifdef SC_EXP
        stampv.start();
#endif
        f32x4 pv[NPV];
#pragma unroll
        for (int k = 0; k < NPV; k++)
            if (tid + 256 * k < PAR_V4) pv[k] = *reinterpret_cast<const f32x4*>(wf + C + (size_t)(tid + 256 * k) * 4);
        __builtin_amdgcn_sched_barrier(0);
        // conv1 -> LN -> ReLU
        acc_init<CT>(acc, Bn);
#pragma unroll
        for (int ct = 0; ct < CT; ct++) sa[ct] = san[ct];
        scale_load<P, CT>(san, net, 2 + 2 * b, wave, lane);
        conv_mma32<P, C, 9, CT, TILES, PSB, RS, TPI, true, AB>(0, wb, wave, lane, px, acc, ring, 9 * C * C * EB, sa);
        SC_MARK(0);
#pragma unroll
        for (int k = 0; k < NPV; k++)
            if (tid + 256 * k < PAR_V4) *reinterpret_cast<f32x4*>(g_smem + PAR_OFF + (tid + 256 * k) * 16) = pv[k];
        {
            LnStat L;
            ln_reduce<CT>(acc, L, C, wave, lane, s_stat, ln_parity, bad SC_STAMP_PASS(1));   // its barrier publishes the parameters
            ChP<CT> G, E;
            ch_load_lds<CT>(G, P_G1, wave, h);
            ch_load_lds<CT>(E, P_E1, wave, h);
            SC_MARK(3);
            ln_apply_relu_store<P, CT>(acc, L, G, E, pixbase, wave, h);
        }
        ch_load_lds<CT>(Bn, P_B2, wave, h);
        __syncthreads();
        SC_MARK(4);
        // conv2 -> LN
        acc_init<CT>(acc, Bn);
        {
            // the loop's last prefetches fetch the first k-steps of the NEXT block's conv1 (or wrap on the last block)
            const int nxt = (b + 1 < net.n_blocks) ? (int)(net.blk_stride_b * 2 - (size_t)9 * C * C * EB) : 0;
#pragma unroll
            for (int ct = 0; ct < CT; ct++) sa[ct] = san[ct];
            scale_load<P, CT>(san, net, b + 1 < net.n_blocks ? 3 + 2 * b : head_conv0, wave, lane);   // (heads reload theirs: two tiles per wave)
            conv_mma32<P, C, 9, CT, TILES, PSB, RS, TPI, true, AB>(0, wb + (size_t)9 * C * C * EB / 2, wave, lane, px, acc, ring, nxt, sa);
        }
        SC_MARK(5);
        // squeeze-excitation weights are requested now: their L2 round trip hides under the LayerNorm (issuing them
        // before conv2 instead measured the same: the conv loop is bound by the same vector-memory path).  The
        // columns of both layers are split over the 4 waves.
        VecW<C, NTW1> w1;
        VecW<C / 2, NTW> w2;
        vec_w_load<C, NTW1, NT1>(w1, wb + (size_t)9 * C * C * EB, wave * NTW1, lane);
        vec_w_load<C / 2, NTW, C / 16>(w2, wb + (size_t)9 * C * C * EB + (size_t)C * (C / 2), wave * NTW, lane);
        __builtin_amdgcn_sched_barrier(0);
        {
            LnStat L;
            ln_reduce<CT>(acc, L, C, wave, lane, s_stat, ln_parity, bad SC_STAMP_PASS(6));
            ChP<CT> G, E;
            ch_load_lds<CT>(G, P_G2, wave, h);
            ch_load_lds<CT>(E, P_E2, wave, h);
            ln_apply<CT>(acc, L, G, E, false);
        }
        SC_MARK(8);
        // global average pool over the 64 pixels: in-lane over the two tiles, then a halving butterfly over the
        // 32 lanes of a half-wave (lane j of a half ends up with register index j, or j>>1 when there are 16)
        {
            constexpr int NV = 16 * CT;
            float v[NV];
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 t = pair(acc[ct][0], r) + pair(acc[ct][1], r);
                    v[ct * 16 + r] = t.x;
                    v[ct * 16 + r + 1] = t.y;
                }
            pool_level<NV, NV / 2, 0>(v, lane);
            const int idx = (NV == 32) ? i32 : (i32 >> 1);
            s_xb[chan32<CT>(wave, idx >> 4, (idx & 15) >> 2, h) + (idx & 3)] = f2bf(v[0] * (1.0f / 64.0f));  // conv inputs are bf16 (autocast)
        }
        // SE biases of this lane's columns (lanes 16..63 mirror lanes 0..15): read before the barrier, off the
        // fc1 -> fc2 latency chain
        float sb1[NTW1], sb2[NTW];
#pragma unroll
        for (int k = 0; k < NTW1; k++) sb1[k] = *reinterpret_cast<const float*>(g_smem + P_SB1 + (wave * (16 * NTW1) + (lane & 15) * NTW1 + k) * 4);
#pragma unroll
        for (int k = 0; k < NTW; k++) sb2[k] = *reinterpret_cast<const float*>(g_smem + P_SB2 + (chan0<NTW>(wave, lane) + k) * 4);
        SC_MARK(9);
        __syncthreads();
        SC_MARK(10);
        bf16_t* s_hid = s_xb + 256;                              // hidden vector
        {
            // fc1: C -> C/2, ReLU
            f32x4 hh[NTW1];
#pragma unroll
            for (int k = 0; k < NTW1; k++) hh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            vec_mma<C, NTW1>(s_xb, w1, lane, hh);
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < NTW1; k++) {
                    // packed column (tile, lane) -> hidden channel ("lane owns NTW1 adjacent channels" order)
                    const int j = wave * (16 * NTW1) + (lane & 15) * NTW1 + k;
                    float t = hh[k][0] + sb1[k];
                    s_hid[j] = f2bf(fmaxf(t, 0.f));
                }
            }
        }
        __syncthreads();
        {
            // fc2: C/2 -> C, sigmoid; every wave produces the scales of exactly its own channels and hands them to
            // its lanes through LDS (wave-private, no barrier)
            f32x4 sc[NTW];
#pragma unroll
            for (int k = 0; k < NTW; k++) sc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            vec_mma<C / 2, NTW>(s_hid, w2, lane, sc);
            if (lane < 16) {
                const int c0 = chan0<NTW>(wave, lane);
#pragma unroll
                for (int k = 0; k < NTW; k++)
                    // v_exp_f32 + v_rcp_f32 (1 ulp each): the correctly rounded reciprocal expands to a 20-instruction division
                    s_scl[c0 + k] = __builtin_amdgcn_rcpf(1.0f + __expf(-(sc[k][0] + sb2[k])));
            }
        }
        SC_MARK(11);
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int ch = chan32<CT>(wave, ct, g, h);
                const f32x4 sv = *reinterpret_cast<const f32x4*>(s_scl + ch);
#pragma unroll
                for (int pt = 0; pt < 2; pt++) {
                    const f32x4 rv = f32x4{res[ct][pt][4 * g], res[ct][pt][4 * g + 1], res[ct][pt][4 * g + 2], res[ct][pt][4 * g + 3]};
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        const f32x2 y = pair(acc[ct][pt], 4 * g + k) * pair(sv, k) + pair(rv, k);
                        acc[ct][pt][4 * g + k] = fmaxf(y.x, 0.f);
                        acc[ct][pt][4 * g + k + 1] = fmaxf(y.y, 0.f);
                    }
                }
            }
        store_res();
        store_image32<P, CT>(acc, pixbase, wave, h);  // conv2 finished reading Xa before the SE barriers
        ch_load_lds<CT>(Bn, P_BN, wave, h);        // next block's conv1 bias: read before the barrier that frees the staging area
        SC_MARK(12);
        __syncthreads();
        SC_MARK(13);
        dump(b + 1);
    }
    dump(1000);
#ifdef SC_EXP
    stampv.start();
#endif

    const int gpb[2] = {i32 * HPSB, (32 + i32) * HPSB};   // rows of the plain (non-haloed) policy image
    // The three heads' per-channel parameters (1920 contiguous floats) are fetched cooperatively under the value
    // conv and staged in LDS (the SE scratch is dead now), like the per-block parameters of the trunk.
    constexpr int HPAR = XA_BYTES + RS_BYTES + 4096;            // byte offset of the staging area (s_vec ...)
    constexpr int HP_V = HPAR, HP_P1 = HPAR + 3 * HEAD * 4, HP_P2 = HPAR + 6 * HEAD * 4;
    static_assert(6 * HEAD * 4 + 3 * POL_PAD * 4 <= 3072 + 1024 + 64 + 15 * C * 2, "head parameters must fit in the staging area");
    f32x4 hpv[2];
#pragma unroll
    for (int k = 0; k < 2; k++)
        if (tid + 256 * k < 480) hpv[k] = *reinterpret_cast<const f32x4*>(net.wf + net.f_vhead + (size_t)(tid + 256 * k) * 4);
    __builtin_amdgcn_sched_barrier(0);
    // The 1x1 head convs are short (8 or 16 k-steps): their rings hold the WHOLE weight stream of a wave (a 3-step
    // prefetch distance is 200-400 MFMA cycles, less than the L2 latency: every k-step stalled, 35 % matrix rate).
    frag hr[HRS][2];      // weight ring of the 256-wide head convs (value conv -> policy conv1 carry)
    frag hr2[H2RS][1];    // ... of the 73-wide one
    int sh[2], sh2[1];    // fp8 weight scales of the head convs
    scale_load<P, 2>(sh, net, head_conv0, wave, lane);
    // ---- value head conv (py/module.py:89-94): conv1x1 C->256, LN, ReLU -> bf16 features in HBM
    {
        f32x16 hv[2][2];
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int pt = 0; pt < 2; pt++)
#pragma unroll
                for (int r = 0; r < 16; r++) hv[ct][pt][r] = 0.f;
        conv_mma32<P, C, 1, 2, 8, PSB, HRS, 1, false, HAB>(0, net.wb + net.o_vconv, wave, lane, px, hv, hr,
                                                           (int)((net.wb + net.o_pconv1) - (net.wb + net.o_vconv)) * 2, sh);
        scale_load<P, 2>(sh, net, head_conv0 + 1, wave, lane);
        scale_load<P, 1>(sh2, net, head_conv0 + 2, wave, lane);
        SC_MARK(20);
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (tid + 256 * k < 480) *reinterpret_cast<f32x4*>(g_smem + HPAR + (tid + 256 * k) * 16) = hpv[k];
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        ChP<2> Bv, G, E;
        ch_load_lds<2>(Bv, HP_V, wave, h);
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int pt = 0; pt < 2; pt++)
#pragma unroll
                for (int r = 0; r < 16; r++) hv[ct][pt][r] += Bv.v[ct][r >> 2][r & 3];
        LnStat L;
        int badv = bad;   // the trunk's NaNs and this head's own: the policy head below does not inherit the latter
        ln_reduce<2>(hv, L, HEAD, wave, lane, s_stat, ln_parity, badv SC_STAMP_PASS(21));
        ch_load_lds<2>(G, HP_V + HEAD * 4, wave, h);
        ch_load_lds<2>(E, HP_V + 2 * HEAD * 4, wave, h);
        ln_apply<2>(hv, L, G, E, true);
        SC_MARK(23);
        // Feature order in HBM = accumulator order ([wave][ct][pt][lane][16 registers], weights.hpp packs the rows of
        // value_head.ffn.0 to match): a lane's 16 values are 32 contiguous bytes and a wave's two stores fill whole
        // cache lines.  Writing [pixel][channel] rows from this layout (8 bytes per lane, 512 B apart) cost 4.6 k cycles.
        // A NaN LayerNorm variance so far (trunk or value head, see LnStat) poisons the lane's features and through them
        // the value; one in the policy head (below) poisons the priors only -- separate heads, as in py/module.py:136-152.
        // With value_head.ffn.0 inside this launch (A.fc1_arrive) the rows are read by OTHER workgroups of the launch:
        // write-through (sc1) stores, drained before the arrival below.
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(A.hval + (size_t)pos * (64 * HEAD), 0, 0x7fffffff, 0x00020000);
        const bool wt = HAND && A.fc1_arrive;
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int off = (((wave * 2 + ct) * 2 + pt) * 64 + lane) * 32;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
                    u32x4 v = {pk_bf16(pair(hv[ct][pt], 8 * q)), pk_bf16(pair(hv[ct][pt], 8 * q + 2)), pk_bf16(pair(hv[ct][pt], 8 * q + 4)),
                               pk_bf16(pair(hv[ct][pt], 8 * q + 6))};
                    if (badv) v = u32x4{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};
                    if (wt) __builtin_amdgcn_raw_buffer_store_b128(v, hrs, off + 16 * q, 0, AUX_SC1);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, hrs, off + 16 * q, 0, 0);
                }
            }
    }
    SC_MARK(16);
    // ---- policy head (py/module.py:70-76): conv1x1 C->256, LN, conv1x1 256->73, LN (no ReLU between)
    {
        ChP<2> Bv, G, E;
        ch_load_lds<2>(Bv, HP_P1, wave, h);
        f32x16 hp[2][2];
        acc_init<2>(hp, Bv);
        conv_mma32<P, C, 1, 2, 8, PSB, HRS, 1, true, HAB>(0, net.wb + net.o_pconv1, wave, lane, px, hp, hr, 0, sh);
        SC_MARK(24);
        // value_head.ffn.0 inside this launch: this wave's feature stores (a conv ago) have left the CU ...
        if (HAND && A.fc1_arrive) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_fill<P, 1, 4, H2RS>(hr2, net.wb + net.o_pconv2, wave, lane);   // hidden under the LayerNorm
        __builtin_amdgcn_sched_barrier(0);
        LnStat L;
        ln_reduce<2>(hp, L, HEAD, wave, lane, s_stat, ln_parity, bad SC_STAMP_PASS(25));
        // ... and, behind the LayerNorm's barrier, so have the other waves': one lane signals for the whole workgroup
        // (MI355X_MICROARCH.md, hand-off forms, row 1)
        if (HAND && A.fc1_arrive && tid == 0)
            __hip_atomic_fetch_add(A.fc1_arrive + (pos >> 6) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ch_load_lds<2>(G, HP_P1 + HEAD * 4, wave, h);
        ch_load_lds<2>(E, HP_P1 + 2 * HEAD * 4, wave, h);
        ln_apply<2>(hp, L, G, E, false);
        const int xb[2] = {XA_BYTES + gpb[0], XA_BYTES + gpb[1]};
        store_image32<P, 2>(hp, xb, wave, h);   // Xh: the head image behind Xa
        SC_MARK(27);
    }
    __syncthreads();
    SC_MARK(17);
    {
        ChP<1> Bv, G, E;
        ch_load_lds<1>(Bv, HP_P2, wave, h);
        f32x16 z[1][2];
        acc_init<1>(z, Bv);
        const int pxh[2] = {gpb[0] + h * P::FB, gpb[1] + h * P::FB};
        conv_mma32<P, HEAD, 1, 1, 4, HPSB, H2RS, 1, true, H2AB>(XA_BYTES, net.wb + net.o_pconv2, wave, lane, pxh, z, hr2, 0, sh2);
        SC_MARK(28);
        // first reading of the block's arrival counter (judged at the barrier in front of the softmax, a round trip later)
        uint32_t first = 0;
        if (HAND && A.fc1_arrive && tid == 0 && !A.fc1_acquire)
            first = __hip_atomic_load(A.fc1_arrive + (pos >> 6) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // padded channels (>=73) have zero weights, bias, gamma, beta: they add 0 to both LN sums
        LnStat L;
        ln_reduce<1>(z, L, 73, wave, lane, s_stat, ln_parity, bad SC_STAMP_PASS(29));
        ch_load_lds<1>(G, HP_P2 + POL_PAD * 4, wave, h);
        ch_load_lds<1>(E, HP_P2 + 2 * POL_PAD * 4, wave, h);
        ln_apply<1>(z, L, G, E, false);
        SC_MARK(31);
        __syncthreads();  // everyone is done with Xa/Xh: the logits may overwrite the image area
        SC_MARK(32);
#pragma unroll
        for (int pt = 0; pt < 2; pt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ch = chan32<1>(wave, 0, r >> 2, h) + (r & 3);
                if (ch < 73) s_z[ch * 64 + bp[pt]] = z[0][pt][r];  // Flatten is channel-major (module.py:75)
            }
        if (bad) s_z[lane] = __builtin_nanf("");   // see LnStat: make the NaN visible in the priors
        if (HAND && A.fc1_arrive && tid == 0) reinterpret_cast<int*>(s_red)[8] = !A.fc1_acquire && (int32_t)(first - A.fc1_target) >= 0;
    }
    __syncthreads();
    SC_MARK(18);
    if constexpr (HAND) {
        if (A.fc1_arrive) {   // the tail's first round trips fly under the softmax
            fc1_wload(fh->w, net, pos & 63, 64, wave, lane);
            fh->have_a = reinterpret_cast<const int*>(s_red)[8];
            if (fh->have_a) fc1_load_a(fh->a, A.hval, pos >> 6, pos & 63, tid);   // the polling lane's reading matched in front of the barrier
            else if (tid == 0) fh->early = __hip_atomic_load(A.fc1_arrive + (pos >> 6) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- log_softmax over 4672 (module.py:80), then the legal-move gather of torch.rs:148-175
    // One pass over LDS with every read in flight at once (a rolled loop paid the LDS latency 2 x 19 times), and one
    // barrier: each wave reduces to (max, sum of exp relative to ITS max), the four pairs are combined by everyone.
    float zv[19];
#pragma unroll
    for (int j = 0; j < 19; j++) {
        const float t = s_z[tid + 256 * j];               // j = 18 reads past 4672 for tid >= 64: still inside the image area
        zv[j] = (tid + 256 * j < 4672) ? t : -3.0e38f;
    }
    float mw = zv[0];
#pragma unroll
    for (int j = 1; j < 19; j++) mw = fmaxf(mw, zv[j]);
    mw = wave_max64(mw);
    float sw = 0.f;
#pragma unroll
    for (int j = 0; j < 19; j++) sw += __expf(zv[j] - mw);
    sw = wave_sum64(sw);
    if (lane == 0) {
        s_red[wave] = mw;
        s_red[4 + wave] = sw;
    }
    __syncthreads();
    const float mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float se = (s_red[4] * __expf(s_red[0] - mx) + s_red[5] * __expf(s_red[1] - mx)) +
                     (s_red[6] * __expf(s_red[2] - mx) + s_red[7] * __expf(s_red[3] - mx));
    const float lse = mx + __logf(se);
    // (Not specialised for the fused step kernel, which never writes the log-probabilities: saying so at compile time was worth +0.9...
    // +1.45 % in the fp8 instantiations, but the softmax / gather code then compiled differently in k_step and in k_tower32 and the fp8
    // priors of `predict` and of the search differed by a few ulps -- tests/test_gpu_netloop.py caught it.  Code that produces
    // numbers must be the same in both kernels.)
    constexpr bool LEAN = false;
    if (!LEAN && A.logp) {
        float* lp = A.logp + (size_t)pos * 4672;
#pragma unroll
        for (int j = 0; j < 19; j++)
            if (tid + 256 * j < 4672) lp[tid + 256 * j] = zv[j] - lse;
    }
    SC_MARK(19);
    if (LEAN || A.prior) {
        const int n = A.n_legal[pos];
        const uint16_t* li = A.legal_idx + (size_t)pos * 224;
        float e = 0.f;
        if (tid < n) e = __expf(s_z[li[tid]] - lse);  // n <= 218 < 256 threads
        float s = e;
        s = wave_sum64(s);
        __syncthreads();
        if (lane == 0) s_red[wave] = s;
        __syncthreads();
        s = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]) + 1e-5f;  // post_process_distr (chess.rs:891)
        if (tid < n) A.prior[(size_t)pos * 224 + tid] = e / s;
    }
#ifdef SC_EXP
    SC_MARK(15);
    if (stampv.on)
        for (int k = 0; k < 40; k++) A.dbg[(size_t)pos * 64 * C + wave * 40 + k] = (float)stampv.t[k];
    if (clk_on) {
        A.dbg[(size_t)pos * 64 * C + 0] = (float)(clock64() - clk_t0);                               // s_memtime ticks
        A.dbg[(size_t)pos * 64 * C + 1] = (float)(__builtin_amdgcn_s_memrealtime() - clk_r0);       // 100 MHz ticks
        A.dbg[(size_t)pos * 64 * C + 2] = (float)(clk_r0 & 0xffffff);                               // start time (low bits): who ran when
    }
#endif
    return true;
}

#ifndef SC_T32_OCC
#define SC_T32_OCC 1
#endif
template <class P, int C, int RS, int TPI, int AB = SC_T32_AB>
__global__ __launch_bounds__(256, SC_T32_OCC) void k_tower32(TowerArgs A) {
    if ((int)blockIdx.x >= A.n_pos) return;
    NoHand nh;
    tower_body<P, C, RS, TPI, AB>(A, (int)blockIdx.x, nullptr, NoPre(), nh);
}

}  // namespace scnn
