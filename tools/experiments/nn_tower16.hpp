// nn_tower16.hpp -- the pixel-major tower (pixels on the MFMA row axis, v_mfma_f32_16x16x32_bf16, N split over the
// waves, LayerNorm by a 16-lane halving butterfly).  EXPERIMENT BUILDS ONLY (-DSC_EXP, tools/build_exp.sh; selected
// with SC_TOWER_V=1): it was the production kernel of the wide trunk until the channel-major tower (nn_tower32.hpp)
// overtook it at both widths and is kept as the A/B baseline.  Same contract as k_tower32.
#pragma once
#include "../../smart-chess-rust_amd/csrc/nn_kernels.hpp"

namespace scnn {

// GEMM row (0..63) -> board pixel.  Rows 8..15 of every 16-row MFMA tile take the files of their rank rotated
// by 6: together with the 32-byte row padding of the LDS image this makes the 16 (row, k-quarter) addresses of
// every ds_read_b128 lane group fall on 16 distinct 16-byte bank slots (conflict-free A-operand reads; a plain
// row = pixel mapping is 2-way conflicted for every padding because of the halo gap between ranks).
__device__ inline int row2pix(int row) {
    int j = (row >> 3) & 1, f = row & 7;
    return (row & ~7) | ((f + 6 * j) & 7);
}

// xoff: byte offset of the image inside g_smem.
// RS: weight ring slots (prefetch distance RS-1 k-steps).  TPI: taps per loop iteration; one iteration covers
// SPG = TPI*KPT k-steps, fully unrolled, and RS divides SPG (RS <= KPT with TPI = 1, or RS = SPG for the narrow
// trunk whose 4-step taps would otherwise cap the prefetch distance at 3 steps = 384 MFMA cycles, less than the
// L2 latency under load).  t0: first tap group (groups are processed cyclically from t0; 0 = natural order).
// Ring carry: `bq` belongs to the caller.  With PRE the first RS-1 slots already hold this layer's first k-steps
// (the previous layer's loop fetched them: its prefetches past its own last group go to byte offset `next_first`
// relative to ITS weights, i.e. to the next layer), so a layer starts without an exposed L2 round trip.
// AB: A-fragment buffers; the LDS reads run AB-1 k-steps ahead (a k-step of the narrow trunk is only 128 MFMA cycles,
// less than the LDS latency with four waves reading: one step ahead leaves the matrix pipe waiting on every m-tile).
template <int CIN, int TAPS, int NTW, int NT_TOTAL, bool HALO, int CP, int RS, int TPI, bool PRE = false, int AB = 4>
__device__ __forceinline__ void conv_mma(int xoff, const bf16_t* __restrict__ Wp, int wave_u, int lane, int t0,
                                         f32x4 (&acc)[4][NTW], bf16x8 (&bq)[RS][NTW], int next_first) {
    constexpr int KPT = CIN / 32;          // k-steps per tap
    constexpr int SPG = KPT * TPI;         // k-steps per loop iteration (tap group)
    constexpr int NG = TAPS / TPI;         // tap groups
    constexpr int SBB = NT_TOTAL * 1024;   // bytes per k-step of packed weights
    constexpr int PD = RS - 1;             // prefetch distance
    constexpr int AD = AB - 1;             // A prefetch distance
    static_assert(TAPS % TPI == 0 && SPG % RS == 0 && PD < SPG, "bad ring / tap-group geometry");
    static_assert(SPG % AB == 0 && AD <= SPG, "bad A buffer geometry");
    const int row16 = lane & 15, kq = lane >> 4;
    // LDS byte offsets of this lane's 4 A rows at the centre tap; tap offsets are scalars, k offsets immediates
    int pa[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        int row = mt * 16 + row16;
        pa[mt] = xoff + ((HALO ? hidx(row2pix(row)) : row) * CP + 8 * kq) * 2;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(Wp) + (size_t)wave_u * NTW * 512, 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    auto toffb = [](int t) { return (TAPS == 9) ? ((((t * 11) >> 5) - 1) * 10 + (t - 3 * ((t * 11) >> 5)) - 1) * CP * 2 : 0; };
    // B: ring of RS register slots, loads run RS-1 k-steps ahead of the MFMAs that consume them (the slot being
    // refilled was consumed one step earlier).  A: double-buffered LDS fragments, one step ahead.  Prefetches
    // past the last group wrap to the first one (valid memory, values unused).
    bf16x8 aq[AB][4];
    int tg = t0;
    int wcur = tg * (SPG * SBB);
    int pc[TPI][4];
#pragma unroll
    for (int tl = 0; tl < TPI; tl++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) pc[tl][mt] = pa[mt] + toffb(tg * TPI + tl);
    if (!PRE) {
#pragma unroll
        for (int st = 0; st < PD; st++)
#pragma unroll
            for (int i = 0; i < NTW; i++) bq[st][i] = wload(rsrc, voff + i * 1024, wcur + st * SBB);
    }
#pragma unroll
    for (int v = 0; v < AD; v++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) aq[v][mt] = lds_frag(pc[v / KPT][mt] + (v % KPT) * 64);
#pragma unroll 1
    for (int j = 0; j < NG; j++) {
        int tn = tg + 1;
        if (tn == NG) tn = 0;
        const int wnext = (j == NG - 1) ? next_first : tn * (SPG * SBB);
        int pn[TPI][4];
#pragma unroll
        for (int tl = 0; tl < TPI; tl++)
#pragma unroll
            for (int mt = 0; mt < 4; mt++) pn[tl][mt] = pa[mt] + toffb(tn * TPI + tl);
#pragma unroll
        for (int u = 0; u < SPG; u++) {
            constexpr int dummy = 0;
            (void)dummy;
            const int slot = u % RS;
#ifndef SC_EXP_NOW
#pragma unroll
            for (int i = 0; i < NTW; i++)
                bq[(slot + PD) % RS][i] = (u + PD < SPG) ? wload(rsrc, voff + i * 1024, wcur + (u + PD) * SBB)
                                                         : wload(rsrc, voff + i * 1024, wnext + (u + PD - SPG) * SBB);
#endif
#ifndef SC_EXP_NOA
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                constexpr int dummy2 = 0;
                (void)dummy2;
                const int v = u + AD;
                aq[v % AB][mt] = (v < SPG) ? lds_frag(pc[v / KPT][mt] + (v % KPT) * 64)
                                           : lds_frag(pn[(v - SPG) / KPT][mt] + ((v - SPG) % KPT) * 64);
            }
#endif
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int i = 0; i < NTW; i++)
                    acc[mt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[u % AB][mt], bq[slot][i], acc[mt][i], 0, 0, 0);
            // Issue order inside the step: one m-tile of MFMAs, then one LDS read and one weight load, ... so the
            // matrix pipe never waits behind a burst of 8 memory instructions; the fence keeps every prefetch in
            // the step it was written in (otherwise the scheduler sinks loads to just before their use and the
            // whole L2 latency is exposed on every step).
#pragma unroll
            for (int g = 0; g < 4; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (g < NTW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        tg = tn;
        wcur = wnext;
#pragma unroll
        for (int tl = 0; tl < TPI; tl++)
#pragma unroll
            for (int mt = 0; mt < 4; mt++) pc[tl][mt] = pn[tl][mt];
    }
}

// bias add + LayerNorm over `count` channels (eps 1e-6, timm LayerNorm2d) + optional ReLU, in place.
// s_stat: LDS [64 rows][4 waves] float2 partial (sum, sumsq); s_mr: LDS [4 waves][64 rows] float2 (mean, rstd).
// Row sums over the wave's columns use a halving butterfly over the 16 lanes that share a row group: 15 shuffles
// per statistic instead of 64, and lane c of each group ends up owning row t = c of its group's 16 rows.
// The per-channel parameters are fetched by ln_load, which the caller issues a layer EARLY (before the conv that
// feeds this LayerNorm) so that no epilogue waits on an L2 round trip.
template <int NTW>
struct LnP {
    float b[NTW], g[NTW], e[NTW];
};
template <int NTW>
__device__ __forceinline__ void ln_load(LnP<NTW>& P, const float* __restrict__ bias, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, int wave, int lane) {
    const int c0 = chan0<NTW>(wave, lane);
#pragma unroll
    for (int i = 0; i < NTW; i++) {
        P.b[i] = bias[c0 + i];
        P.g[i] = gamma[c0 + i];
        P.e[i] = beta[c0 + i];
    }
}
// ReLU is one v_max_f32 (NaN -> 0, unlike torch.relu): NaNs are caught at the LayerNorm variances (`bad`) and poison the
// kernel's outputs at the end, as in nn_tower32.hpp.
template <int NTW>
__device__ inline void bias_layernorm(f32x4 (&acc)[4][NTW], const LnP<NTW>& P, int count, bool relu, int wave, int lane,
                                      float* s_stat2, float* s_mr, int& parity, int& bad) {
    // two alternating partial-sum buffers: the buffer written here was last read two LayerNorms ago, and every
    // wave has passed the barrier of the LayerNorm in between since then -> one barrier per LayerNorm suffices
    float* s_stat = s_stat2 + (parity & 1) * 512;
    parity ^= 1;
    float sm[16], sq[16];  // index t = mt*4 + r
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                float v = acc[mt][i][r] + P.b[i];
                acc[mt][i][r] = v;
                s += v;
                q += v * v;
            }
            sm[mt * 4 + r] = s;
            sq[mt * 4 + r] = q;
        }
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1) {
        const bool hi = (lane & w) != 0;
#pragma unroll
        for (int t = 0; t < w; t++) {
            float send_s = hi ? sm[t] : sm[t + w], keep_s = hi ? sm[t + w] : sm[t];
            float send_q = hi ? sq[t] : sq[t + w], keep_q = hi ? sq[t + w] : sq[t];
            sm[t] = keep_s + __shfl_xor(send_s, w, 64);
            sq[t] = keep_q + __shfl_xor(send_q, w, 64);
        }
    }
    const int t_own = lane & 15;
    const int row_own = (t_own >> 2) * 16 + (lane >> 4) * 4 + (t_own & 3);
    reinterpret_cast<float2*>(s_stat)[row_own * 4 + wave] = make_float2(sm[0], sq[0]);
    __syncthreads();
    {
        const float4* st = reinterpret_cast<const float4*>(s_stat) + row_own * 2;
        float4 a = st[0], b = st[1];
        float s = (a.x + a.z) + (b.x + b.z);
        float q = (a.y + a.w) + (b.y + b.w);
        const float inv = 1.0f / (float)count;
        float mean = s * inv;
        float var = q * inv - mean * mean;
        bad |= (var != var) ? 1 : 0;
        var = fmaxf(var, 0.f);
        float rstd = __frsqrt_rn(var + 1e-6f);
        // wave-private broadcast through LDS (DS ops of one wave execute in order; no barrier needed)
        reinterpret_cast<float2*>(s_mr)[wave * 64 + row_own] = make_float2(mean, rstd);
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int row = mt * 16 + (lane >> 4) * 4 + r;
            float2 mr = reinterpret_cast<const float2*>(s_mr)[wave * 64 + row];
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                float y = (acc[mt][i][r] - mr.x) * mr.y * P.g[i] + P.e[i];
                acc[mt][i][r] = relu ? fmaxf(y, 0.f) : y;
            }
        }
}

// store the accumulator tile as bf16 into an LDS image (pixel stride CP elements; plain row index when !HALO)
template <int NTW, bool HALO, int CP>
__device__ inline void store_image(const f32x4 (&acc)[4][NTW], bf16_t* X, int wave, int lane) {
    const int c0 = chan0<NTW>(wave, lane);
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int row = mt * 16 + (lane >> 4) * 4 + r;
            bf16_t* dst = X + (HALO ? hidx(row2pix(row)) : row) * CP + c0;
            if (NTW == 4) {
                uint2 v;
                v.x = (uint32_t)f2bf(acc[mt][0][r]) | ((uint32_t)f2bf(acc[mt][1][r]) << 16);
                v.y = (uint32_t)f2bf(acc[mt][2][r]) | ((uint32_t)f2bf(acc[mt][3][r]) << 16);
                *reinterpret_cast<uint2*>(dst) = v;
            } else {
                uint32_t v = (uint32_t)f2bf(acc[mt][0][r]) | ((uint32_t)f2bf(acc[mt][1][r]) << 16);
                *reinterpret_cast<uint32_t*>(dst) = v;
            }
        }
}


template <int C, int RS, int TPI>
__global__ __launch_bounds__(256, 1) void k_tower(TowerArgs A) {
    constexpr int NTW = C / 64;        // column tiles per wave in the trunk (4 or 2)
    constexpr int NT = C / 16;
    constexpr int CP = C + 16;         // image pixel stride (elements): +32 B, see row2pix()
    constexpr int HP = HEAD + 16;
    unsigned char* smem = g_smem;
    constexpr int RP = C + 4;          // residual row stride (floats): +16 B skews LDS banks
    constexpr int XA_BYTES = 100 * CP * 2;
    constexpr int RS_BYTES = (64 * RP * 4 > 64 * HP * 2) ? 64 * RP * 4 : 64 * HP * 2;
    bf16_t* Xa = reinterpret_cast<bf16_t*>(smem);                           // [100][CP] bf16 image (conv A operand)
    float* Rs = reinterpret_cast<float*>(smem + XA_BYTES);                  // [64][RP] fp32 residual stream (trunk)
    bf16_t* Xh = reinterpret_cast<bf16_t*>(smem + XA_BYTES);                // [64][HP] policy hidden; aliases Rs (heads only)
    float* s_stat = reinterpret_cast<float*>(smem + XA_BYTES + RS_BYTES);   // [64][4][2]
    float* s_vec = s_stat + 1024;                                           // pooled[256] | hidden: 4 waves x [128]
    float* s_red = s_vec + 768;                                             // [8]
    float* s_mr = s_red + 8;                                                // [4 waves][64] (mean, rstd)
    bf16_t* s_xb = reinterpret_cast<bf16_t*>(s_vec);                        // SE vectors as packed bf16
    float* s_z = reinterpret_cast<float*>(smem);                            // policy logits [4672], aliases Xa (after the trunk)
    static_assert(4672 * 4 <= XA_BYTES, "policy logits must fit in the image area");

    const int pos = blockIdx.x;
    if (pos >= A.n_pos) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar addressing of the weights
    const NetDev& net = A.net;

    // ---- zero the image (halo stays zero for the whole kernel), then load the 112 input planes
    {
        uint4* z = reinterpret_cast<uint4*>(Xa);
        for (int i = tid; i < 100 * CP * 2 / 16; i += 256) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    {
        const int p = tid >> 2, q = tid & 3;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(A.boards + (size_t)pos * 7168 + p * 112 + q * 28);
        bf16_t* dst = Xa + hidx(p) * CP + q * 28;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            uint32_t w = src[k];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int8_t v = (int8_t)((w >> (8 * b)) & 0xff);
                dst[k * 4 + b] = f2bf((float)v);
            }
        }
    }
    __syncthreads();

    // The fp32 residual stream is parked in LDS between blocks (160 KiB/CU and one workgroup per CU make
    // that free) so that the conv loops keep only accumulators + operand rings in registers.
    f32x4 acc[4][NTW];
    int ln_parity = 0;
    int bad = 0;   // a LayerNorm saw a NaN variance
    auto store_res = [&]() {
        const int c0 = chan0<NTW>(wave, lane);
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float* d = Rs + (mt * 16 + (lane >> 4) * 4 + r) * RP + c0;
                if (NTW == 4)
                    *reinterpret_cast<float4*>(d) = make_float4(acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]);
                else
                    *reinterpret_cast<float2*>(d) = make_float2(acc[mt][0][r], acc[mt][1][r]);
            }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) acc[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- conv_block (py/module.py:120-126): conv3x3 112->C (K padded to 128/tap), LN, ReLU
    zero_acc();
    // stem: K = 9 taps x 128 padded input planes (4 k-steps per tap)
    constexpr int SRS = (TPI > 1) ? 4 * TPI : 4;
    static_assert(SRS == RS, "the stem shares the trunk's weight ring (ring carry across layers)");
    constexpr int t0 = 0;   // taps in natural order (cyclic per-workgroup start offsets were tried: no gain, and they
                            // break the bitwise independence of a position's result from its slot)
    constexpr int GRP_BYTES = (C / 32) * TPI * NT * 1024;   // bytes of one tap group of a trunk conv
    bf16x8 ring[RS][NTW];                                  // weight prefetch ring, carried from layer to layer
    {
        LnP<NTW> P;
        const float* f = net.wf + net.f_stem;
        ln_load<NTW>(P, f, f + C, f + 2 * C, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        const bf16_t* w0 = net.wb + net.o_stem;
        const int to_blk0 = (int)((net.wb + net.o_blocks) - w0) * 2 + t0 * GRP_BYTES;
        conv_mma<128, 9, NTW, NT, true, CP, RS, TPI, false>(0, w0, wave, lane, t0, acc, ring, to_blk0);
        bias_layernorm<NTW>(acc, P, C, true, wave, lane, s_stat, s_mr, ln_parity, bad);
    }
    store_res();
    store_image<NTW, true, CP>(acc, Xa, wave, lane);  // every wave passed the LN barriers: the input image is dead
    __syncthreads();
    auto dump = [&](int stage) {
        if (A.dbg && A.dbg_stage == stage) {
            const int c0 = chan0<NTW>(wave, lane);
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int i = 0; i < NTW; i++)
                        A.dbg[((size_t)pos * 64 + row2pix(mt * 16 + (lane >> 4) * 4 + r)) * C + c0 + i] = acc[mt][i][r];
        }
    };
    dump(0);

    // developer aid: dbg_stage 2000 -> cycle stamps (summed over blocks) of the block's phases in dbg[pos][0..7]
    const bool stamp = A.dbg && A.dbg_stage == 2000 && tid == 0;
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = 0;
    auto mark = [&](int k) {
        if (stamp) {
            long long t = clock64();
            tacc[k] += t - tprev;
            tprev = t;
        }
    };
    // ---- residual tower (ResBlockSE.forward, py/module.py:38-46)
#pragma unroll 1
    for (int b = 0; b < net.n_blocks; b++) {
        const bf16_t* wb = net.wb + net.o_blocks + (size_t)b * net.blk_stride_b;
        const float* wf = net.wf + net.f_blocks + (size_t)b * net.blk_stride_f;
        // conv1 -> LN -> ReLU.  Per-channel parameters are requested BEFORE the conv that feeds them.
        if (stamp) tprev = clock64();
        constexpr int NT1 = C / 32;                       // column tiles of SE fc1
        constexpr bool FULL1 = (C == 128);
        constexpr int NTW1 = FULL1 ? NT1 : NT1 / 4;
        constexpr int PW1 = NT1 / 4;                      // fc1 tiles per wave in the packed column order
        LnP<NTW> P1, P2;
        float b1v[NTW1], b2v[NTW];
        ln_load<NTW>(P1, wf, wf + C, wf + 2 * C, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        zero_acc();
        conv_mma<C, 9, NTW, NT, true, CP, RS, TPI, true>(0, wb, wave, lane, t0, acc, ring, 9 * C * C * 2 + t0 * GRP_BYTES);
        mark(0);
        ln_load<NTW>(P2, wf + 3 * C, wf + 4 * C, wf + 5 * C, wave, lane);
        {
            const float* b1 = wf + 6 * C;
            const float* b2 = wf + 6 * C + C / 2;
            const int c0 = chan0<NTW>(wave, lane);
#pragma unroll
            for (int i = 0; i < NTW1; i++) {
                // packed column (tile, lane) -> hidden channel: tiles were laid out for a 4-wave split
                const int tile = FULL1 ? i : wave * NTW1 + i;
                b1v[i] = b1[(tile / PW1) * (16 * PW1) + (lane & 15) * PW1 + (tile % PW1)];
            }
#pragma unroll
            for (int i = 0; i < NTW; i++) b2v[i] = b2[c0 + i];
        }
        __builtin_amdgcn_sched_barrier(0);
        bias_layernorm<NTW>(acc, P1, C, true, wave, lane, s_stat, s_mr, ln_parity, bad);
        store_image<NTW, true, CP>(acc, Xa, wave, lane);
        __syncthreads();
        mark(1);
        // conv2 -> LN
        zero_acc();
        {
            // the loop's last prefetches fetch the first k-steps of the NEXT block's conv1 (or wrap on the last block)
            const int nxt = (b + 1 < net.n_blocks) ? (int)(net.blk_stride_b - (size_t)9 * C * C) * 2 + t0 * GRP_BYTES : t0 * GRP_BYTES;
            conv_mma<C, 9, NTW, NT, true, CP, RS, TPI, true>(0, wb + (size_t)9 * C * C, wave, lane, t0, acc, ring, nxt);
        }
        mark(2);
        // squeeze-excitation weights are requested now: their L2 round trip hides under the LayerNorm below.
        // Narrow trunk: every wave computes the whole C -> C/2 layer itself (16 fragments), which removes the
        // hidden-vector exchange and its barrier; wide trunk: the columns are split over the 4 waves.
        VecW<C, NTW1> w1;
        vec_w_load<C, NTW1, NT1>(w1, wb + (size_t)18 * C * C, FULL1 ? 0 : wave * NTW1, lane);
        VecW<C / 2, NTW> w2;
        vec_w_load<C / 2, NTW, NT>(w2, wb + (size_t)18 * C * C + (size_t)C * (C / 2), wave * NTW, lane);
        __builtin_amdgcn_sched_barrier(0);  // keep the loads HERE (the scheduler would sink them to their use)
        bias_layernorm<NTW>(acc, P2, C, false, wave, lane, s_stat, s_mr, ln_parity, bad);
        mark(3);
        // global average pool over the 64 pixels
        {
            float cs[NTW];
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                float s = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) s += acc[mt][i][r];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                cs[i] = s * (1.0f / 64.0f);
            }
            if (lane < 16) {
                const int c0 = chan0<NTW>(wave, lane);
#pragma unroll
                for (int i = 0; i < NTW; i++) s_xb[c0 + i] = f2bf(cs[i]);  // conv inputs are bf16 (autocast)
            }
        }
        __syncthreads();
        bf16_t* s_hid = s_xb + 256 + (FULL1 ? wave * 128 : 0);   // hidden vector (wave-private when FULL1)
        {
            // fc1: C -> C/2, ReLU
            f32x4 h[NTW1];
#pragma unroll
            for (int i = 0; i < NTW1; i++) h[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            vec_mma<C, NTW1>(s_xb, w1, lane, h);
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < NTW1; i++) {
                    const int tile = FULL1 ? i : wave * NTW1 + i;
                    const int j = (tile / PW1) * (16 * PW1) + (lane & 15) * PW1 + (tile % PW1);
                    float v = h[i][0] + b1v[i];
                    s_hid[j] = f2bf(fmaxf(v, 0.f));
                }
            }
        }
        if (!FULL1) __syncthreads();  // FULL1: the hidden vector is wave-private (DS ops of a wave execute in order)
        float scl[NTW];
        {
            // fc2: C/2 -> C, sigmoid; every wave produces the scales of exactly its own channels
            f32x4 sc[NTW];
#pragma unroll
            for (int i = 0; i < NTW; i++) sc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            vec_mma<C / 2, NTW>(s_hid, w2, lane, sc);
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                float v = 1.0f / (1.0f + __expf(-(sc[i][0] + b2v[i])));  // valid in lanes 0..15
                scl[i] = __shfl(v, lane & 15, 64);
            }
        }
        mark(4);
        {
            const int c0 = chan0<NTW>(wave, lane);
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float* rp = Rs + (mt * 16 + (lane >> 4) * 4 + r) * RP + c0;
                    float rv[NTW];
                    if (NTW == 4) {
                        float4 t = *reinterpret_cast<const float4*>(rp);
                        rv[0] = t.x; rv[1] = t.y; rv[NTW - 2] = t.z; rv[NTW - 1] = t.w;
                    } else {
                        float2 t = *reinterpret_cast<const float2*>(rp);
                        rv[0] = t.x; rv[1] = t.y;
                    }
#pragma unroll
                    for (int i = 0; i < NTW; i++) {
                        float y = acc[mt][i][r] * scl[i] + rv[i];
                        acc[mt][i][r] = fmaxf(y, 0.f);
                    }
                }
        }
        store_res();                                      // each lane rewrites exactly the cells it just read
        store_image<NTW, true, CP>(acc, Xa, wave, lane);  // conv2 finished reading Xa before the SE barriers
        __syncthreads();
        mark(5);
        dump(b + 1);
    }
    dump(1000);
    if (stamp) {
        tprev = clock64();
    }

    // ---- value head conv (py/module.py:89-94): conv1x1 C->256, LN, ReLU -> bf16 features in HBM
    {
        f32x4 hv[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int i = 0; i < 4; i++) hv[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* f = net.wf + net.f_vhead;
        LnP<4> P;
        ln_load<4>(P, f, f + HEAD, f + 2 * HEAD, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 hr[4][4];
        conv_mma<C, 1, 4, 16, true, CP, 4, 1>(0, net.wb + net.o_vconv, wave, lane, 0, hv, hr, 0);
        bias_layernorm<4>(hv, P, HEAD, true, wave, lane, s_stat, s_mr, ln_parity, bad);
        const int c0 = chan0<4>(wave, lane);
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int p = row2pix(mt * 16 + (lane >> 4) * 4 + r);
                uint2 v;
                v.x = (uint32_t)f2bf(hv[mt][0][r]) | ((uint32_t)f2bf(hv[mt][1][r]) << 16);
                v.y = (uint32_t)f2bf(hv[mt][2][r]) | ((uint32_t)f2bf(hv[mt][3][r]) << 16);
                *reinterpret_cast<uint2*>(A.hval + ((size_t)pos * 64 + p) * HEAD + c0) = v;
            }
    }
    // ---- policy head (py/module.py:70-76): conv1x1 C->256, LN, conv1x1 256->73, LN (no ReLU between)
    {
        f32x4 hp[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int i = 0; i < 4; i++) hp[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* f = net.wf + net.f_phead1;
        LnP<4> P;
        ln_load<4>(P, f, f + HEAD, f + 2 * HEAD, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 hr[4][4];
        conv_mma<C, 1, 4, 16, true, CP, 4, 1>(0, net.wb + net.o_pconv1, wave, lane, 0, hp, hr, 0);
        bias_layernorm<4>(hp, P, HEAD, false, wave, lane, s_stat, s_mr, ln_parity, bad);
        store_image<4, false, HP>(hp, Xh, wave, lane);
    }
    __syncthreads();
    {
        f32x4 z[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int i = 0; i < 2; i++) z[mt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* f = net.wf + net.f_phead2;
        LnP<2> P;
        ln_load<2>(P, f, f + POL_PAD, f + 2 * POL_PAD, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 hr[4][2];
        conv_mma<HEAD, 1, 2, 8, false, HP, 4, 1>(XA_BYTES, net.wb + net.o_pconv2, wave, lane, 0, z, hr, 0);
        // padded channels (>=73) have zero weights, bias, gamma, beta: they add 0 to both LN sums
        bias_layernorm<2>(z, P, 73, false, wave, lane, s_stat, s_mr, ln_parity, bad);
        __syncthreads();  // everyone is done with Xa/Xh: the logits may overwrite the image area
        const int c0 = chan0<2>(wave, lane);
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    int ch = c0 + i, p = row2pix(mt * 16 + (lane >> 4) * 4 + r);
                    if (ch < 73) s_z[ch * 64 + p] = z[mt][i][r];  // Flatten is channel-major (module.py:75)
                }
        if (bad) {   // make the NaN visible in the priors and (through the value features) in the value
            s_z[lane] = __builtin_nanf("");
            A.hval[(size_t)pos * 64 * HEAD + tid] = 0x7fc0;
        }
    }
    __syncthreads();
    // ---- log_softmax over 4672 (module.py:80), then the legal-move gather of torch.rs:148-175
    float mx = -3.0e38f;
    for (int i = tid; i < 4672; i += 256) mx = fmaxf(mx, s_z[i]);
    mx = wave_max64(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float se = 0.f;
    for (int i = tid; i < 4672; i += 256) se += __expf(s_z[i] - mx);
    se = wave_sum64(se);
    if (lane == 0) s_red[4 + wave] = se;
    __syncthreads();
    se = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    const float lse = mx + __logf(se);
    if (A.logp) {
        float* lp = A.logp + (size_t)pos * 4672;
        for (int i = tid; i < 4672; i += 256) lp[i] = s_z[i] - lse;
    }
    if (A.prior) {
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
    if (stamp) {
        mark(6);
        for (int k = 0; k < 8; k++) A.dbg[(size_t)pos * 64 * C + k] = (float)tacc[k];
    }
}

}  // namespace scnn
