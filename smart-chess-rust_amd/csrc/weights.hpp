// weights.hpp -- host side of the network parameters: deterministic init / SCW1 blob reading, and
// repacking of the reference state_dict layout (py/module.py; names in SURVEY.md section 8 a19)
// into the device layouts of nn_kernels.hpp:
//   * GEMM operands -> bf16, MFMA 16x16x32 B-fragment order [kstep][col tile][lane][8], with the
//     column permutation "lane owns NTW adjacent channels" (nn_kernels.hpp chan0());
//   * per-channel parameters -> fp32 in logical channel order.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "nn_kernels_layout.hpp"

namespace scw {

inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline float prng_weight(uint64_t seed, int tensor, uint64_t idx, double scale, double shift) {
    uint64_t h = mix64(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)tensor * 0xD1B54A32D192ED03ULL + idx);
    double u = (double)(h >> 40);
    double x = (u + 0.5) / 8388608.0 - 1.0;
    return (float)(shift + x * scale);
}
inline uint16_t f2bf(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ---- OCP e4m3 (fp8) with per-output-channel power-of-two scales: BASELINE configs[4] -------------------------------
// The same three rules in tools/scw.py (export) and oracle/nn.c (checker):
//   value -> e4m3: round to nearest even, subnormals kept (quantum 2^-9), magnitudes above 448 clamp to 448;
//   channel exponent e = the smallest integer with max|w| / 2^e <= 448 (0 for an all-zero channel), clipped to +-100;
//   stored byte q = e4m3(w / 2^e); the MFMA's E8M0 block scale of the row is 127 + e.
inline float e4m3_round(float x) {
    if (x != x) return x;
    float a = fabsf(x);
    if (a > 448.f) a = 448.f;
    int ex;
    (void)frexpf(a, &ex);                       // a = f * 2^ex, f in [0.5, 1)
    const float q = a >= 0.015625f ? ldexpf(1.f, ex - 4) : 0.001953125f;   // 3 mantissa bits; 2^-9 below 2^-6
    const float r = nearbyintf(a / q) * q;      // default rounding mode: ties to even
    return x < 0 ? -r : r;
}
inline uint8_t e4m3_encode(float x) {           // x must already be representable (e4m3_round)
    const uint8_t sgn = (x < 0 || (x == 0 && signbit(x))) ? 0x80 : 0;
    float a = fabsf(x);
    if (a == 0.f) return sgn;
    int ex;
    const float f = frexpf(a, &ex);             // a = f * 2^ex
    if (a < 0.015625f) return (uint8_t)(sgn | (int)(a * 512.f));            // subnormal: m * 2^-9
    const int e = ex - 1 + 7, m = (int)((f * 2.f - 1.f) * 8.f);
    return (uint8_t)(sgn | (e << 3) | m);
}
inline float e4m3_decode(uint8_t v) {
    const int e = (v >> 3) & 15, m = v & 7;
    const float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return (v & 0x80) ? -f : f;
}
inline int channel_exp(float maxabs) {
    if (!(maxabs > 0.f)) return 0;
    int ex;
    const float f = frexpf(maxabs / 448.f, &ex);   // maxabs / 448 = f * 2^ex
    int e = f == 0.5f ? ex - 1 : ex;
    return e < -100 ? -100 : e > 100 ? 100 : e;
}

struct TensorInfo {
    int shape[4];
    int ndim;
    int kind;  // 0 weight, 1 bias, 2 LN weight, 3 LN bias
    int fan_in;
    size_t numel;
};

// ChessModule.state_dict() order (tools/scw.py tensor_table)
inline std::vector<TensorInfo> tensor_table(int n_blocks, int C) {
    const int H = 256;
    std::vector<TensorInfo> t;
    auto add4 = [&](int a, int b, int c, int d, int k, int f) { t.push_back({{a, b, c, d}, 4, k, f, (size_t)a * b * c * d}); };
    auto add2 = [&](int a, int b, int k, int f) { t.push_back({{a, b, 1, 1}, 2, k, f, (size_t)a * b}); };
    auto add1 = [&](int a, int k, int f) { t.push_back({{a, 1, 1, 1}, 1, k, f, (size_t)a}); };
    add4(C, 112, 3, 3, 0, 112 * 9); add1(C, 1, 112 * 9); add1(C, 2, 0); add1(C, 3, 0);
    for (int b = 0; b < n_blocks; b++) {
        add4(C, C, 3, 3, 0, C * 9); add1(C, 1, C * 9); add1(C, 2, 0); add1(C, 3, 0);
        add4(C, C, 3, 3, 0, C * 9); add1(C, 1, C * 9); add1(C, 2, 0); add1(C, 3, 0);
        add4(C / 2, C, 1, 1, 0, C); add1(C / 2, 1, C);
        add4(C, C / 2, 1, 1, 0, C / 2); add1(C, 1, C / 2);
    }
    add4(H, C, 1, 1, 0, C); add1(H, 1, C); add1(H, 2, 0); add1(H, 3, 0);
    add2(128, 64 * H + 7, 0, 64 * H + 7); add1(128, 1, 64 * H + 7);
    add2(1, 128, 0, 128); add1(1, 1, 128);
    add4(H, C, 1, 1, 0, C); add1(H, 1, C); add1(H, 2, 0); add1(H, 3, 0);
    add4(73, H, 1, 1, 0, H); add1(73, 1, H); add1(73, 2, 0); add1(73, 3, 0);
    return t;
}

struct HostWeights {
    int n_blocks, C;
    std::vector<std::vector<float>> t;  // state_dict order, PyTorch layout
    bool fp8 = false;                   // conv weights are to be packed as e4m3 (SCW2 blob, or sc_net_config.precision)
    std::vector<std::vector<int8_t>> exps;   // per tensor: channel exponents from an SCW2 blob (empty: derived from the data)
};
// the conv tensors that run in e4m3 (everything but the squeeze-excitation 1x1s and the Linear layers)
inline bool is_fp8_conv(int n_blocks, int t) {
    if (t == 0) return true;
    t -= 4;
    if (t >= 0 && t < 12 * n_blocks) return t % 12 == 0 || t % 12 == 4;
    t -= 12 * n_blocks;
    return t == 0 || t == 8 || t == 12;
}

inline HostWeights init_prng(int n_blocks, int C, uint64_t seed) {
    HostWeights w;
    w.n_blocks = n_blocks;
    w.C = C;
    auto tab = tensor_table(n_blocks, C);
    w.t.resize(tab.size());
    for (size_t i = 0; i < tab.size(); i++) {
        const TensorInfo& ti = tab[i];
        double scale = ti.kind <= 1 ? 1.0 / sqrt((double)ti.fan_in) : 0.25;
        double shift = ti.kind == 2 ? 1.0 : 0.0;
        w.t[i].resize(ti.numel);
        for (size_t k = 0; k < ti.numel; k++) w.t[i][k] = prng_weight(seed, (int)i, k, scale, shift);
    }
    return w;
}

// SCW1 blob (tools/scw.py write_scw). Returns empty string on success, else the error.
inline std::string load_scw(const char* path, HostWeights& w) {
    FILE* f = fopen(path, "rb");
    if (!f) return std::string("cannot open ") + path;
    char magic[4];
    uint32_t hdr[3];
    if (fread(magic, 1, 4, f) != 4 || (memcmp(magic, "SCW1", 4) && memcmp(magic, "SCW2", 4)) || fread(hdr, 4, 3, f) != 3) {
        fclose(f);
        return "not an SCW1 / SCW2 file";
    }
    const bool v2 = !memcmp(magic, "SCW2", 4);   // tools/scw.py: + u32 precision in the header, + u32 encoding per tensor
    if (v2) {
        uint32_t prec;
        if (fread(&prec, 4, 1, f) != 1 || prec > 1) {
            fclose(f);
            return "unsupported precision in SCW2 header";
        }
        w.fp8 = prec == 1;
    }
    w.n_blocks = (int)hdr[0];
    w.C = (int)hdr[1];
    if ((w.C != 128 && w.C != 256) || w.n_blocks < 0 || w.n_blocks > 80) {
        fclose(f);
        return "unsupported network shape in SCW1 header";
    }
    auto tab = tensor_table(w.n_blocks, w.C);
    if (hdr[2] != tab.size()) {
        fclose(f);
        return "tensor count mismatch";
    }
    w.t.resize(tab.size());
    w.exps.assign(tab.size(), {});
    for (size_t i = 0; i < tab.size(); i++) {
        uint32_t th[5], enc = 0;
        uint64_t numel;
        if (fread(th, 4, 5, f) != 5 || fread(&numel, 8, 1, f) != 1 || numel != tab[i].numel || (v2 && fread(&enc, 4, 1, f) != 1) || enc > 1) {
            fclose(f);
            return "tensor header mismatch at index " + std::to_string(i);
        }
        w.t[i].resize(numel);
        if (enc == 0) {
            if (fread(w.t[i].data(), 4, numel, f) != numel) {
                fclose(f);
                return "truncated SCW file";
            }
        } else {
            // e4m3 bytes + one exponent per output channel: kept exactly (value = e4m3 * 2^e is what pack() re-encodes)
            const size_t O = (size_t)tab[i].shape[0], per = numel / O;
            std::vector<uint8_t> q(numel);
            w.exps[i].resize(O);
            // (an e4m3 tensor belongs in a file whose header says fp8, and only the conv tensors are exported that way)
            if (!w.fp8 || !is_fp8_conv(w.n_blocks, (int)i) || fread(w.exps[i].data(), 1, O, f) != O || fread(q.data(), 1, numel, f) != numel) {
                fclose(f);
                return "bad fp8 tensor at index " + std::to_string(i);
            }
            // channel exponents are clipped to +-100 by the exporter (tools/scw.py channel_exps, channel_exp() here); pack8()
            // writes 127 + e as the E8M0 block scale, where 255 is the NaN code: a foreign or damaged blob must not load
            bool ok = true;
            for (size_t o = 0; o < O; o++) ok = ok && w.exps[i][o] >= -100 && w.exps[i][o] <= 100;
            for (size_t k = 0; k < numel; k++) ok = ok && (q[k] & 0x7f) != 0x7f;   // the e4m3 NaN code is never exported
            if (!ok) {
                fclose(f);
                return "bad fp8 tensor at index " + std::to_string(i) + " (channel exponent outside [-100, 100] or a NaN code)";
            }
            for (size_t k = 0; k < numel; k++) w.t[i][k] = ldexpf(e4m3_decode(q[k]), w.exps[i][k / per]);
        }
    }
    fclose(f);
    return "";
}

// B[k][n] accessor -> packed [K/32][NT_TOTAL][64][8] with the channel permutation of chan0()
template <class F>
inline void pack_B(uint16_t* out, int K, int NT_TOTAL, int NTW, F getB) {
    const int S = K / 32;
    for (int s = 0; s < S; s++)
        for (int nt = 0; nt < NT_TOTAL; nt++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    int w = nt / NTW, i = nt % NTW, c = l & 15;
                    int n = w * (16 * NTW) + c * NTW + i;
                    int k = s * 32 + 8 * (l >> 4) + j;
                    out[(((size_t)s * NT_TOTAL + nt) * 64 + l) * 8 + j] = f2bf(getB(k, n));
                }
}

// W[k][n] accessor -> packed [K/16][TILES][64][8]: v_mfma_f32_32x32x16_bf16 A-fragment order for the transposed
// (channels on rows) tower, natural channel order (nn_tower32.hpp)
// host copy of nn_tower32.hpp:gpix2board (GEMM pixel (tile pt, lane-in-tile i) -> board pixel); the value-parity
// tests fail if the two ever disagree
inline int gpix2board32(int pt, int i) {
    const bool inA = (i < 4) || (i >= 12 && i < 16) || (i >= 20 && i < 28);
    const int a = inA ? (i < 4 ? i : (i < 16 ? i - 8 : i - 12)) : (i < 12 ? i - 4 : (i < 20 ? i - 8 : i - 16));
    return ((2 * pt + (inA ? 0 : 1) + 4 * (a >> 3)) << 3) | (a & 7);
}
template <class F>
inline void pack_A32(uint16_t* out, int K, int TILES, F getW) {
    const int S = K / 16;
    for (int s = 0; s < S; s++)
        for (int t = 0; t < TILES; t++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    int n = t * 32 + (l & 31);
                    int k = s * 16 + 8 * (l >> 5) + j;
                    out[(((size_t)s * TILES + t) * 64 + l) * 8 + j] = f2bf(getW(k, n));
                }
}

// e4m3 A fragments of v_mfma_scale_f32_32x32x64_f8f6f4: lane l holds row l & 31, k = 32 (l >> 5) .. + 31 of a k-step of
// 64; stored [K/64][TILES][2 halves][64 lanes][16 bytes] so that each half is one contiguous 1 KiB wave-load.
// exps[n]: the channel's exponent; the scale table row gets the E8M0 byte 127 + e for all 64 lanes of the tile.
template <class F>
inline void pack_A8(uint8_t* out, int K, int TILES, F getW, const std::vector<int>& exps) {
    const int S = K / 64;
    for (int s = 0; s < S; s++)
        for (int t = 0; t < TILES; t++)
            for (int q = 0; q < 2; q++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 16; j++) {
                        const int n = t * 32 + (l & 31);
                        const int k = s * 64 + 32 * (l >> 5) + 16 * q + j;
                        out[((((size_t)s * TILES + t) * 2 + q) * 64 + l) * 16 + j] = e4m3_encode(e4m3_round(ldexpf(getW(k, n), -exps[(size_t)n])));
                    }
}

struct Packed {
    std::vector<uint16_t> wb;
    std::vector<float> wf;
    scnn::NetLayout lay;
};

inline Packed pack(const HostWeights& w, bool v32 = true) {
    const int C = w.C, nb = w.n_blocks, H = 256;
    const bool fp8 = w.fp8 && v32;
    const int EB = fp8 ? 1 : 2;   // bytes per conv weight; offsets stay in units of 2 bytes
    Packed p;
    scnn::NetLayout& L = p.lay;
    L.n_blocks = nb;
    L.C = C;
    L.tower32 = v32 ? 1 : 0;
    L.fp8 = fp8 ? 1 : 0;
    size_t ob = 0, of = 0;
    L.o_stem = ob; ob += (size_t)9 * 128 * C * EB / 2;
    L.o_blocks = ob; L.blk_stride_b = (size_t)18 * C * C * EB / 2 + (size_t)C * (C / 2) * 2; ob += L.blk_stride_b * nb;
    L.o_vconv = ob; ob += (size_t)C * H * EB / 2;
    L.o_pconv1 = ob; ob += (size_t)C * H * EB / 2;
    L.o_pconv2 = ob; ob += (size_t)H * 128 * EB / 2;
    L.o_fc1 = ob; ob += (size_t)64 * H * 128;
    ob += (size_t)4 * 16 * 64 * 8;  // slack: kernels prefetch up to 3 k-steps (x16 column tiles) past a tensor
    L.f_stem = of; of += 3 * C;
    L.f_blocks = of; L.blk_stride_f = (size_t)6 * C + C / 2 + C; of += L.blk_stride_f * nb;
    L.f_vhead = of; of += 3 * H;
    L.f_phead1 = of; of += 3 * H;
    L.f_phead2 = of; of += 3 * 128;
    L.f_fc1b = of; of += 128;
    L.f_fc1m = of; of += 7 * 128;
    L.f_fc2w = of; of += 128;
    L.f_fc2b = of; of += 4;
    L.f_scales = of;
    if (fp8) of += (size_t)(2 * nb + 4) * 8 * 64;
    p.wb.assign(ob, 0);
    p.wf.assign(of, 0.f);
    // fp8: one conv's fragments + its row of the scale table (conv index: 0 stem, 1 + 2b / 2 + 2b block b, then the heads)
    auto pack8 = [&](int tensor, int conv_idx, size_t off_units, int K, int n_out, int TILES, auto getW) {
        std::vector<int> ex((size_t)TILES * 32, 0);
        for (int n = 0; n < n_out; n++) {
            if (!w.exps.empty() && !w.exps[(size_t)tensor].empty()) {
                ex[(size_t)n] = w.exps[(size_t)tensor][(size_t)n];
            } else {
                float m = 0.f;
                for (int k = 0; k < K; k++) m = fmaxf(m, fabsf(getW(k, n)));
                ex[(size_t)n] = channel_exp(m);
            }
        }
        pack_A8(reinterpret_cast<uint8_t*>(p.wb.data() + off_units), K, TILES, getW, ex);
        for (int t = 0; t < TILES; t++)
            for (int l = 0; l < 64; l++) {
                const int32_t sc = 127 + ex[(size_t)t * 32 + (l & 31)];
                memcpy(&p.wf[L.f_scales + (size_t)((conv_idx * 8 + t) * 64 + l)], &sc, 4);
            }
    };
    const int NTW = C / 64, NT = C / 16;
    // stem: conv_block.0 [C][112][3][3], K padded to 128 per tap
    {
        const float* W = w.t[0].data();
        auto get = [&](int k, int n) {
            int tap = k / 128, ci = k % 128;
            return ci < 112 ? W[((size_t)n * 112 + ci) * 9 + tap] : 0.f;
        };
        if (fp8) pack8(0, 0, L.o_stem, 9 * 128, C, C / 32, get);
        else if (v32) pack_A32(p.wb.data() + L.o_stem, 9 * 128, C / 32, get);
        else pack_B(p.wb.data() + L.o_stem, 9 * 128, NT, NTW, get);
        for (int c = 0; c < C; c++) {
            p.wf[L.f_stem + c] = w.t[1][c];
            p.wf[L.f_stem + C + c] = w.t[2][c];
            p.wf[L.f_stem + 2 * C + c] = w.t[3][c];
        }
    }
    for (int b = 0; b < nb; b++) {
        int t0 = 4 + 12 * b;
        uint16_t* wb = p.wb.data() + L.o_blocks + (size_t)b * L.blk_stride_b;
        float* wf = p.wf.data() + L.f_blocks + (size_t)b * L.blk_stride_f;
        for (int cv = 0; cv < 2; cv++) {
            const float* W = w.t[t0 + 4 * cv].data();
            auto get = [&](int k, int n) {
                int tap = k / C, ci = k % C;
                return W[((size_t)n * C + ci) * 9 + tap];
            };
            if (fp8) pack8(t0 + 4 * cv, 1 + 2 * b + cv, L.o_blocks + (size_t)b * L.blk_stride_b + (size_t)cv * 9 * C * C / 2, 9 * C, C, C / 32, get);
            else if (v32) pack_A32(wb + (size_t)cv * 9 * C * C, 9 * C, C / 32, get);
            else pack_B(wb + (size_t)cv * 9 * C * C, 9 * C, NT, NTW, get);
            for (int c = 0; c < C; c++) {
                wf[(3 * cv + 0) * C + c] = w.t[t0 + 4 * cv + 1][c];
                wf[(3 * cv + 1) * C + c] = w.t[t0 + 4 * cv + 2][c];
                wf[(3 * cv + 2) * C + c] = w.t[t0 + 4 * cv + 3][c];
            }
        }
        {
            const float* W1 = w.t[t0 + 8].data();  // [C/2][C]
            int NT1 = C / 32, NTW1 = NT1 / 4;
            uint16_t* se = wb + (size_t)18 * C * C * EB / 2;
            pack_B(se, C, NT1, NTW1, [&](int k, int n) { return W1[(size_t)n * C + k]; });
            const float* W2 = w.t[t0 + 10].data();  // [C][C/2]
            pack_B(se + (size_t)C * (C / 2), C / 2, NT, NTW, [&](int k, int n) { return W2[(size_t)n * (C / 2) + k]; });
            for (int j = 0; j < C / 2; j++) wf[6 * C + j] = w.t[t0 + 9][j];
            for (int c = 0; c < C; c++) wf[6 * C + C / 2 + c] = w.t[t0 + 11][c];
        }
    }
    int vt = 4 + 12 * nb, pt = vt + 8;
    {
        const float* W = w.t[vt].data();  // [256][C]
        auto getv = [&](int k, int n) { return W[(size_t)n * C + k]; };
        if (fp8) pack8(vt, 1 + 2 * nb, L.o_vconv, C, H, 8, getv);
        else if (v32) pack_A32(p.wb.data() + L.o_vconv, C, 8, getv);
        else pack_B(p.wb.data() + L.o_vconv, C, 16, 4, getv);
        for (int c = 0; c < H; c++) {
            p.wf[L.f_vhead + c] = w.t[vt + 1][c];
            p.wf[L.f_vhead + H + c] = w.t[vt + 2][c];
            p.wf[L.f_vhead + 2 * H + c] = w.t[vt + 3][c];
        }
        const float* F1 = w.t[vt + 4].data();  // [128][16391], column = ch*64 + px
        pack_B(p.wb.data() + L.o_fc1, 64 * H, 8, 2, [&](int k, int n) {
            int px = k / H, ch = k % H;   // pixel-major kernel (nn_tower16.hpp): features as [pixel][channel]
            if (v32) {
                // k_tower32 writes them in accumulator order: [wave][ct][pt][lane][register] (nn_tower32.hpp, value head)
                const int r = k & 15, lane = (k >> 4) & 63, pt = (k >> 10) & 1, ct = (k >> 11) & 1, wave = (k >> 12) & 3;
                ch = wave * 64 + ct * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                px = gpix2board32(pt, lane & 31);
            }
            return F1[(size_t)n * (64 * H + 7) + (size_t)ch * 64 + px];
        });
        for (int j = 0; j < 128; j++) {
            p.wf[L.f_fc1b + j] = w.t[vt + 5][j];
            for (int m = 0; m < 7; m++) p.wf[L.f_fc1m + m * 128 + j] = bf2f(f2bf(F1[(size_t)j * (64 * H + 7) + 64 * H + m]));
            p.wf[L.f_fc2w + j] = bf2f(f2bf(w.t[vt + 6][j]));
        }
        p.wf[L.f_fc2b] = w.t[vt + 7][0];
    }
    {
        const float* W = w.t[pt].data();
        auto getp = [&](int k, int n) { return W[(size_t)n * C + k]; };
        if (fp8) pack8(pt, 2 + 2 * nb, L.o_pconv1, C, H, 8, getp);
        else if (v32) pack_A32(p.wb.data() + L.o_pconv1, C, 8, getp);
        else pack_B(p.wb.data() + L.o_pconv1, C, 16, 4, getp);
        for (int c = 0; c < H; c++) {
            p.wf[L.f_phead1 + c] = w.t[pt + 1][c];
            p.wf[L.f_phead1 + H + c] = w.t[pt + 2][c];
            p.wf[L.f_phead1 + 2 * H + c] = w.t[pt + 3][c];
        }
        const float* W2 = w.t[pt + 4].data();  // [73][256]
        auto getp2 = [&](int k, int n) { return n < 73 ? W2[(size_t)n * H + k] : 0.f; };
        if (fp8) pack8(pt + 4, 3 + 2 * nb, L.o_pconv2, H, 73, 4, getp2);
        else if (v32) pack_A32(p.wb.data() + L.o_pconv2, H, 4, getp2);
        else pack_B(p.wb.data() + L.o_pconv2, H, 8, 2, getp2);
        for (int c = 0; c < 73; c++) {
            p.wf[L.f_phead2 + c] = w.t[pt + 5][c];
            p.wf[L.f_phead2 + 128 + c] = w.t[pt + 6][c];
            p.wf[L.f_phead2 + 256 + c] = w.t[pt + 7][c];
        }
    }
    return p;
}

}  // namespace scw
