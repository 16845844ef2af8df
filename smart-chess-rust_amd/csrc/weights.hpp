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
};

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
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "SCW1", 4) || fread(hdr, 4, 3, f) != 3) {
        fclose(f);
        return "not an SCW1 file";
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
    for (size_t i = 0; i < tab.size(); i++) {
        uint32_t th[5];
        uint64_t numel;
        if (fread(th, 4, 5, f) != 5 || fread(&numel, 8, 1, f) != 1 || numel != tab[i].numel) {
            fclose(f);
            return "tensor header mismatch at index " + std::to_string(i);
        }
        w.t[i].resize(numel);
        if (fread(w.t[i].data(), 4, numel, f) != numel) {
            fclose(f);
            return "truncated SCW1 file";
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

struct Packed {
    std::vector<uint16_t> wb;
    std::vector<float> wf;
    scnn::NetLayout lay;
};

inline Packed pack(const HostWeights& w, bool v32 = true) {
    const int C = w.C, nb = w.n_blocks, H = 256;
    Packed p;
    scnn::NetLayout& L = p.lay;
    L.n_blocks = nb;
    L.C = C;
    L.tower32 = v32 ? 1 : 0;
    size_t ob = 0, of = 0;
    L.o_stem = ob; ob += (size_t)9 * 128 * C;
    L.o_blocks = ob; L.blk_stride_b = (size_t)18 * C * C + (size_t)C * (C / 2) * 2; ob += L.blk_stride_b * nb;
    L.o_vconv = ob; ob += (size_t)C * H;
    L.o_pconv1 = ob; ob += (size_t)C * H;
    L.o_pconv2 = ob; ob += (size_t)H * 128;
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
    p.wb.assign(ob, 0);
    p.wf.assign(of, 0.f);
    const int NTW = C / 64, NT = C / 16;
    // stem: conv_block.0 [C][112][3][3], K padded to 128 per tap
    {
        const float* W = w.t[0].data();
        auto get = [&](int k, int n) {
            int tap = k / 128, ci = k % 128;
            return ci < 112 ? W[((size_t)n * 112 + ci) * 9 + tap] : 0.f;
        };
        if (v32) pack_A32(p.wb.data() + L.o_stem, 9 * 128, C / 32, get);
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
            if (v32) pack_A32(wb + (size_t)cv * 9 * C * C, 9 * C, C / 32, get);
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
            pack_B(wb + (size_t)18 * C * C, C, NT1, NTW1, [&](int k, int n) { return W1[(size_t)n * C + k]; });
            const float* W2 = w.t[t0 + 10].data();  // [C][C/2]
            pack_B(wb + (size_t)18 * C * C + (size_t)C * (C / 2), C / 2, NT, NTW, [&](int k, int n) { return W2[(size_t)n * (C / 2) + k]; });
            for (int j = 0; j < C / 2; j++) wf[6 * C + j] = w.t[t0 + 9][j];
            for (int c = 0; c < C; c++) wf[6 * C + C / 2 + c] = w.t[t0 + 11][c];
        }
    }
    int vt = 4 + 12 * nb, pt = vt + 8;
    {
        const float* W = w.t[vt].data();  // [256][C]
        auto getv = [&](int k, int n) { return W[(size_t)n * C + k]; };
        if (v32) pack_A32(p.wb.data() + L.o_vconv, C, 8, getv);
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
        if (v32) pack_A32(p.wb.data() + L.o_pconv1, C, 8, getp);
        else pack_B(p.wb.data() + L.o_pconv1, C, 16, 4, getp);
        for (int c = 0; c < H; c++) {
            p.wf[L.f_phead1 + c] = w.t[pt + 1][c];
            p.wf[L.f_phead1 + H + c] = w.t[pt + 2][c];
            p.wf[L.f_phead1 + 2 * H + c] = w.t[pt + 3][c];
        }
        const float* W2 = w.t[pt + 4].data();  // [73][256]
        auto getp2 = [&](int k, int n) { return n < 73 ? W2[(size_t)n * H + k] : 0.f; };
        if (v32) pack_A32(p.wb.data() + L.o_pconv2, H, 4, getp2);
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
