/*
 * oracle/nn.c -- fp32 policy/value network forward of the CPU ORACLE (test infrastructure only).
 *
 * Restates reference py/module.py: ResBlockSE :14-46, PolicyHead :65-80, ValueHead :83-106,
 * ChessModule.forward :135-154.  torchvision SqueezeExcitation / timm LayerNorm2d are un-vendored
 * third-party modules (uv.lock:2411,2471); restated from their published definitions:
 *   SE(x)   = x * sigmoid(fc2(relu(fc1(avgpool(x)))))        (1x1 convs fc1: C->C/2, fc2: C/2->C)
 *   LN2d(x) = per-pixel LayerNorm over channels, eps=1e-6, affine.
 * Pinned by golden vectors generated from the reference module itself (tools/gen_golden_nn.py).
 *
 * `channels` is 256 in the reference (module.py:120-133); 128 is the build-defined variant of
 * BASELINE.json configs[1] (trunk C=128, heads 256 wide).
 *
 * emulate_bf16 = 1: round GEMM operands (weights + conv/linear inputs) to bfloat16 the way the HIP
 * engine does, keeping fp32 accumulate / LayerNorm / residual.  Used to separate quantisation
 * error from kernel bugs; the parity claim vs the reference is made against the fp32 mode.
 * emulate_bf16 = 2: the engine's fp8 mode (BASELINE configs[4]; no reference counterpart): the
 * convs (stem, residual blocks, the three head convs) take OCP e4m3 operands -- weights with one power-of-two scale per
 * output channel, inputs clamped to +-448 and rounded to nearest even -- everything else as in mode 1.
 */
#include "sc_oracle_nn.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float bf16_round(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN */
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&x, &u, 4);
    return x;
}

/* OCP e4m3: 4 exponent bits (bias 7), 3 mantissa bits, subnormals of 2^-9, maximum 448, no infinity.  The same rules
 * as smart-chess-rust_amd/csrc/weights.hpp and tools/scw.py. */
float orc_e4m3_round(float x) {
    if (x != x) return x;
    float a = fabsf(x);
    if (a > 448.f) a = 448.f;
    int ex;
    (void)frexpf(a, &ex);
    float q = a >= 0.015625f ? ldexpf(1.f, ex - 4) : 0.001953125f;
    float r = nearbyintf(a / q) * q;
    return x < 0 ? -r : r;
}
int orc_fp8_channel_exp(float maxabs) {
    if (!(maxabs > 0.f)) return 0;
    int ex;
    float f = frexpf(maxabs / 448.f, &ex);
    int e = f == 0.5f ? ex - 1 : ex;
    return e < -100 ? -100 : e > 100 ? 100 : e;
}
static int is_fp8_conv(int n_blocks, int t) {
    if (t == 0) return 1;
    t -= 4;
    if (t >= 0 && t < 12 * n_blocks) return t % 12 == 0 || t % 12 == 4;
    t -= 12 * n_blocks;
    return t == 0 || t == 8 || t == 12;
}

/* ---- deterministic weights shared with the engine and tools/scw.py ---- */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
float orc_prng_weight(uint64_t seed, int tensor, uint64_t idx, double scale, double shift) {
    uint64_t h = mix64(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)tensor * 0xD1B54A32D192ED03ULL + idx);
    double u = (double)(h >> 40);
    double x = (u + 0.5) / 8388608.0 - 1.0;
    return (float)(shift + x * scale);
}

int orc_net_num_tensors(int n_blocks) { return 4 + 12 * n_blocks + 8 + 8; }

/* tensor table in ChessModule.state_dict() order; returns element count, fills shape (<=4 dims) */
static int64_t tensor_info(int n_blocks, int C, int t, int shape[4], int* ndim, int* kind, int* fan_in) {
    /* kind: 0 weight (uniform +-1/sqrt(fan_in)), 1 bias (same bound), 2 LN weight (1+-0.25), 3 LN bias (+-0.25) */
    int H = 256;
#define SET4(a, b, c, d, k, f) do { shape[0] = a; shape[1] = b; shape[2] = c; shape[3] = d; *ndim = 4; *kind = k; *fan_in = f; return (int64_t)(a) * (b) * (c) * (d); } while (0)
#define SET2(a, b, k, f) do { shape[0] = a; shape[1] = b; *ndim = 2; *kind = k; *fan_in = f; return (int64_t)(a) * (b); } while (0)
#define SET1(a, k, f) do { shape[0] = a; *ndim = 1; *kind = k; *fan_in = f; return (int64_t)(a); } while (0)
    if (t == 0) SET4(C, 112, 3, 3, 0, 112 * 9);
    if (t == 1) SET1(C, 1, 112 * 9);
    if (t == 2) SET1(C, 2, 0);
    if (t == 3) SET1(C, 3, 0);
    t -= 4;
    if (t < 12 * n_blocks) {
        int j = t % 12;
        switch (j) {
            case 0: SET4(C, C, 3, 3, 0, C * 9);
            case 1: SET1(C, 1, C * 9);
            case 2: SET1(C, 2, 0);
            case 3: SET1(C, 3, 0);
            case 4: SET4(C, C, 3, 3, 0, C * 9);
            case 5: SET1(C, 1, C * 9);
            case 6: SET1(C, 2, 0);
            case 7: SET1(C, 3, 0);
            case 8: SET4(C / 2, C, 1, 1, 0, C);
            case 9: SET1(C / 2, 1, C);
            case 10: SET4(C, C / 2, 1, 1, 0, C / 2);
            case 11: SET1(C, 1, C / 2);
        }
    }
    t -= 12 * n_blocks;
    switch (t) {
        case 0: SET4(H, C, 1, 1, 0, C);          /* value_head.conv.0.weight */
        case 1: SET1(H, 1, C);
        case 2: SET1(H, 2, 0);
        case 3: SET1(H, 3, 0);
        case 4: SET2(128, 64 * H + 7, 0, 64 * H + 7); /* value_head.ffn.0.weight */
        case 5: SET1(128, 1, 64 * H + 7);
        case 6: SET2(1, 128, 0, 128);
        case 7: SET1(1, 1, 128);
        case 8: SET4(H, C, 1, 1, 0, C);          /* policy_head.model.0.weight */
        case 9: SET1(H, 1, C);
        case 10: SET1(H, 2, 0);
        case 11: SET1(H, 3, 0);
        case 12: SET4(73, H, 1, 1, 0, H);
        case 13: SET1(73, 1, H);
        case 14: SET1(73, 2, 0);
        case 15: SET1(73, 3, 0);
    }
    return -1;
}
int64_t orc_net_tensor_shape(int n_blocks, int C, int t, int shape[4], int* ndim) {
    int kind, fan;
    return tensor_info(n_blocks, C, t, shape, ndim, &kind, &fan);
}

struct orc_net {
    int n_blocks, C, emulate_bf16;
    int nt;
    float** w; /* tensors in state_dict order and PyTorch layout */
    int64_t* numel;
    /* derived layouts */
    float** wt; /* for conv weights: [tap][cin][cout]; for linear: [in][out] */
};

static void derive(struct orc_net* n, int t) {
    int shape[4], nd, kind, fan;
    int64_t ne = tensor_info(n->n_blocks, n->C, t, shape, &nd, &kind, &fan);
    free(n->wt[t]);
    n->wt[t] = NULL;
    if (kind != 0) return;
    float* d = (float*)malloc(sizeof(float) * (size_t)ne);
    const float* s = n->w[t];
    if (nd == 4) {
        int O = shape[0], I = shape[1], K = shape[2] * shape[3];
        int fp8 = n->emulate_bf16 == 2 && is_fp8_conv(n->n_blocks, t);
        for (int o = 0; o < O; o++) {
            int e = 0;
            if (fp8) {
                float m = 0;
                for (int64_t q = 0; q < (int64_t)I * K; q++) m = fmaxf(m, fabsf(s[(int64_t)o * I * K + q]));
                e = orc_fp8_channel_exp(m);
            }
            for (int i = 0; i < I; i++)
                for (int k = 0; k < K; k++) {
                    float v = s[((int64_t)o * I + i) * K + k];
                    d[((int64_t)k * I + i) * O + o] = fp8 ? ldexpf(orc_e4m3_round(ldexpf(v, -e)), e) : n->emulate_bf16 ? bf16_round(v) : v;
                }
        }
    } else {
        int O = shape[0], I = shape[1];
        for (int o = 0; o < O; o++)
            for (int i = 0; i < I; i++) {
                float v = s[(int64_t)o * I + i];
                d[(int64_t)i * O + o] = n->emulate_bf16 ? bf16_round(v) : v;
            }
    }
    n->wt[t] = d;
}

orc_net* orc_net_create(int n_blocks, int channels, uint64_t seed, int emulate_bf16) {
    struct orc_net* n = (struct orc_net*)calloc(1, sizeof *n);
    n->n_blocks = n_blocks;
    n->C = channels;
    n->emulate_bf16 = emulate_bf16;
    n->nt = orc_net_num_tensors(n_blocks);
    n->w = (float**)calloc((size_t)n->nt, sizeof(float*));
    n->wt = (float**)calloc((size_t)n->nt, sizeof(float*));
    n->numel = (int64_t*)calloc((size_t)n->nt, sizeof(int64_t));
    for (int t = 0; t < n->nt; t++) {
        int shape[4], nd, kind, fan;
        int64_t ne = tensor_info(n_blocks, channels, t, shape, &nd, &kind, &fan);
        n->numel[t] = ne;
        n->w[t] = (float*)malloc(sizeof(float) * (size_t)ne);
        double scale = kind <= 1 ? 1.0 / sqrt((double)fan) : 0.25;
        double shift = kind == 2 ? 1.0 : 0.0;
        for (int64_t i = 0; i < ne; i++) n->w[t][i] = orc_prng_weight(seed, t, (uint64_t)i, scale, shift);
        derive(n, t);
    }
    return n;
}
void orc_net_free(orc_net* n) {
    if (!n) return;
    for (int t = 0; t < n->nt; t++) {
        free(n->w[t]);
        free(n->wt[t]);
    }
    free(n->w);
    free(n->wt);
    free(n->numel);
    free(n);
}
int orc_net_set_tensor(orc_net* n, int t, const float* data, int64_t numel) {
    if (t < 0 || t >= n->nt || numel != n->numel[t]) return -1;
    memcpy(n->w[t], data, sizeof(float) * (size_t)numel);
    derive(n, t);
    return 0;
}
int orc_net_get_tensor(const orc_net* n, int t, float* out, int64_t numel) {
    if (t < 0 || t >= n->nt || numel != n->numel[t]) return -1;
    memcpy(out, n->w[t], sizeof(float) * (size_t)numel);
    return 0;
}

/* ---- layers; activations are [64 pixels][C] fp32 (pixel = rank*8+file) ---- */
/* input of a conv that runs in e4m3 in mode 2 */
static void round_conv_in(const struct orc_net* n, const float* in, float* out, int64_t cnt) {
    if (n->emulate_bf16 == 2)
        for (int64_t i = 0; i < cnt; i++) out[i] = orc_e4m3_round(in[i]);
    else if (n->emulate_bf16)
        for (int64_t i = 0; i < cnt; i++) out[i] = bf16_round(in[i]);
    else if (in != out)
        memcpy(out, in, sizeof(float) * (size_t)cnt);
}
static void maybe_round(const struct orc_net* n, const float* in, float* out, int64_t cnt) {
    if (n->emulate_bf16)
        for (int64_t i = 0; i < cnt; i++) out[i] = bf16_round(in[i]);
    else if (in != out)
        memcpy(out, in, sizeof(float) * (size_t)cnt);
}

/* w: [tap][Cin][Cout] */
static void conv(const float* in, int Cin, const float* w, const float* bias, int Cout, int ksz, float* out) {
    for (int r = 0; r < 8; r++) {
        float* acc = out + (size_t)r * 8 * Cout;
        for (int f = 0; f < 8; f++)
            for (int co = 0; co < Cout; co++) acc[f * Cout + co] = bias[co];
        for (int tap = 0; tap < ksz * ksz; tap++) {
            int dy = ksz == 3 ? tap / 3 - 1 : 0, dx = ksz == 3 ? tap % 3 - 1 : 0;
            int rr = r + dy;
            if (rr < 0 || rr > 7) continue;
            const float* wt = w + (size_t)tap * Cin * Cout;
            for (int ci = 0; ci < Cin; ci++) {
                const float* wrow = wt + (size_t)ci * Cout;
                for (int f = 0; f < 8; f++) {
                    int ff = f + dx;
                    if (ff < 0 || ff > 7) continue;
                    float a = in[((size_t)rr * 8 + ff) * Cin + ci];
                    if (a == 0.0f) continue;
                    float* o = acc + (size_t)f * Cout;
                    for (int co = 0; co < Cout; co++) o[co] += a * wrow[co];
                }
            }
        }
    }
}
static void layernorm(float* x, int C, const float* g, const float* b, int relu) {
    for (int p = 0; p < 64; p++) {
        float* v = x + (size_t)p * C;
        double m = 0;
        for (int c = 0; c < C; c++) m += v[c];
        m /= C;
        double var = 0;
        for (int c = 0; c < C; c++) var += (v[c] - m) * (v[c] - m);
        var /= C;
        float rstd = (float)(1.0 / sqrt(var + 1e-6));
        for (int c = 0; c < C; c++) {
            float y = (float)(v[c] - m) * rstd * g[c] + b[c];
            v[c] = relu && y < 0 ? 0 : y;
        }
    }
}

void orc_net_forward(const orc_net* n, const int8_t* boards, const int32_t* meta, float* logp, float* value,
                     float* dbg_latent) {
    int C = n->C, H = 256;
    float* x = (float*)malloc(sizeof(float) * 64 * 256);
    float* a = (float*)malloc(sizeof(float) * 64 * 256);
    float* t1 = (float*)malloc(sizeof(float) * 64 * 256);
    float* t2 = (float*)malloc(sizeof(float) * 64 * 256);
    float inp[64 * 112];
    for (int i = 0; i < 64 * 112; i++) inp[i] = (float)boards[i];
    /* conv_block */
    conv(inp, 112, n->wt[0], n->w[1], C, 3, x);
    layernorm(x, C, n->w[2], n->w[3], 1);
    for (int b = 0; b < n->n_blocks; b++) {
        int t = 4 + 12 * b;
        round_conv_in(n, x, a, 64 * C);
        conv(a, C, n->wt[t + 0], n->w[t + 1], C, 3, t1);
        layernorm(t1, C, n->w[t + 2], n->w[t + 3], 1);
        round_conv_in(n, t1, a, 64 * C);
        conv(a, C, n->wt[t + 4], n->w[t + 5], C, 3, t2);
        layernorm(t2, C, n->w[t + 6], n->w[t + 7], 0);
        /* squeeze-excitation */
        float pool[256], h1[128], sc[256];
        for (int c = 0; c < C; c++) {
            float s = 0;
            for (int p = 0; p < 64; p++) s += t2[p * C + c];
            pool[c] = s / 64.0f;
            if (n->emulate_bf16) pool[c] = bf16_round(pool[c]);
        }
        for (int j = 0; j < C / 2; j++) {
            float s = n->w[t + 9][j];
            for (int c = 0; c < C; c++) s += pool[c] * n->wt[t + 8][(size_t)c * (C / 2) + j];
            h1[j] = s > 0 ? s : 0;
            if (n->emulate_bf16) h1[j] = bf16_round(h1[j]);
        }
        for (int c = 0; c < C; c++) {
            float s = n->w[t + 11][c];
            for (int j = 0; j < C / 2; j++) s += h1[j] * n->wt[t + 10][(size_t)j * C + c];
            sc[c] = 1.0f / (1.0f + expf(-s));
        }
        for (int p = 0; p < 64; p++)
            for (int c = 0; c < C; c++) {
                float y = t2[p * C + c] * sc[c] + x[p * C + c];
                x[p * C + c] = y > 0 ? y : 0;
            }
    }
    if (dbg_latent) memcpy(dbg_latent, x, sizeof(float) * 64 * (size_t)C);
    int vt = 4 + 12 * n->n_blocks, pt = vt + 8;
    round_conv_in(n, x, a, 64 * C);
    /* policy head (module.py:70-80) */
    conv(a, C, n->wt[pt + 0], n->w[pt + 1], H, 1, t1);
    layernorm(t1, H, n->w[pt + 2], n->w[pt + 3], 0);
    round_conv_in(n, t1, t1, 64 * H);
    conv(t1, H, n->wt[pt + 4], n->w[pt + 5], 73, 1, t2);
    layernorm(t2, 73, n->w[pt + 6], n->w[pt + 7], 0);
    {
        /* Flatten is channel-major: flat = c*64 + pixel */
        double mx = -1e30;
        for (int i = 0; i < 64 * 73; i++)
            if (t2[i] > mx) mx = t2[i];
        double se = 0;
        for (int i = 0; i < 64 * 73; i++) se += exp((double)t2[i] - mx);
        double lse = mx + log(se);
        for (int p = 0; p < 64; p++)
            for (int c = 0; c < 73; c++) logp[c * 64 + p] = (float)((double)t2[p * 73 + c] - lse);
    }
    /* value head (module.py:89-106) */
    conv(a, C, n->wt[vt + 0], n->w[vt + 1], H, 1, t1);
    layernorm(t1, H, n->w[vt + 2], n->w[vt + 3], 1);
    maybe_round(n, t1, t1, 64 * H);
    {
        float h[128];
        const float* W = n->wt[vt + 4]; /* [16391][128] */
        for (int j = 0; j < 128; j++) h[j] = n->w[vt + 5][j];
        for (int c = 0; c < H; c++)
            for (int p = 0; p < 64; p++) {
                float v = t1[p * H + c];
                if (v == 0.0f) continue;
                const float* wr = W + (size_t)(c * 64 + p) * 128;
                for (int j = 0; j < 128; j++) h[j] += v * wr[j];
            }
        for (int m = 0; m < 7; m++) {
            float v = (float)meta[m];
            if (n->emulate_bf16) v = bf16_round(v);
            const float* wr = W + (size_t)(64 * H + m) * 128;
            for (int j = 0; j < 128; j++) h[j] += v * wr[j];
        }
        float s = n->w[vt + 7][0];
        for (int j = 0; j < 128; j++) {
            float r = h[j] > 0 ? h[j] : 0; /* engine keeps the FC1 output in fp32 */
            s += r * n->wt[vt + 6][j];
        }
        float v = tanhf(s);
        *value = v * (float)(meta[0] * 2 - 1);
    }
    free(x);
    free(a);
    free(t1);
    free(t2);
}
