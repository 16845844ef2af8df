// engine.hip -- host side of libsc_engine.so: the C ABI of include/sc_engine.h over the HIP kernels.
//
// Replaces, behind the reference's own interfaces: backend construction + Game::predict
// (src/main.rs:83-128, src/backends/torch.rs:89-146), the rules/encoder calls into python-chess
// (src/chess.rs:665-877) and the per-game self-play loop (src/main.rs:155-238) -- the latter for
// many concurrent games per GPU.  All compute runs on the GPU; there is no CPU fallback: every
// entry point fails with an error code if HIP or the device is unavailable.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <initializer_list>
#include <vector>

#include "../../include/sc_engine.h"
#include "launchers.hpp"
#include "trace_json.hpp"
#include "weights.hpp"

// cleanup paths: free a list of device pointers, ignoring errors
static void dfree(std::initializer_list<void*> ptrs) {
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
}

// The C ABI's structs are bound field by field from Rust (integration/hip.rs), ctypes (scamd/binding.py) and C: their layout
// is part of the interface.  A new field goes at the END, together with the bindings (tests/test_abi.py compares all three).
#define SC_LAYOUT(T, size) static_assert(sizeof(T) == (size), #T ": size changed -- update integration/hip.rs and scamd/binding.py")
#define SC_FIELD(T, f, off) static_assert(offsetof(T, f) == (off), #T "." #f ": offset changed")
SC_LAYOUT(sc_net_config, 24);
SC_FIELD(sc_net_config, n_res_blocks, 0); SC_FIELD(sc_net_config, channels, 4); SC_FIELD(sc_net_config, seed, 8);
SC_FIELD(sc_net_config, precision, 16); SC_FIELD(sc_net_config, reserved, 20);
SC_LAYOUT(sc_selfplay_config, 88);
SC_FIELD(sc_selfplay_config, n_slots, 0); SC_FIELD(sc_selfplay_config, n_games, 4); SC_FIELD(sc_selfplay_config, rollout_num, 8);
SC_FIELD(sc_selfplay_config, num_steps, 12); SC_FIELD(sc_selfplay_config, cpuct, 16); SC_FIELD(sc_selfplay_config, temperature, 20);
SC_FIELD(sc_selfplay_config, temperature_switch, 24); SC_FIELD(sc_selfplay_config, epsilon, 28); SC_FIELD(sc_selfplay_config, with_noise, 32);
SC_FIELD(sc_selfplay_config, outcome_gate, 36); SC_FIELD(sc_selfplay_config, evaluator, 40); SC_FIELD(sc_selfplay_config, external_noise, 44);
SC_FIELD(sc_selfplay_config, seed, 48); SC_FIELD(sc_selfplay_config, first_game_id, 56); SC_FIELD(sc_selfplay_config, trace_capacity, 64);
SC_FIELD(sc_selfplay_config, own_stream, 68); SC_FIELD(sc_selfplay_config, tie_random, 72); SC_FIELD(sc_selfplay_config, trace_hold, 76);
SC_FIELD(sc_selfplay_config, rollout_factor, 80);
SC_LAYOUT(sc_selfplay_stats, 32);
SC_FIELD(sc_selfplay_stats, sims_done, 0); SC_FIELD(sc_selfplay_stats, nn_evals, 8); SC_FIELD(sc_selfplay_stats, games_finished, 16);
SC_FIELD(sc_selfplay_stats, games_active, 20); SC_FIELD(sc_selfplay_stats, error_flags, 24); SC_FIELD(sc_selfplay_stats, plies_done, 28);
SC_LAYOUT(sc_trace_info, 32);
SC_FIELD(sc_trace_info, n_steps, 0); SC_FIELD(sc_trace_info, n_children_total, 4); SC_FIELD(sc_trace_info, has_outcome, 8);
SC_FIELD(sc_trace_info, termination, 12); SC_FIELD(sc_trace_info, winner, 16); SC_FIELD(sc_trace_info, game_id, 24);
#undef SC_LAYOUT
#undef SC_FIELD

static thread_local std::string g_err;
static thread_local float g_encode_ms[2] = {0.f, 0.f};   // last sc_encode_steps of this thread: kernels (HIP events), whole call
static thread_local std::string g_warn;
static int fail(const std::string& m, int code = -1) {
    g_err = m;
    return code;
}
#define HIPOK(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_), -2);     \
    } while (0)

template <class T>
static hipError_t dalloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
}

struct sc_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    scnn::NetDev net{};
    uint16_t* d_wb = nullptr;
    float* d_wf = nullptr;
    int n_cu = 256;    // compute units of the device
    int step_blocks_per_cu = 1;   // resident fused step workgroups per CU for this network (1 or 2)
    int ksplit = 64;   // split-K of value_head.ffn.0 (the search kernel's fused tail sums 32 or 64 partials)
    // scratch, grown on demand
    int cap = 0;
    int8_t* d_boards = nullptr;
    int32_t* d_meta = nullptr;
    uint16_t* d_lidx = nullptr;
    int32_t* d_nlegal = nullptr;
    float* d_prior = nullptr;
    float* d_value = nullptr;
    float* d_logp = nullptr;
    scnn::bf16_t* d_hval = nullptr;
    float* d_vpart = nullptr;
    float* d_dbg = nullptr;
    // scratch arena of sc_encode_positions (Level-1 callers encode one position per call: no malloc/free per call)
    void* d_enc = nullptr;
    size_t enc_cap = 0;
    // the one-slot handle sc_search keeps between calls (NNPlayer::bestmove calls it once per move: building and freeing ~35 device
    // buffers per call cost more than a short search) and the rollout its node pools are sized for
    struct sc_selfplay* search_sp = nullptr;
    int search_rollout_cap = 0;
};

static void engine_free_scratch(sc_engine* e) {
    dfree({e->d_boards, e->d_meta, e->d_lidx, e->d_nlegal, e->d_prior});
    dfree({e->d_value, e->d_logp, e->d_hval, e->d_vpart, e->d_dbg});
    e->d_boards = nullptr; e->d_meta = nullptr; e->d_lidx = nullptr; e->d_nlegal = nullptr; e->d_prior = nullptr;
    e->d_value = nullptr; e->d_logp = nullptr; e->d_hval = nullptr; e->d_vpart = nullptr; e->d_dbg = nullptr;
    e->cap = 0;
    dfree({e->d_enc});
    e->d_enc = nullptr;
    e->enc_cap = 0;
}
static int engine_reserve(sc_engine* e, int n) {
    if (n <= e->cap) return 0;
    HIPOK(hipStreamSynchronize(e->stream));
    engine_free_scratch(e);
    int cap = std::max(n, 64);
    HIPOK(dalloc(&e->d_boards, (size_t)cap * 7168));
    HIPOK(dalloc(&e->d_meta, (size_t)cap * 8));
    HIPOK(dalloc(&e->d_lidx, (size_t)cap * 224));
    HIPOK(dalloc(&e->d_nlegal, (size_t)cap));
    HIPOK(dalloc(&e->d_prior, (size_t)cap * 224));
    HIPOK(dalloc(&e->d_value, (size_t)cap));
    HIPOK(dalloc(&e->d_logp, (size_t)cap * 4672));
    HIPOK(dalloc(&e->d_hval, (size_t)cap * 64 * 256));
    HIPOK(dalloc(&e->d_vpart, (size_t)e->ksplit * cap * 128));
    HIPOK(dalloc(&e->d_dbg, (size_t)cap * 64 * 256));
    e->cap = cap;
    return 0;
}

// enqueue the three network kernels for n positions (device pointers)
static void enqueue_forward(sc_engine* e, int n, const int8_t* boards, const int32_t* meta, int meta_stride,
                            const uint16_t* lidx, const int32_t* nlegal, float* prior, float* value, float* logp, float* dbg,
                            int dbg_stage, hipStream_t s) {
    scnn::TowerArgs t{};
    t.net = e->net;
    t.n_pos = n;
    t.boards = boards;
    t.meta = meta;
    t.meta_stride = meta_stride;
    t.legal_idx = lidx;
    t.n_legal = nlegal;
    t.prior = (lidx && nlegal) ? prior : nullptr;
    t.logp = logp;
    t.hval = e->d_hval;
    t.dbg = dbg;
    t.dbg_stage = dbg ? dbg_stage : -1;
    scl::tower(t, s);
    scnn::Fc1Args f{};
    f.net = e->net;
    f.n_pos = n;
    f.ksplit = e->ksplit;
    f.hval = e->d_hval;
    f.vpart = e->d_vpart;
    scl::value_fc1(f, s);
    scnn::VfinArgs v{};
    v.net = e->net;
    v.n_pos = n;
    v.ksplit = e->ksplit;
    v.vpart = e->d_vpart;
    v.meta = meta;
    v.meta_stride = meta_stride;
    v.value = value;
    scl::value_finish(v, s);
}

extern "C" {

const char* sc_last_error(void) { return g_err.c_str(); }

int sc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Kernel arguments in device memory: with the default (host-coherent) placement every launch starts with a PCIe round
// trip for its argument block -- three launches per simulation step, +6 % simulations/s measured.  The HIP runtime reads the
// switch when it initialises, so it is set when this library is loaded (no effect if the host application has already
// initialised HIP: export HIP_FORCE_DEV_KERNARG=1 there).  An explicit setting of the host wins.
// SC_ENGINE_KEEP_ENV=1 disables the hook.  sc_runtime_flags() reports what happened (include/sc_engine.h).
namespace {
struct RuntimeEnv {
    int flags = 0;
    RuntimeEnv() {
        const char* keep = getenv("SC_ENGINE_KEEP_ENV");
        if (keep && keep[0] == '1') {
            flags |= 4;
        } else if (!getenv("HIP_FORCE_DEV_KERNARG")) {
            setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
            flags |= 2;
        }
    }
} g_runtime_env;
}  // namespace

int sc_runtime_flags(void) {
    const char* v = getenv("HIP_FORCE_DEV_KERNARG");
    const int on = (v && v[0] == '1' && v[1] == 0) ? 1 : 0;
    return on | (on ? (g_runtime_env.flags & 2) : 0) | (g_runtime_env.flags & 4);
}
const char* sc_last_warning(void) { return g_warn.c_str(); }

int sc_engine_create(const sc_net_config* cfg, const char* weights_path, int device_id, sc_engine** out) {
    if (!cfg || !out) return fail("null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("no HIP device available: libsc_engine has no CPU fallback", -3);
    if (device_id < 0 || device_id >= ndev) return fail("device_id out of range");
    HIPOK(hipSetDevice(device_id));
    scw::HostWeights hw;
    if (weights_path) {
        std::string err = scw::load_scw(weights_path, hw);
        if (!err.empty()) return fail(err);
    } else {
        if (cfg->channels != 128 && cfg->channels != 256) return fail("channels must be 128 or 256");
        if (cfg->n_res_blocks < 0 || cfg->n_res_blocks > 80) return fail("n_res_blocks out of range");
        hw = scw::init_prng(cfg->n_res_blocks, cfg->channels, cfg->seed);
    }
    if (cfg->precision != SC_PREC_BF16 && cfg->precision != SC_PREC_FP8) return fail("unknown precision");
    if (cfg->precision == SC_PREC_FP8) hw.fp8 = true;   // (an SCW2 fp8 blob sets it by itself)
    // Both trunk widths run the channel-major 32x32x16 tower.  Experiment builds also carry the
    // pixel-major 16x16x32 kernel; SC_TOWER_V=1 selects it there (developer switch, ignored by the production build).
    const int tv = getenv("SC_TOWER_V") ? atoi(getenv("SC_TOWER_V")) : 0;
    bool v32 = tv != 1;
    if (!scl::tower_variant_available(hw.C, v32)) v32 = true;
    scw::Packed pk = scw::pack(hw, v32);
    sc_engine* e = new sc_engine();
    e->device = device_id;
    // every failure from here on goes through sc_engine_destroy (stream, weight blobs)
    auto bail = [&](hipError_t err, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(err);
        sc_engine_destroy(e);
        return fail(m, -2);
    };
    const char* ierr = scl::nn_init();
    if (!ierr) ierr = scl::step_init();
    if (ierr) {
        sc_engine_destroy(e);
        return fail(std::string("kernel attribute setup failed: ") + ierr);
    }
    hipError_t he;
    if ((he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess) return bail(he, "hipStreamCreate");
    if ((he = dalloc(&e->d_wb, pk.wb.size())) != hipSuccess) return bail(he, "hipMalloc(weights)");
    if ((he = dalloc(&e->d_wf, pk.wf.size())) != hipSuccess) return bail(he, "hipMalloc(parameters)");
    if ((he = hipMemcpy(e->d_wb, pk.wb.data(), pk.wb.size() * 2, hipMemcpyHostToDevice)) != hipSuccess) return bail(he, "hipMemcpy(weights)");
    if ((he = hipMemcpy(e->d_wf, pk.wf.data(), pk.wf.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(he, "hipMemcpy(parameters)");
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && ncu > 0) e->n_cu = ncu;
    }
    static_cast<scnn::NetLayout&>(e->net) = pk.lay;
    e->step_blocks_per_cu = std::min(2, scl::step_blocks_per_cu(pk.lay));
    e->net.wb = e->d_wb;
    e->net.wf = e->d_wf;
    const int rf = sc_runtime_flags();
    g_warn.clear();
    if (!(rf & 1))
        g_warn = "HIP_FORCE_DEV_KERNARG=1 is not set: kernel arguments travel through host-coherent memory (about -6 % simulations/s)";
    else if (rf & 2)
        g_warn = "HIP_FORCE_DEV_KERNARG=1 was set by libsc_engine's load hook: it has no effect if the host initialised HIP before "
                 "loading the library (export it in the host's environment instead)";
    *out = e;
    return 0;
}

void sc_engine_destroy(sc_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->search_sp) {
        sc_selfplay_destroy(e->search_sp);
        e->search_sp = nullptr;
    }
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    engine_free_scratch(e);
    dfree({e->d_wb});
    dfree({e->d_wf});
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int sc_engine_max_batch(const sc_engine*) { return 65536; }
int sc_engine_precision(const sc_engine* e) { return e && e->net.fp8 ? SC_PREC_FP8 : SC_PREC_BF16; }
int sc_engine_synchronize(sc_engine* e) {
    if (!e) return fail("null engine");
    HIPOK(hipSetDevice(e->device));
    HIPOK(hipStreamSynchronize(e->stream));
    return 0;
}

static int forward_host(sc_engine* e, int n, const int8_t* boards, const int32_t* meta, const uint16_t* lidx_rows,
                        const int32_t* nlegal, float* prior_rows, float* value, float* logp, float* dbg, int dbg_stage) {
    if (!e || !boards || !meta || n < 0) return fail("bad argument");
    if (n == 0) return 0;
    HIPOK(hipSetDevice(e->device));
    int rc = engine_reserve(e, n);
    if (rc) return rc;
    hipStream_t s = e->stream;
    HIPOK(hipMemcpyAsync(e->d_boards, boards, (size_t)n * 7168, hipMemcpyHostToDevice, s));
    {
        std::vector<int32_t> m8((size_t)n * 8, 0);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 7; k++) m8[(size_t)i * 8 + k] = meta[(size_t)i * 7 + k];
        HIPOK(hipMemcpy(e->d_meta, m8.data(), m8.size() * 4, hipMemcpyHostToDevice));
    }
    if (lidx_rows) {
        HIPOK(hipMemcpyAsync(e->d_lidx, lidx_rows, (size_t)n * 224 * 2, hipMemcpyHostToDevice, s));
        HIPOK(hipMemcpyAsync(e->d_nlegal, nlegal, (size_t)n * 4, hipMemcpyHostToDevice, s));
    }
#ifdef SC_EXP   // experiment builds: the same launches back to back first (steady clocks for in-kernel stamps, tools/dbg_clock.py)
    for (int rep = getenv("SC_EXP_REPEAT") ? atoi(getenv("SC_EXP_REPEAT")) : 0; rep > 0; rep--)
        enqueue_forward(e, n, e->d_boards, e->d_meta, 8, lidx_rows ? e->d_lidx : nullptr, lidx_rows ? e->d_nlegal : nullptr,
                        e->d_prior, e->d_value, logp ? e->d_logp : nullptr, dbg ? e->d_dbg : nullptr, dbg_stage, s);
#endif
    enqueue_forward(e, n, e->d_boards, e->d_meta, 8, lidx_rows ? e->d_lidx : nullptr, lidx_rows ? e->d_nlegal : nullptr,
                    e->d_prior, e->d_value, logp ? e->d_logp : nullptr, dbg ? e->d_dbg : nullptr, dbg_stage, s);
    HIPOK(hipGetLastError());
    if (value) HIPOK(hipMemcpyAsync(value, e->d_value, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (logp) HIPOK(hipMemcpyAsync(logp, e->d_logp, (size_t)n * 4672 * 4, hipMemcpyDeviceToHost, s));
    if (prior_rows) HIPOK(hipMemcpyAsync(prior_rows, e->d_prior, (size_t)n * 224 * 4, hipMemcpyDeviceToHost, s));
    if (dbg) HIPOK(hipMemcpyAsync(dbg, e->d_dbg, (size_t)n * 64 * e->net.C * 4, hipMemcpyDeviceToHost, s));
    HIPOK(hipStreamSynchronize(s));
    return 0;
}

int sc_forward_batch(sc_engine* e, int n, const int8_t* boards, const int32_t* meta, float* logp, float* value) {
    return forward_host(e, n, boards, meta, nullptr, nullptr, nullptr, value, logp, nullptr, -1);
}

/* debugging aid for tests: residual stream after `stage` (0 stem, b block b, 1000 latent): out[n][64][C] */
int sc_forward_debug(sc_engine* e, int n, const int8_t* boards, const int32_t* meta, int stage, float* out) {
    return forward_host(e, n, boards, meta, nullptr, nullptr, nullptr, nullptr, nullptr, out, stage);
}

int sc_predict_batch(sc_engine* e, int n, const int8_t* boards, const int32_t* meta, const uint16_t* legal_idx,
                     const uint32_t* legal_off, float* priors, float* value) {
    if (!legal_idx || !legal_off || !priors) return fail("bad argument");
    std::vector<uint16_t> rows((size_t)n * 224, 0);
    std::vector<int32_t> nl((size_t)n);
    for (int i = 0; i < n; i++) {
        uint32_t a = legal_off[i], b = legal_off[i + 1];
        if (b < a || b - a > 218) return fail("legal_off: more than 218 moves for one position");
        nl[(size_t)i] = (int32_t)(b - a);
        for (uint32_t k = a; k < b; k++) {
            if (legal_idx[k] >= 4672) return fail("legal_idx out of range");
            rows[(size_t)i * 224 + (k - a)] = legal_idx[k];
        }
    }
    std::vector<float> pr((size_t)n * 224);
    int rc = forward_host(e, n, boards, meta, rows.data(), nl.data(), pr.data(), value, nullptr, nullptr, -1);
    if (rc) return rc;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < nl[(size_t)i]; k++) priors[legal_off[i] + (uint32_t)k] = pr[(size_t)i * 224 + k];
    return 0;
}

/* Game::predict with argmax = true: post_process_distr's first branch (src/chess.rs:880-889) -- a one-hot vector at the
 * LAST maximal prior of each position (Iterator::max_by).  The maximum is taken over the renormalised priors: the
 * reference takes it over exp(logp[idx]) before the division, which orders the moves the same way except where two
 * different values round to the same quotient. */
int sc_predict_batch_argmax(sc_engine* e, int n, const int8_t* boards, const int32_t* meta, const uint16_t* legal_idx,
                            const uint32_t* legal_off, float* priors, float* value) {
    int rc = sc_predict_batch(e, n, boards, meta, legal_idx, legal_off, priors, value);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        const uint32_t a = legal_off[i], b = legal_off[i + 1];
        if (b == a) continue;
        uint32_t best = a;
        for (uint32_t k = a + 1; k < b; k++)
            if (!(priors[k] < priors[best])) best = k;   // >= : the last maximum; a NaN prior wins like partial_cmp().unwrap() would panic
        for (uint32_t k = a; k < b; k++) priors[k] = k == best ? 1.0f : 0.0f;
    }
    return 0;
}

int sc_encode_positions(sc_engine* e, int device_id, int n, const uint16_t* moves, const uint32_t* move_off, int8_t* boards,
                        int32_t* meta, uint16_t* legal_moves, uint16_t* legal_idx, int32_t* n_legal, int32_t* outcome) {
    if (n < 0 || !move_off) return fail("bad argument");
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device available: libsc_engine has no CPU fallback", -3);
    int dev = e ? e->device : device_id;
    HIPOK(hipSetDevice(dev));
    uint32_t total = move_off[n];
    uint32_t maxlen = 0;
    for (int i = 0; i < n; i++) {
        if (move_off[i + 1] < move_off[i]) return fail("move_off not monotonic");
        maxlen = std::max(maxlen, move_off[i + 1] - move_off[i]);
    }
    if (maxlen > 4000) return fail("move list too long");
    int hist_cap = (int)maxlen + 2;
    // one arena for all device buffers of the call; with an engine it persists (grown on demand) and the copies run on
    // the engine's stream behind a single synchronisation
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_moves = take((size_t)total * 2 + 2), o_off = take(((size_t)n + 1) * 4), o_hist = take((size_t)n * hist_cap * sizeof(sc::Position));
    const size_t o_boards = take((size_t)n * 7168), o_meta = take((size_t)n * 28), o_nl = take((size_t)n * 4), o_out = take((size_t)n * 16);
    const size_t o_lm = take((size_t)n * 448), o_li = take((size_t)n * 448);
    char* base = nullptr;
    hipStream_t st = e ? e->stream : nullptr;
    if (e) {
        if (off > e->enc_cap) {
            HIPOK(hipStreamSynchronize(e->stream));
            dfree({e->d_enc});
            e->d_enc = nullptr;
            e->enc_cap = 0;
            HIPOK(hipMalloc(&e->d_enc, off));
            e->enc_cap = off;
        }
        base = static_cast<char*>(e->d_enc);
    } else {
        HIPOK(hipMalloc(reinterpret_cast<void**>(&base), off));
    }
    uint16_t* d_moves = reinterpret_cast<uint16_t*>(base + o_moves);
    uint32_t* d_off = reinterpret_cast<uint32_t*>(base + o_off);
    sc::Position* d_hist = reinterpret_cast<sc::Position*>(base + o_hist);
    int8_t* d_boards = reinterpret_cast<int8_t*>(base + o_boards);
    int32_t* d_meta = reinterpret_cast<int32_t*>(base + o_meta);
    int32_t* d_nl = reinterpret_cast<int32_t*>(base + o_nl);
    int32_t* d_out = reinterpret_cast<int32_t*>(base + o_out);
    uint16_t* d_lm = reinterpret_cast<uint16_t*>(base + o_lm);
    uint16_t* d_li = reinterpret_cast<uint16_t*>(base + o_li);
    int rc = 0;
    auto ok = [&](hipError_t err) {
        if (err != hipSuccess && !rc) rc = fail(hipGetErrorString(err), -2);
    };
    if (total) ok(hipMemcpyAsync(d_moves, moves, (size_t)total * 2, hipMemcpyHostToDevice, st));
    ok(hipMemcpyAsync(d_off, move_off, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st));
    ok(hipMemsetAsync(d_lm, 0, (size_t)n * 448, st));   // rows are zero past n_legal (take() pads each region to 256 B:
    ok(hipMemsetAsync(d_li, 0, (size_t)n * 448, st));   // the two tables are not adjacent in general)
    if (!rc) scl::encode_positions(n, d_moves, d_off, nullptr, d_hist, hist_cap, d_boards, d_meta, d_lm, d_li, d_nl, d_out, st);
    ok(hipGetLastError());
    if (boards) ok(hipMemcpyAsync(boards, d_boards, (size_t)n * 7168, hipMemcpyDeviceToHost, st));
    if (meta) ok(hipMemcpyAsync(meta, d_meta, (size_t)n * 7 * 4, hipMemcpyDeviceToHost, st));
    if (legal_moves) ok(hipMemcpyAsync(legal_moves, d_lm, (size_t)n * 224 * 2, hipMemcpyDeviceToHost, st));
    if (legal_idx) ok(hipMemcpyAsync(legal_idx, d_li, (size_t)n * 224 * 2, hipMemcpyDeviceToHost, st));
    if (n_legal) ok(hipMemcpyAsync(n_legal, d_nl, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (outcome) ok(hipMemcpyAsync(outcome, d_out, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    ok(hipStreamSynchronize(st));
    if (!e) dfree({base});
    return rc;
}

// libsmartchess.chess_encode_steps (reference src/lib.rs:46-128) for a batch of recorded games; see include/sc_engine.h
int sc_encode_steps(sc_engine* e, int device_id, int n_games, const uint16_t* moves, const uint32_t* move_off,
                    const uint16_t* child_mv, const uint32_t* child_n, const uint32_t* child_off, int apply_mirror, int8_t* boards,
                    int32_t* meta, float* dist, uint16_t* legal_idx, int32_t* n_legal, int32_t* status) {
    if (n_games < 0 || !move_off || !child_off || !status) return fail("bad argument");
    if (n_games == 0) return 0;
    const auto t_call = std::chrono::steady_clock::now();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device available: libsc_engine has no CPU fallback", -3);
    HIPOK(hipSetDevice(e ? e->device : device_id));
    const uint32_t total = move_off[n_games];
    uint32_t maxlen = 0;
    for (int g = 0; g < n_games; g++) {
        if (move_off[g + 1] < move_off[g]) return fail("move_off not monotonic");
        maxlen = std::max(maxlen, move_off[g + 1] - move_off[g]);
        status[g] = 0;
    }
    if (maxlen > 4000) return fail("move list too long");
    for (uint32_t p = 0; p < total; p++)
        if (child_off[p + 1] < child_off[p] || child_off[p + 1] - child_off[p] > 224) return fail("child_off: more than 224 children or not monotonic");
    if (total == 0) return 0;
    // one position per ply: (record index of its game's start position, number of moves already played)
    const int hist_cap = (int)maxlen + 2;
    std::vector<uint32_t> pstart(total), plen(total), pgame(total);
    for (int g = 0; g < n_games; g++)
        for (uint32_t t = move_off[g]; t < move_off[g + 1]; t++) {
            pstart[t] = (uint32_t)g * (uint32_t)hist_cap;
            plen[t] = t - move_off[g];
            pgame[t] = (uint32_t)g;
        }
    if ((size_t)n_games * (size_t)hist_cap > (size_t)0xffffffffu) return fail("too many game records for one call");
    const uint32_t CH = 8192;  // plies per launch of the per-ply kernels (bounds the output staging: CH * 26 KB)
    const uint32_t nchild = child_off[total];
    uint16_t *d_moves = nullptr, *d_cmv = nullptr, *d_lm = nullptr, *d_li = nullptr;
    uint32_t *d_start = nullptr, *d_len = nullptr, *d_cn = nullptr, *d_coff = nullptr, *d_moff = nullptr;
    sc::Position* d_hist = nullptr;
    int8_t* d_boards = nullptr;
    int32_t *d_meta = nullptr, *d_nl = nullptr, *d_flags = nullptr, *d_out = nullptr;
    float* d_dist = nullptr;
    const uint32_t cap = std::min(CH, total);
    HIPOK(dalloc(&d_moves, total));
    HIPOK(dalloc(&d_cmv, (size_t)nchild + 1));
    HIPOK(dalloc(&d_cn, (size_t)nchild + 1));
    HIPOK(dalloc(&d_coff, (size_t)cap + 1));
    HIPOK(dalloc(&d_start, total));   // (all plies: the key / repetition kernels run over the whole batch at once)
    HIPOK(dalloc(&d_len, total));
    HIPOK(dalloc(&d_hist, (size_t)n_games * hist_cap));   // one 80-byte record per ply of every game (k_replay_games)
    HIPOK(dalloc(&d_moff, (size_t)n_games + 1));
    HIPOK(dalloc(&d_boards, (size_t)cap * 7168));
    HIPOK(dalloc(&d_meta, (size_t)cap * 7));
    HIPOK(dalloc(&d_nl, cap));
    HIPOK(dalloc(&d_flags, cap));
    HIPOK(dalloc(&d_out, (size_t)cap * 4));
    HIPOK(dalloc(&d_lm, (size_t)cap * 224));
    HIPOK(dalloc(&d_li, (size_t)cap * 224));
    HIPOK(dalloc(&d_dist, (size_t)cap * 4672));
    HIPOK(hipMemcpy(d_moves, moves, (size_t)total * 2, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_moff, move_off, ((size_t)n_games + 1) * 4, hipMemcpyHostToDevice));
    if (nchild) {
        HIPOK(hipMemcpy(d_cmv, child_mv, (size_t)nchild * 2, hipMemcpyHostToDevice));
        HIPOK(hipMemcpy(d_cn, child_n, (size_t)nchild * 4, hipMemcpyHostToDevice));
    }
    std::vector<int32_t> flags(cap), replay(4 * (size_t)cap);
    std::vector<uint32_t> coff(cap + 1);
    hipEvent_t evk[2] = {nullptr, nullptr};
    HIPOK(hipEventCreate(&evk[0]));
    HIPOK(hipEventCreate(&evk[1]));
    float kernels_ms = 0.f;
    // every game is walked once (board updates only: one record per ply), keys and repetition flags of all plies follow in parallel;
    // then the plies are encoded from the records, 8 192 at a time
    HIPOK(hipMemcpy(d_start, pstart.data(), (size_t)total * 4, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_len, plen.data(), (size_t)total * 4, hipMemcpyHostToDevice));
    HIPOK(hipEventRecord(evk[0], nullptr));
    scl::replay_games(n_games, (int)total, d_moves, d_moff, d_hist, hist_cap, d_start, d_len, nullptr);
    HIPOK(hipEventRecord(evk[1], nullptr));
    HIPOK(hipGetLastError());
    HIPOK(hipDeviceSynchronize());
    HIPOK(hipEventElapsedTime(&kernels_ms, evk[0], evk[1]));
    for (uint32_t p0 = 0; p0 < total; p0 += CH) {
        const uint32_t n = std::min(CH, total - p0);
        for (uint32_t i = 0; i <= n; i++) coff[i] = child_off[p0 + i];   // absolute offsets into d_cmv / d_cn
        HIPOK(hipMemcpy(d_coff, coff.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
        HIPOK(hipEventRecord(evk[0], nullptr));
        scl::encode_plies((int)n, d_hist, d_start + p0, d_len + p0, d_boards, d_meta, d_lm, d_li, d_nl, nullptr);
        // the ply's own move is moves[start + len] = d_moves + p0 + i
        scl::steps_dist((int)n, d_lm, d_nl, d_moves + p0, d_cmv, d_cn, d_coff, apply_mirror, d_meta, d_dist, d_flags, nullptr);
        HIPOK(hipEventRecord(evk[1], nullptr));
        HIPOK(hipGetLastError());
        HIPOK(hipDeviceSynchronize());
        {
            float ms = 0.f;
            HIPOK(hipEventElapsedTime(&ms, evk[0], evk[1]));
            kernels_ms += ms;
        }
        if (boards) HIPOK(hipMemcpy(boards + (size_t)p0 * 7168, d_boards, (size_t)n * 7168, hipMemcpyDeviceToHost));
        if (meta) HIPOK(hipMemcpy(meta + (size_t)p0 * 7, d_meta, (size_t)n * 28, hipMemcpyDeviceToHost));
        if (dist) HIPOK(hipMemcpy(dist + (size_t)p0 * 4672, d_dist, (size_t)n * 4672 * 4, hipMemcpyDeviceToHost));
        if (legal_idx) HIPOK(hipMemcpy(legal_idx + (size_t)p0 * 224, d_li, (size_t)n * 224 * 2, hipMemcpyDeviceToHost));
        if (n_legal) HIPOK(hipMemcpy(n_legal + p0, d_nl, (size_t)n * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(flags.data(), d_flags, (size_t)n * 4, hipMemcpyDeviceToHost));
        // first failing ply of each game, with the reference's precedence (children first, then the played move)
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t g = pgame[p0 + i];
            if (status[g] != 0) continue;
            const int ply = (int)plen[p0 + i];
            if (flags[i] & 1) status[g] = 1000 + ply;
            else if (flags[i] & 2) status[g] = -(ply + 1);
        }
    }
    dfree({d_moves, d_cmv, d_cn, d_coff, d_start, d_len, d_hist, d_moff});
    dfree({d_boards, d_meta, d_nl, d_flags, d_out, d_lm, d_li, d_dist});
    (void)hipEventDestroy(evk[0]);
    (void)hipEventDestroy(evk[1]);
    g_encode_ms[0] = kernels_ms;
    g_encode_ms[1] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    return 0;
}

int sc_encode_steps_last_timing(float* kernels_ms, float* total_ms) {
    if (kernels_ms) *kernels_ms = g_encode_ms[0];
    if (total_ms) *total_ms = g_encode_ms[1];
    return 0;
}

}  // extern "C"

// ============================================================================================
// self-play
// ============================================================================================
// Step launches that compute value_head.ffn.0 inside the launch (sp->fc1_in_step) make workgroups wait for other workgroups
// of the same launch.  Two such launches running side by side on one device (two streams) could each hold compute units the
// other's late workgroups need: on one device the form is therefore granted to the handles of ONE stream at a time (the first
// to ask; an engine's handles share its stream and run one after the other); any other handle uses the two-launch form.
// (Two PROCESSES sharing a GPU are not covered: the waits are bounded -- error flag 32 -- but one self-play process per GPU
// is the deployment this library is written for.)
static std::mutex g_fc1_mu;
static std::map<int, std::pair<hipStream_t, int>> g_fc1_stream;   // device -> (stream, live handles)
// A device on which an in-launch hand-off has timed out once (the workgroups of a launch were not all resident: someone else
// is using the GPU) is not trusted with that form again by this process: handles created afterwards use the two-launch form,
// whose launches do not wait for each other.
static std::map<int, bool> g_fc1_failed;
static bool fc1_stream_acquire(int device, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_fc1_mu);
    if (g_fc1_failed.count(device)) return false;
    auto it = g_fc1_stream.find(device);
    if (it == g_fc1_stream.end() || it->second.second == 0) {
        g_fc1_stream[device] = {s, 1};
        return true;
    }
    if (it->second.first != s) return false;
    it->second.second++;
    return true;
}
static void fc1_stream_release(int device) {
    std::lock_guard<std::mutex> lk(g_fc1_mu);
    auto it = g_fc1_stream.find(device);
    if (it != g_fc1_stream.end() && it->second.second > 0) it->second.second--;
}

struct sc_selfplay {
    sc_engine* engine = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    sc_selfplay_config cfg{};
    sc::SpParams p{};
    std::vector<void*> allocs;
    int64_t sim_steps_enqueued = 0;
    // timing
    int timing_stride = 0;       // > 0: every n-th step runs as separate launches with the TOWER bracketed by an event pair
    int step_stride = 0;         // > 0: every n-th step is bracketed as a whole, in the handle's own launch form
    std::vector<hipEvent_t> ev;  // pairs
    int ev_next = 0;
    int64_t ev_recorded = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool have_span = false;
    int64_t nn_launches = 0;
    bool pending_final = false;
    // match play (sc_selfplay_set_players): player of even / odd plies
    bool match = false;
    sc_engine* player[2] = {nullptr, nullptr};
    uint64_t salt[2] = {0, 0};
    scnn::bf16_t* d_hval = nullptr;  // value-head features of the current leaves [n_slots][64][256]
    float* d_vpart = nullptr;        // split-K partials of value_head.ffn.0 [ksplit][n_slots][128]
    // streaming drain (sc_selfplay_poll): per trace-ring row, the game id last reported to the host (+1; 0 = none)
    std::vector<uint64_t> reported;
    std::vector<int> to_release;     // rows handed out by the previous poll (trace_hold)
    // search wave + tower in one launch (step_kernels.hip).  Chosen at creation: only with at most one game per compute
    // unit and a single group -- with more games than CUs the separate search launch runs all of them at once while the
    // fused workgroups (83 KB of LDS: one per CU) would take turns, and with several interleaved groups one group's search
    // launch is what hides under another group's tower (measured: 512 games 3.48 M vs 3.02 M simulations/s at fp8, two
    // groups of 256 2.65 M vs 2.38 M at bf16, in favour of the separate launches)
    bool fused = false;
    bool fc1_in_step = false;        // ... and value_head.ffn.0 runs inside that launch too (one launch per simulation step)
    uint32_t* d_fc1_ctr = nullptr;   // its arrival counters, one per 64-position block, 128 B apart (monotonic)
    uint32_t fc1_launches = 0;       // step launches that counted on them so far
    uint32_t fc1_target_skew = 0;    // test aid (sc_selfplay_debug_break_handoff): arrivals that will never come
    // An internal hand-off of a step launch timed out (error_flags & (16 | 32)): tiles were computed from stale rows, the
    // values backed up since are wrong.  Latched when the host first sees the flag; from then on the handle refuses work.
    bool poisoned = false;
};

static int sp_refuse(const sc_selfplay*) {
    return fail("an internal hand-off of this handle's step launches timed out (error_flags & 48): its trees and traces are "
                "invalid; destroy the handle -- a new handle on this device uses the two-launch step", SC_ERR_HANDOFF);
}
// called with the stream idle: looks at the device's error word
static int sp_latch(sc_selfplay* sp) {
    if (sp->poisoned) return 0;
    int32_t err = 0;
    HIPOK(hipMemcpy(&err, reinterpret_cast<const char*>(sp->p.cnt) + offsetof(sc::Counters, err), 4, hipMemcpyDeviceToHost));
    if (err & (sc::ERR_HELPER_TIMEOUT | sc::ERR_HANDOFF_TIMEOUT)) {
        sp->poisoned = true;
        if (err & sc::ERR_HANDOFF_TIMEOUT) {
            std::lock_guard<std::mutex> lk(g_fc1_mu);
            g_fc1_failed[sp->device] = true;
        }
    }
    return 0;
}

// complete the last enqueued simulation (expand / backward / ply transition) so that host reads see a
// fully backed-up state; a following enqueue would have done the same work in its first launch
// match play: the value tail of simulation step t uses the weights of the player that evaluated step t
static void match_tail_params(const sc_selfplay* sp, sc::SpParams& q, int64_t t) {
    if (!sp->match || sp->p.evaluator != SC_EVAL_NET || t < 0) return;
    const sc_engine* ep = sp->player[(t / sp->p.rollout) & 1];
    q.vf_w = ep->d_wf;
    q.vf_fc1b = (uint32_t)ep->net.f_fc1b;
    q.vf_fc1m = (uint32_t)ep->net.f_fc1m;
    q.vf_fc2w = (uint32_t)ep->net.f_fc2w;
    q.vf_fc2b = (uint32_t)ep->net.f_fc2b;
}
static void sp_flush(sc_selfplay* sp) {
    if (sp->pending_final) {
        sc::SpParams q = sp->p;
        match_tail_params(sp, q, sp->sim_steps_enqueued - 1);
        scl::mcts(q, 1, 0, sp->stream);
        sp->pending_final = false;
    }
}

template <class T>
static int sp_alloc(sc_selfplay* sp, T** ptr, size_t n, bool zero = true) {
    HIPOK(dalloc(ptr, n));
    sp->allocs.push_back(*ptr);
    if (zero) HIPOK(hipMemset(*ptr, 0, std::max<size_t>(n, 1) * sizeof(T)));
    return 0;
}

extern "C" {

int sc_selfplay_create(sc_engine* e, int device_id, const sc_selfplay_config* cfg, sc_selfplay** out) {
    if (!cfg || !out) return fail("null argument");
    *out = nullptr;
    if (cfg->evaluator == SC_EVAL_NET && !e) return fail("SC_EVAL_NET needs an engine");
    if (cfg->n_slots <= 0 || cfg->n_games <= 0 || cfg->rollout_num < 1 || cfg->num_steps < 1 || cfg->num_steps > 4000)
        return fail("bad self-play configuration");
    if (cfg->rollout_num > 60000) return fail("rollout_num too large");
    if (cfg->evaluator < SC_EVAL_NET || cfg->evaluator > SC_EVAL_SYNTH_UNIFORM) return fail("unknown evaluator");
    if (cfg->rollout_factor < 0.f || (cfg->rollout_factor > 0.f && cfg->rollout_num != 300))
        return fail("rollout_factor needs rollout_num = 300 (the cap of min(300, n_legal * factor), src/main.rs:176)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device available: libsc_engine has no CPU fallback", -3);
    int dev = e ? e->device : device_id;
    if (dev < 0 || dev >= ndev) return fail("device_id out of range");
    HIPOK(hipSetDevice(dev));
    sc_selfplay* sp = new sc_selfplay();
    sp->engine = e;
    sp->device = dev;
    sp->cfg = *cfg;
    // every failure from here on goes through sc_selfplay_destroy
    auto bail = [&](hipError_t err, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(err);
        sc_selfplay_destroy(sp);
        return fail(m, -2);
    };
    if (e && !cfg->own_stream) {
        sp->stream = e->stream;
    } else {
        hipError_t he = hipStreamCreateWithFlags(&sp->stream, hipStreamNonBlocking);
        if (he != hipSuccess) {
            sp->stream = nullptr;
            return bail(he, "hipStreamCreate");
        }
        sp->own_stream = true;
    }
    sc::SpParams& p = sp->p;
    p.n_slots = cfg->n_slots;
    p.rollout = cfg->rollout_num;
    p.num_steps = cfg->num_steps;
    p.temp_switch = cfg->temperature_switch;
    p.with_noise = cfg->with_noise;
    p.outcome_gate = cfg->outcome_gate;
    p.evaluator = cfg->evaluator;
    p.external_noise = cfg->external_noise;
    p.tie_random = cfg->tie_random;
    p.trace_hold = cfg->trace_hold ? 1 : 0;
    p.rollout_factor = cfg->rollout_factor;
    p.synth_salt = 0;
    p.cpuct = cfg->cpuct;
    p.temperature = cfg->temperature;
    p.epsilon = cfg->epsilon;
    p.seed = cfg->seed;
    p.first_game_id = cfg->first_game_id;
    p.node_cap = 1 + cfg->rollout_num * 218;         // worst case: every expansion adds 218 children
    p.max_depth = std::min(cfg->rollout_num + 2, 1024);  // the path is tracked in LDS (mcts_kernels.hpp DEPTH_LDS)
    p.hist_cap = cfg->num_steps + 2 + 600;           // room for sc_selfplay_set_position prefixes
    p.tpos_cap = cfg->rollout_num + 2;
    p.trace_cap = cfg->trace_capacity > 0 ? std::min(cfg->n_games, std::max(cfg->trace_capacity, 2 * cfg->n_slots)) : cfg->n_games;
    p.total_games = cfg->n_games;
    const size_t G = (size_t)cfg->n_slots, NC = (size_t)p.node_cap;
    int rc = 0;
    rc |= sp_alloc(sp, &p.ctl, G);
    rc |= sp_alloc(sp, &p.hist, G * p.hist_cap, false);
    rc |= sp_alloc(sp, &p.tpos, G * p.tpos_cap, false);
    rc |= sp_alloc(sp, &p.path, G * p.max_depth);
    rc |= sp_alloc(sp, &p.N, G * NC, false);
    rc |= sp_alloc(sp, &p.W, G * NC, false);
    rc |= sp_alloc(sp, &p.P, G * NC, false);
    rc |= sp_alloc(sp, &p.U, G * NC, false);
    rc |= sp_alloc(sp, &p.MV, G * NC, false);
    rc |= sp_alloc(sp, &p.H, G * NC, false);
    rc |= sp_alloc(sp, &p.boards, G * 7168);
    rc |= sp_alloc(sp, &p.meta, G * 8);
    rc |= sp_alloc(sp, &p.legal_mv, G * 224);
    rc |= sp_alloc(sp, &p.legal_idx, G * 224);
    rc |= sp_alloc(sp, &p.n_legal, G);
    rc |= sp_alloc(sp, &p.prior, G * 224);
    rc |= sp_alloc(sp, &p.value, G);
    rc |= sp_alloc(sp, &p.noise, G * 224);
    const size_t T = (size_t)p.trace_cap, S = (size_t)p.num_steps;
    rc |= sp_alloc(sp, &p.thdr, T);
    rc |= sp_alloc(sp, &p.t_move, T * S);
    rc |= sp_alloc(sp, &p.t_q, T * S);
    rc |= sp_alloc(sp, &p.t_nchild, T * S);
    rc |= sp_alloc(sp, &p.t_cmove, T * S * 224, false);
    rc |= sp_alloc(sp, &p.t_cn, T * S * 224, false);
    rc |= sp_alloc(sp, &p.t_cq, T * S * 224, false);
    rc |= sp_alloc(sp, &p.t_cu, T * S * 224, false);
    rc |= sp_alloc(sp, &p.cnt, 1);
    rc |= sp_alloc(sp, &p.slot_cnt, (size_t)cfg->n_slots * 2);
    if (rc) {
        std::string keep = g_err;
        sc_selfplay_destroy(sp);
        return fail("self-play allocation failed: " + keep, -2);
    }
    if (e) {
        rc = engine_reserve(e, cfg->n_slots);
        if (rc) {
            sc_selfplay_destroy(sp);
            return rc;
        }
    }
    if (e && cfg->evaluator == SC_EVAL_NET) {
        rc |= sp_alloc(sp, &sp->d_hval, G * 64 * 256);
        rc |= sp_alloc(sp, &sp->d_vpart, (size_t)e->ksplit * G * 128);
        rc |= sp_alloc(sp, &sp->d_fc1_ctr, (G + 63) / 64 * 32);
        if (rc) {
            sc_selfplay_destroy(sp);
            return fail("self-play allocation failed", -2);
        }
        p.vf_fused = 1;
        p.vf_ksplit = e->ksplit;
        p.vpart = sp->d_vpart;
        p.vf_w = e->d_wf;
        p.vf_fc1b = (uint32_t)e->net.f_fc1b;
        p.vf_fc1m = (uint32_t)e->net.f_fc1m;
        p.vf_fc2w = (uint32_t)e->net.f_fc2w;
        p.vf_fc2b = (uint32_t)e->net.f_fc2b;
    }
    sp->reported.assign((size_t)p.trace_cap, 0);
    // (the narrow fp8 tower's workgroup is small enough -- 59 KB of LDS, 249 VGPRs -- for two fused workgroups per CU; the
    // runtime's occupancy answer is used, capped at 2: a third would leave CUs empty at 512 games.  The bf16 one is not.)
    sp->fused = e && cfg->evaluator == SC_EVAL_NET && !cfg->own_stream && cfg->n_slots <= e->n_cu * e->step_blocks_per_cu;
#ifdef SC_EXP
    if (getenv("SC_FUSED")) sp->fused = getenv("SC_FUSED")[0] != '0';   // experiment builds: A/B
#endif
    // One launch per step: value_head.ffn.0's 64-position tiles are computed by the step kernel's own workgroups (workgroup g:
    // tile (g / 64, K chunk g % 64), step_kernels.hip) -- when the slots fill whole blocks and the split is the kernel's 64.
    sp->fc1_in_step = sp->fused && cfg->n_slots % 64 == 0 && e->ksplit == 64;
#ifdef SC_FC1_IN_STEP_OFF   // A/B builds
    sp->fc1_in_step = false;
#endif
    if (sp->fc1_in_step) sp->fc1_in_step = fc1_stream_acquire(sp->device, sp->stream);
    // the zero-fills above ran on the NULL stream, which does not order against the (non-blocking) launch
    // stream: make them complete before the first kernel touches the buffers
    hipError_t he = hipDeviceSynchronize();
    if (he != hipSuccess) return bail(he, "hipDeviceSynchronize");
    scl::init_slots(p, sp->stream);
    if ((he = hipGetLastError()) != hipSuccess) return bail(he, "k_init_slots");
    if ((he = hipStreamSynchronize(sp->stream)) != hipSuccess) return bail(he, "k_init_slots");
    *out = sp;
    return 0;
}

void sc_selfplay_destroy(sc_selfplay* sp) {
    if (!sp) return;
    (void)hipSetDevice(sp->device);
    if (sp->stream) (void)hipStreamSynchronize(sp->stream);
    else (void)hipDeviceSynchronize();
    if (sp->fc1_in_step) fc1_stream_release(sp->device);
    for (void* a : sp->allocs) (void)hipFree(a);
    for (hipEvent_t ev : sp->ev) (void)hipEventDestroy(ev);
    if (sp->ev_begin) (void)hipEventDestroy(sp->ev_begin);
    if (sp->ev_end) (void)hipEventDestroy(sp->ev_end);
    if (sp->own_stream && sp->stream) (void)hipStreamDestroy(sp->stream);
    delete sp;
}

int sc_selfplay_enable_timing(sc_selfplay* sp, int stride) {
    if (!sp) return fail("null handle");
    HIPOK(hipSetDevice(sp->device));
    sp->timing_stride = stride > 0 ? stride : 0;
    sp->step_stride = stride < 0 ? -stride : 0;
    if (stride != 0 && sp->ev.empty()) {
        sp->ev.resize(2 * 4096);
        for (auto& ev : sp->ev) HIPOK(hipEventCreate(&ev));
        HIPOK(hipEventCreate(&sp->ev_begin));
        HIPOK(hipEventCreate(&sp->ev_end));
    }
    return 0;
}

int sc_selfplay_enqueue_sims(sc_selfplay* sp, int n) {
    if (!sp || n < 0) return fail("bad argument");
    if (sp->poisoned) return sp_refuse(sp);
    HIPOK(hipSetDevice(sp->device));
    hipStream_t s = sp->stream;
    const sc::SpParams& p = sp->p;
    const bool any_timing = sp->timing_stride > 0 || sp->step_stride > 0;
    if (any_timing && !sp->have_span) {
        HIPOK(hipEventRecord(sp->ev_begin, s));
        sp->have_span = true;
    }
    for (int i = 0; i < n; i++) {
        sc_engine* e = sp->engine;
        sc::SpParams q = p;
        if (sp->match) {
            // All games are at the same ply: simulation step t belongs to ply t / rollout.  The search kernel first
            // finishes step t-1 (value tail: the weights of THAT step's player), then selects the leaf that this
            // step's player evaluates.
            const int64_t t = sp->sim_steps_enqueued + i;
            const int cur = (int)((t / p.rollout) & 1);
            e = sp->player[cur];
            q.synth_salt = sp->salt[cur];
            match_tail_params(sp, q, t - 1);
        }
        const bool timed = p.evaluator == SC_EVAL_NET && sp->timing_stride > 0 && (sp->nn_launches % sp->timing_stride) == 0;
        // whole-step sampling (the dominant kernel of production is the step launch itself): the launch form is not changed
        const bool step_timed = p.evaluator == SC_EVAL_NET && sp->step_stride > 0 && (sp->nn_launches % sp->step_stride) == 0;
        int step_slot = 0;
        if (step_timed) {
            step_slot = sp->ev_next;
            sp->ev_next = (sp->ev_next + 1) % 4096;
            HIPOK(hipEventRecord(sp->ev[2 * step_slot], s));
        }
        auto step_timed_end = [&]() -> hipError_t {
            if (!step_timed) return hipSuccess;
            sp->ev_recorded++;
            return hipEventRecord(sp->ev[2 * step_slot + 1], s);
        };
        if (p.evaluator == SC_EVAL_NET && !timed && sp->fused) {
            // one launch: the game's search wave (finish the previous simulation, select + encode the next leaf) is wave 0
            // of its tower workgroup (step_kernels.hip); bit-identical to the two launches below
            scnn::TowerArgs t{};
            t.net = e->net;
            t.n_pos = p.n_slots;
            t.boards = p.boards;
            t.meta = p.meta;
            t.meta_stride = 8;
            t.legal_idx = p.legal_idx;
            t.n_legal = p.n_legal;
            t.prior = p.prior;
            t.logp = nullptr;
            t.hval = sp->d_hval;
            t.dbg = nullptr;
            t.dbg_stage = -1;
            if (sp->fc1_in_step) {
                t.fc1_arrive = sp->d_fc1_ctr;
                t.fc1_target = 64u * ++sp->fc1_launches + sp->fc1_target_skew;   // every workgroup of a block arrives once per launch (wraps with the counter)
                t.vpart = sp->d_vpart;
                t.fc1_acquire = e->step_blocks_per_cu > 1;
                scl::step(t, q, 1, s);
            } else {
                scl::step(t, q, 1, s);
                scnn::Fc1Args f{};
                f.net = e->net;
                f.n_pos = p.n_slots;
                f.ksplit = e->ksplit;
                f.hval = sp->d_hval;
                f.vpart = sp->d_vpart;
                scl::value_fc1(f, s);
            }
            HIPOK(step_timed_end());
            sp->nn_launches++;
            continue;
        }
        // finish the previous simulation (expand/backward/ply transition) and select + encode the next leaf
        scl::mcts(q, 1, 1, s);
        if (p.evaluator != SC_EVAL_NET) {
            scl::synth_eval(q, s);
        } else {
            int slot = 0;
            if (timed) {
                slot = sp->ev_next;
                sp->ev_next = (sp->ev_next + 1) % 4096;
                HIPOK(hipEventRecord(sp->ev[2 * slot], s));
            }
            // tower only inside the timed bracket: it is the dominant kernel priced by the roofline
            scnn::TowerArgs t{};
            t.net = e->net;
            t.n_pos = p.n_slots;
            t.boards = p.boards;
            t.meta = p.meta;
            t.meta_stride = 8;
            t.legal_idx = p.legal_idx;
            t.n_legal = p.n_legal;
            t.prior = p.prior;
            t.logp = nullptr;
            t.hval = sp->d_hval;
            t.dbg = nullptr;
            t.dbg_stage = -1;
            scl::tower(t, s);
            if (timed) {
                HIPOK(hipEventRecord(sp->ev[2 * slot + 1], s));
                sp->ev_recorded++;
            }
            scnn::Fc1Args f{};
            f.net = e->net;
            f.n_pos = p.n_slots;
            f.ksplit = e->ksplit;
            f.hval = sp->d_hval;
            f.vpart = sp->d_vpart;
            scl::value_fc1(f, s);   // the tail of the value head is fused into the next k_mcts launch
            HIPOK(step_timed_end());
            sp->nn_launches++;
        }
    }
    if (n > 0) sp->pending_final = true;  // the last simulation is completed lazily (sp_flush) before any host read
    sp->sim_steps_enqueued += n;
    if (any_timing) HIPOK(hipEventRecord(sp->ev_end, s));
    HIPOK(hipGetLastError());
    return 0;
}

int sc_selfplay_set_players(sc_selfplay* sp, sc_engine* white, sc_engine* black, uint64_t salt_white, uint64_t salt_black) {
    if (!sp) return fail("null handle");
    if (sp->sim_steps_enqueued != 0) return fail("set_players must precede the first enqueue");
    if (sp->cfg.n_games != sp->cfg.n_slots) return fail("match play needs n_games == n_slots (lockstep plies, no slot recycling)");
    if (sp->cfg.rollout_factor > 0.f) return fail("match play needs a fixed rollout (lockstep plies)");
    if (sp->cfg.evaluator == SC_EVAL_NET) {
        if (!white || !black) return fail("match play with SC_EVAL_NET needs two engines");
        if (white->device != sp->device || black->device != sp->device) return fail("both engines must live on the handle's device");
        if (white->ksplit != sp->engine->ksplit || black->ksplit != sp->engine->ksplit) return fail("engines differ in split-K");
        int rc = engine_reserve(white, sp->cfg.n_slots);
        if (!rc) rc = engine_reserve(black, sp->cfg.n_slots);
        if (rc) return rc;
    }
    sp->match = true;
    sp->player[0] = white;
    sp->player[1] = black;
    sp->salt[0] = salt_white;
    sp->salt[1] = salt_black;
    return 0;
}

int sc_selfplay_enqueue_interleaved(sc_selfplay** handles, int n_handles, int n) {
    if (!handles || n_handles <= 0 || n < 0) return fail("bad argument");
    for (int i = 0; i < n; i++)
        for (int h = 0; h < n_handles; h++) {
            int rc = sc_selfplay_enqueue_sims(handles[h], 1);
            if (rc) return rc;
        }
    return 0;
}

int sc_selfplay_synchronize(sc_selfplay* sp) {
    if (!sp) return fail("null handle");
    HIPOK(hipSetDevice(sp->device));
    sp_flush(sp);
    HIPOK(hipStreamSynchronize(sp->stream));
    int rc = sp_latch(sp);
    if (rc) return rc;
    return sp->poisoned ? sp_refuse(sp) : 0;
}

int sc_selfplay_get_stats(sc_selfplay* sp, sc_selfplay_stats* out) {
    if (!sp || !out) return fail("bad argument");
    HIPOK(hipSetDevice(sp->device));
    sp_flush(sp);
    HIPOK(hipStreamSynchronize(sp->stream));
    sc::Counters c;
    HIPOK(hipMemcpy(&c, sp->p.cnt, sizeof c, hipMemcpyDeviceToHost));
    std::vector<sc::GameCtl> ctl((size_t)sp->p.n_slots);
    HIPOK(hipMemcpy(ctl.data(), sp->p.ctl, ctl.size() * sizeof(sc::GameCtl), hipMemcpyDeviceToHost));
    int active = 0;
    for (auto& g : ctl) active += g.status == sc::ST_ACTIVE || g.status == sc::ST_PENDING;
    std::vector<unsigned long long> sc((size_t)sp->p.n_slots * 2);
    HIPOK(hipMemcpy(sc.data(), sp->p.slot_cnt, sc.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long sims = 0, evals = 0;
    for (int g = 0; g < sp->p.n_slots; g++) {
        sims += sc[(size_t)g * 2];
        evals += sc[(size_t)g * 2 + 1];
    }
    out->sims_done = (int64_t)sims;
    out->nn_evals = (int64_t)evals;
    out->games_finished = c.games_finished;
    out->games_active = active;
    out->error_flags = c.err;
    out->plies_done = (int32_t)c.plies_done;
    return sp_latch(sp);   // (the statistics stay readable on a poisoned handle: that is how the host learns the flags)
}

int sc_selfplay_run(sc_selfplay* sp, int64_t max_sim_steps) {
    if (!sp) return fail("null handle");
    if (sp->p.trace_hold && sp->p.trace_cap < sp->p.total_games)
        return fail("trace_hold with a ring smaller than n_games: drive the handle with sc_selfplay_enqueue_sims + sc_selfplay_poll");
    int64_t done = 0;
    for (;;) {
        int chunk = sp->p.rollout;
        if (max_sim_steps > 0 && done + chunk > max_sim_steps) chunk = (int)(max_sim_steps - done);
        if (chunk <= 0) break;
        int rc = sc_selfplay_enqueue_sims(sp, chunk);
        if (rc) return rc;
        done += chunk;
        sc_selfplay_stats st;
        rc = sc_selfplay_get_stats(sp, &st);
        if (rc) return rc;
        if (sp->poisoned) return sp_refuse(sp);
        if (st.games_active == 0) break;
    }
    return 0;
}

int sc_selfplay_launches_per_step(const sc_selfplay* sp) {
    if (!sp || sp->p.evaluator != SC_EVAL_NET) return 0;
    return sp->fc1_in_step ? 1 : sp->fused ? 2 : 3;
}

int sc_selfplay_timing(sc_selfplay* sp, int reset, float* ms_total, float* ms_nn, int64_t* nn_launches) {
    if (!sp) return fail("null handle");
    HIPOK(hipSetDevice(sp->device));
    sp_flush(sp);
    HIPOK(hipStreamSynchronize(sp->stream));
    float tot = 0.f, nn = 0.f;
    int64_t cnt = std::min<int64_t>(sp->ev_recorded, 4096);
    if (sp->timing_stride > 0 || sp->step_stride > 0) {
        if (sp->have_span) HIPOK(hipEventElapsedTime(&tot, sp->ev_begin, sp->ev_end));
        for (int64_t k = 0; k < cnt; k++) {
            int slot = (int)(((int64_t)sp->ev_next - 1 - k + 4096 * 2) % 4096);
            float ms = 0.f;
            HIPOK(hipEventElapsedTime(&ms, sp->ev[2 * slot], sp->ev[2 * slot + 1]));
            nn += ms;
        }
    }
    if (ms_total) *ms_total = tot;
    if (ms_nn) *ms_nn = nn;          // sum over the `cnt` sampled tower launches
    if (nn_launches) *nn_launches = cnt;
    if (reset) {
        sp->ev_recorded = 0;
        sp->ev_next = 0;
        sp->have_span = false;
        sp->nn_launches = 0;
    }
    return 0;
}

int sc_selfplay_get_trace(sc_selfplay* sp, int game, sc_trace_info* info, uint16_t* step_move, float* step_q,
                          int32_t* child_off, uint16_t* child_move, int32_t* child_n, float* child_q, float* child_uct) {
    if (!sp || !info || game < 0 || game >= sp->p.total_games) return fail("bad argument");
    const uint64_t want_id = sp->p.first_game_id + (uint64_t)game;
    game %= sp->p.trace_cap;
    HIPOK(hipSetDevice(sp->device));
    // A row that sc_selfplay_poll has reported and holds (trace_hold) is final and no kernel writes it until the next poll
    // releases it: it is read without waiting for the stream, so a consumer can enqueue the next simulation steps first and
    // fetch / write the finished traces while they run (the copies below are on the NULL stream; the launch stream is
    // non-blocking).  Anything else is read from an idle stream.
    const bool held = sp->p.trace_hold && sp->reported[(size_t)game] == want_id + 1 &&
                      std::find(sp->to_release.begin(), sp->to_release.end(), game) != sp->to_release.end();
    if (sp->poisoned) return sp_refuse(sp);
    if (!held) {
        sp_flush(sp);
        HIPOK(hipStreamSynchronize(sp->stream));
        int lrc = sp_latch(sp);
        if (lrc) return lrc;
        if (sp->poisoned) return sp_refuse(sp);
    }
    const sc::SpParams& p = sp->p;
    sc::TraceHdr h;
    HIPOK(hipMemcpy(&h, p.thdr + game, sizeof h, hipMemcpyDeviceToHost));
    // the ring row may belong to an earlier game (this one has not started), to a later one (overwritten) or be released
    if (h.state == sc::TR_FREE) return fail(sp->reported[(size_t)game] > want_id ? "trace released or overwritten" : "game not finished",
                                            sp->reported[(size_t)game] > want_id ? 2 : 1);
    if (h.game_id < want_id) return fail("game not finished", 1);
    if (h.game_id > want_id) return fail("trace overwritten by a later game (trace_capacity ring)", 2);
    if (h.state != sc::TR_DONE) return fail("game not finished", 1);
    const size_t S = (size_t)p.num_steps, base = (size_t)game * S;
    int ns = h.n_steps;
    std::vector<int32_t> nch((size_t)std::max(ns, 1));
    if (ns) HIPOK(hipMemcpy(nch.data(), p.t_nchild + base, (size_t)ns * 4, hipMemcpyDeviceToHost));
    int total = 0;
    for (int i = 0; i < ns; i++) total += nch[(size_t)i];
    info->n_steps = ns;
    info->n_children_total = total;
    info->has_outcome = h.has_outcome;
    info->termination = h.termination;
    info->winner = h.winner;
    info->game_id = h.game_id;
    if (step_move && ns) HIPOK(hipMemcpy(step_move, p.t_move + base, (size_t)ns * 2, hipMemcpyDeviceToHost));
    if (step_q && ns) HIPOK(hipMemcpy(step_q, p.t_q + base, (size_t)ns * 4, hipMemcpyDeviceToHost));
    if (child_off) {
        int off = 0;
        for (int i = 0; i < ns; i++) {
            child_off[i] = off;
            off += nch[(size_t)i];
        }
        child_off[ns] = off;
    }
    if ((child_move || child_n || child_q || child_uct) && ns) {
        // one copy per array for the whole game (rows of 224 entries per ply), compacted on the host: a copy per ply and
        // array cost more than the games themselves when many short games stream out
        const size_t rows = (size_t)ns * 224, src = base * 224;
        std::vector<uint16_t> mv(child_move ? rows : 0);
        std::vector<int32_t> cn(child_n ? rows : 0);
        std::vector<float> cq(child_q ? rows : 0), cu(child_uct ? rows : 0);
        if (child_move) HIPOK(hipMemcpy(mv.data(), p.t_cmove + src, rows * 2, hipMemcpyDeviceToHost));
        if (child_n) HIPOK(hipMemcpy(cn.data(), p.t_cn + src, rows * 4, hipMemcpyDeviceToHost));
        if (child_q) HIPOK(hipMemcpy(cq.data(), p.t_cq + src, rows * 4, hipMemcpyDeviceToHost));
        if (child_uct) HIPOK(hipMemcpy(cu.data(), p.t_cu + src, rows * 4, hipMemcpyDeviceToHost));
        int off = 0;
        for (int i = 0; i < ns; i++) {
            const size_t n = (size_t)nch[(size_t)i], r0 = (size_t)i * 224;
            if (child_move) memcpy(child_move + off, mv.data() + r0, n * 2);
            if (child_n) memcpy(child_n + off, cn.data() + r0, n * 4);
            if (child_q) memcpy(child_q + off, cq.data() + r0, n * 4);
            if (child_uct) memcpy(child_uct + off, cu.data() + r0, n * 4);
            off += (int)n;
        }
    }
    return 0;
}

int sc_selfplay_poll(sc_selfplay* sp, int32_t* finished_games, int cap) {
    if (!sp || cap < 0 || (cap > 0 && !finished_games)) return fail("bad argument");
    HIPOK(hipSetDevice(sp->device));
    if (sp->poisoned) return sp_refuse(sp);
    sp_flush(sp);
    HIPOK(hipStreamSynchronize(sp->stream));
    {
        int lrc = sp_latch(sp);
        if (lrc) return lrc;
        if (sp->poisoned) return sp_refuse(sp);   // nothing of this batch is reported: the games that "finished" are not real
    }
    const sc::SpParams& p = sp->p;
    // rows handed out by the previous poll go back to the device (the stream is idle: no kernel reads them now)
    for (int row : sp->to_release) {
        const int32_t free_state = sc::TR_FREE;
        HIPOK(hipMemcpy(reinterpret_cast<char*>(p.thdr + row) + offsetof(sc::TraceHdr, state), &free_state, 4, hipMemcpyHostToDevice));
    }
    sp->to_release.clear();
    std::vector<sc::TraceHdr> hdr((size_t)p.trace_cap);
    HIPOK(hipMemcpy(hdr.data(), p.thdr, hdr.size() * sizeof(sc::TraceHdr), hipMemcpyDeviceToHost));
    // oldest games first: a consumer that writes trace{N}.json sees them in the order the reference's jobs would finish
    std::vector<std::pair<uint64_t, int>> fin;
    for (int row = 0; row < p.trace_cap; row++) {
        const sc::TraceHdr& h = hdr[(size_t)row];
        if (h.state == sc::TR_DONE && sp->reported[(size_t)row] != h.game_id + 1) fin.emplace_back(h.game_id, row);
    }
    std::sort(fin.begin(), fin.end());
    int n = 0;
    for (auto& f : fin) {
        if (n >= cap) break;
        finished_games[n++] = (int32_t)(f.first - p.first_game_id);
        sp->reported[(size_t)f.second] = f.first + 1;
        if (p.trace_hold) sp->to_release.push_back(f.second);
    }
    return n;
}

int sc_selfplay_debug_break_handoff(sc_selfplay* sp, int missing) {
    if (!sp) return fail("null handle");
    if (!sp->fc1_in_step) return 1;
    sp->fc1_target_skew += (uint32_t)missing;
    return 0;
}

int sc_debug_clear_handoff_failure(int device_id) {
    std::lock_guard<std::mutex> lk(g_fc1_mu);
    return g_fc1_failed.erase(device_id) ? 0 : 1;
}

int sc_debug_find_max(int device_id, const float* values, int n, int32_t* out2) {
    if (!values || !out2 || n < 1 || n > 256) return fail("bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device available: libsc_engine has no CPU fallback", -3);
    HIPOK(hipSetDevice(device_id));
    float* d_u = nullptr;
    int* d_o = nullptr;
    HIPOK(dalloc(&d_u, 256));
    HIPOK(dalloc(&d_o, 2));
    HIPOK(hipMemcpy(d_u, values, (size_t)n * 4, hipMemcpyHostToDevice));
    scl::debug_find_max(d_u, n, d_o, nullptr);
    hipError_t e1 = hipGetLastError(), e2 = hipMemcpy(out2, d_o, 8, hipMemcpyDeviceToHost);
    dfree({d_u, d_o});
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(hipGetErrorString(e1 != hipSuccess ? e1 : e2), -2);
    return 0;
}

int sc_trace_write_json(const char* path, const sc_trace_info* info, const uint16_t* step_move, const float* step_q,
                        const int32_t* child_off, const uint16_t* child_move, const int32_t* child_n, const float* child_q,
                        const float* child_uct) {
    if (!path || !info) return fail("bad argument");
    std::string js = sctrace::trace_to_json(info->n_steps, info->has_outcome, info->termination, info->winner, step_move, step_q,
                                            child_off, child_move, child_n, child_q, child_uct);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(std::string("cannot open ") + path);
    size_t w = fwrite(js.data(), 1, js.size(), f);
    fclose(f);
    if (w != js.size()) return fail("short write");
    return 0;
}

int sc_selfplay_write_trace_json(sc_selfplay* sp, int game, const char* path) {
    sc_trace_info info;
    int rc = sc_selfplay_get_trace(sp, game, &info, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    std::vector<uint16_t> sm((size_t)info.n_steps + 1), cm((size_t)info.n_children_total + 1);
    std::vector<float> sq((size_t)info.n_steps + 1), cq((size_t)info.n_children_total + 1), cu((size_t)info.n_children_total + 1);
    std::vector<int32_t> co((size_t)info.n_steps + 2), cn((size_t)info.n_children_total + 1);
    rc = sc_selfplay_get_trace(sp, game, &info, sm.data(), sq.data(), co.data(), cm.data(), cn.data(), cq.data(), cu.data());
    if (rc) return rc;
    return sc_trace_write_json(path, &info, sm.data(), sq.data(), co.data(), cm.data(), cn.data(), cq.data(), cu.data());
}

int sc_selfplay_get_tree(sc_selfplay* sp, int slot, int cap, int32_t* n, float* q, float* uct, float* prior, uint16_t* move,
                         int32_t* first_child, int32_t* n_child) {
    if (!sp || slot < 0 || slot >= sp->p.n_slots) return fail("bad argument");
    HIPOK(hipSetDevice(sp->device));
    sp_flush(sp);
    HIPOK(hipStreamSynchronize(sp->stream));
    const sc::SpParams& p = sp->p;
    sc::GameCtl c;
    HIPOK(hipMemcpy(&c, p.ctl + slot, sizeof c, hipMemcpyDeviceToHost));
    int nn = std::min(c.n_nodes, cap);
    size_t nb = (size_t)slot * p.node_cap;
    if (nn > 0) {
        if (n) HIPOK(hipMemcpy(n, p.N + nb, (size_t)nn * 4, hipMemcpyDeviceToHost));
        if (q) HIPOK(hipMemcpy(q, p.W + nb, (size_t)nn * 4, hipMemcpyDeviceToHost));
        if (uct) HIPOK(hipMemcpy(uct, p.U + nb, (size_t)nn * 4, hipMemcpyDeviceToHost));
        if (prior) HIPOK(hipMemcpy(prior, p.P + nb, (size_t)nn * 4, hipMemcpyDeviceToHost));
        if (move) HIPOK(hipMemcpy(move, p.MV + nb, (size_t)nn * 2, hipMemcpyDeviceToHost));
        if (first_child || n_child) {
            std::vector<sc::NodeHdr> t((size_t)nn);
            HIPOK(hipMemcpy(t.data(), p.H + nb, (size_t)nn * sizeof(sc::NodeHdr), hipMemcpyDeviceToHost));
            for (int i = 0; i < nn; i++) {
                if (first_child) first_child[i] = t[(size_t)i].fc;
                if (n_child) n_child[i] = t[(size_t)i].nc;
            }
        }
    }
    return c.n_nodes;
}

int sc_selfplay_get_slot(sc_selfplay* sp, int slot, int32_t* ply, int32_t* sim, int32_t* status, uint64_t* game_id,
                         int32_t* last_path, int32_t* last_path_len) {
    if (!sp || slot < 0 || slot >= sp->p.n_slots) return fail("bad argument");
    HIPOK(hipSetDevice(sp->device));
    sp_flush(sp);
    HIPOK(hipStreamSynchronize(sp->stream));
    sc::GameCtl c;
    HIPOK(hipMemcpy(&c, sp->p.ctl + slot, sizeof c, hipMemcpyDeviceToHost));
    if (ply) *ply = c.ply;
    if (sim) *sim = c.sim;
    if (status) *status = c.status;
    if (game_id) *game_id = c.game_id;
    if (last_path_len) *last_path_len = c.path_len;
    if (last_path && c.path_len > 0)
        HIPOK(hipMemcpy(last_path, sp->p.path + (size_t)slot * sp->p.max_depth, (size_t)std::min(c.path_len, 1024) * 4,
                        hipMemcpyDeviceToHost));
    return 0;
}

int sc_selfplay_set_noise(sc_selfplay* sp, int slot, const float* noise, int n) {
    if (!sp || slot < 0 || slot >= sp->p.n_slots || n < 0 || n > 224) return fail("bad argument");
    HIPOK(hipSetDevice(sp->device));
    sp_flush(sp);
    HIPOK(hipStreamSynchronize(sp->stream));
    HIPOK(hipMemcpy(sp->p.noise + (size_t)slot * 224, noise, (size_t)n * 4, hipMemcpyHostToDevice));
    return 0;
}
int sc_selfplay_get_noise(sc_selfplay* sp, int slot, float* noise, int cap) {
    if (!sp || slot < 0 || slot >= sp->p.n_slots) return fail("bad argument");
    HIPOK(hipSetDevice(sp->device));
    sp_flush(sp);
    HIPOK(hipStreamSynchronize(sp->stream));
    HIPOK(hipMemcpy(noise, sp->p.noise + (size_t)slot * 224, (size_t)std::min(cap, 224) * 4, hipMemcpyDeviceToHost));
    return 0;
}

int sc_selfplay_set_position(sc_selfplay* sp, int slot, const uint16_t* moves, int n_moves) {
    if (!sp || slot < 0 || slot >= sp->p.n_slots || n_moves < 0 || n_moves > 590) return fail("bad argument");
    HIPOK(hipSetDevice(sp->device));
    sp_flush(sp);
    {
        HIPOK(hipStreamSynchronize(sp->stream));
        sc::GameCtl c;
        HIPOK(hipMemcpy(&c, sp->p.ctl + slot, sizeof c, hipMemcpyDeviceToHost));
        if (c.status == sc::ST_PENDING) return fail("slot is waiting for a trace-ring row: no game to reposition");
    }
    uint16_t* d_moves = nullptr;
    HIPOK(dalloc(&d_moves, (size_t)n_moves));
    if (n_moves) HIPOK(hipMemcpy(d_moves, moves, (size_t)n_moves * 2, hipMemcpyHostToDevice));
    scl::set_position(sp->p, slot, d_moves, n_moves, sp->stream);
    HIPOK(hipGetLastError());
    HIPOK(hipStreamSynchronize(sp->stream));
    dfree({d_moves});
    return 0;
}

int sc_selfplay_set_search(sc_selfplay* sp, float cpuct, float epsilon, int with_noise) {
    if (!sp) return fail("null handle");
    if (!(cpuct >= 0.f) || !(epsilon >= 0.f && epsilon <= 1.f)) return fail("bad search parameters");
    // kernel parameters travel by value with every launch: the change applies to the launches enqueued after it
    sp->p.cpuct = cpuct;
    sp->p.epsilon = epsilon;
    sp->p.with_noise = with_noise ? 1 : 0;
    return 0;
}

// One search from a given position: the body of NNPlayer::bestmove (src/play.rs:241-252) / chess_play_mcts
// (src/lib.rs:233-247) as a single call; see include/sc_engine.h
int sc_search(sc_engine* e, const uint16_t* moves, int n_moves, int rollout, float cpuct, int with_noise, uint64_t seed,
              int cap, uint16_t* child_move, int32_t* child_n, float* child_q, float* child_prior, float* root_q) {
    if (!e || rollout < 1 || rollout >= 60000 || cap < 0) return fail("bad argument");
    // one cached handle per engine, rebuilt only when a call asks for more simulations than its node pools hold; every call starts
    // from a fresh one-node tree (sc_selfplay_set_position) with its own options and seed, so the result is that of a new handle
    int rc = 0;
    if (e->search_sp && (rollout + 1 > e->search_rollout_cap || e->search_sp->poisoned)) {
        sc_selfplay_destroy(e->search_sp);
        e->search_sp = nullptr;
    }
    if (!e->search_sp) {
        sc_selfplay_config c{};
        c.n_slots = 1;
        c.n_games = 1;
        c.rollout_num = std::max(rollout + 1, 512);   // sizes the node pool; more than any call runs, so no ply transition happens
        c.num_steps = 4000;
        c.cpuct = cpuct;
        c.epsilon = 0.15f;         // mcts::mcts(.., 0.15, noise) at both call sites
        c.with_noise = with_noise ? 1 : 0;
        c.outcome_gate = 1 << 30;
        c.evaluator = SC_EVAL_NET;
        c.seed = seed;
        rc = sc_selfplay_create(e, e->device, &c, &e->search_sp);
        if (rc) {
            e->search_sp = nullptr;
            return rc;
        }
        e->search_rollout_cap = c.rollout_num;
    }
    sc_selfplay* sp = e->search_sp;
    HIPOK(hipSetDevice(e->device));
    HIPOK(hipStreamSynchronize(sp->stream));
    sp->p.seed = seed;
    sp->cfg.seed = seed;
    rc = sc_selfplay_set_search(sp, cpuct, 0.15f, with_noise);
    {   // an earlier call's error flags are not this call's
        const int32_t zero = 0;
        HIPOK(hipMemcpy(reinterpret_cast<char*>(sp->p.cnt) + offsetof(sc::Counters, err), &zero, 4, hipMemcpyHostToDevice));
    }
    if (rc) return rc;
    rc = sc_selfplay_set_position(sp, 0, moves, n_moves);
    if (!rc) rc = sc_selfplay_enqueue_sims(sp, rollout);
    int n_children = 0;
    if (!rc) {
        std::vector<int32_t> n(1 + 224), fc(1 + 224), nc(1 + 224);
        std::vector<float> q(1 + 224), pr(1 + 224);
        std::vector<uint16_t> mv(1 + 224);
        int nn = sc_selfplay_get_tree(sp, 0, 1 + 224, n.data(), q.data(), nullptr, pr.data(), mv.data(), fc.data(), nc.data());
        if (nn < 0) {
            rc = nn;
        } else {
            sc_selfplay_stats st{};
            rc = sc_selfplay_get_stats(sp, &st);
            if (!rc && st.error_flags) rc = fail("search error flags set (non-finite PUCT value or pool overflow)", -4);
            if (root_q) *root_q = nn > 0 ? q[0] : 0.f;
            n_children = nn > 0 && fc[0] == 1 ? nc[0] : 0;   // the root's children are nodes 1..nc
            for (int i = 0; i < n_children && i < cap; i++) {
                if (child_move) child_move[i] = mv[(size_t)1 + i];
                if (child_n) child_n[i] = n[(size_t)1 + i];
                if (child_q) child_q[i] = q[(size_t)1 + i];
                if (child_prior) child_prior[i] = pr[(size_t)1 + i];
            }
        }
    }
    return rc ? rc : n_children;
}

int sc_move_uci(uint16_t move, char* buf8) { return sctrace::move_uci(move, buf8); }

// libsmartchess.chess_encode_move(turn, move) (reference src/lib.rs:37-44): the 4672-wide action index of a move for
// the side to move (Black's moves are rotated first); -1 if the move has no index.  Pure integer host function.
int sc_move_index(uint16_t move, int white_to_move) { return sc::move_index((sc::move_t)move, white_to_move ? sc::WHITE : sc::BLACK); }

/* developer aid: cycle stamps of the last k_mcts launch, out[n_slots][8] */
int sc_selfplay_debug_cycles(sc_selfplay* sp, int enable, unsigned long long* out) {
    if (!sp) return fail("null handle");
    HIPOK(hipSetDevice(sp->device));
    HIPOK(hipStreamSynchronize(sp->stream));
    if (enable && !sp->p.dbg_cycles) {
        HIPOK(dalloc(&sp->p.dbg_cycles, (size_t)sp->p.n_slots * 32));
        sp->allocs.push_back(sp->p.dbg_cycles);
        HIPOK(hipMemset(sp->p.dbg_cycles, 0, (size_t)sp->p.n_slots * 256));
        HIPOK(hipDeviceSynchronize());
    }
    if (out && sp->p.dbg_cycles)
        HIPOK(hipMemcpy(out, sp->p.dbg_cycles, (size_t)sp->p.n_slots * 256, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
