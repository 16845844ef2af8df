// sc-selfplay -- launcher with the reference's `selfplay` command line (src/main.rs:25-60), running
// many games concurrently on one or more MI355X and writing one reference-format trace file per game
// (src/trace.rs:23-32), so that scripts/run_batch / scripts/iterate / scripts/train.py -t '<dir>/*.json'
// keep working unchanged.
//
//   reference:  cargo r --release --bin selfplay -- -d cuda --rollout-num 300 -n 200 --temperature 0 \
//                   --cpuct 2 -c <ckpt>.pt -t <PREFIX>/trace<JOB>.json              (scripts/run_batch:17)
//   here:       sc-selfplay -d cuda --rollout-num 300 -n 200 --temperature 0 --cpuct 2 -c <weights>.scw \
//                   -t <PREFIX>/trace{}.json --games 2048 --concurrency 256 --gpus 8
//
// Same flags, same defaults; `-t` may contain `{}` (replaced by the 1-based game number, run_batch's
// JOB_ID) -- without it and with --games 1 the file name is used verbatim, exactly like the reference.
// Extra flags (no reference counterpart): --games, --concurrency, --groups, --gpus, --seed, --blocks/--channels
// (random-init network when no checkpoint is given), --first-game, --fp8 (e4m3 convs; BASELINE configs[4]).
// One host thread per GPU; games are sharded statically over GPUs, no collective.
// Traces stream out: each game's file is written when the game ends (the reference saves at the end of its one game,
// src/main.rs:235-238), from a bounded ring of traces on the device (sc_selfplay_poll), whatever --games is.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/sc_engine.h"

struct Args {
    std::string device = "cuda";
    float rollout_factor = -1.f;
    int rollout_num = -1;
    int num_steps = 100;
    std::string trace_file = "trace.json";
    std::string checkpoint = "__no_checkpoint__";
    std::string endpoint = "__no_endpoint__";
    float temperature = 0.0f;
    float cpuct = 1.0f;
    int temperature_switch = 30;
    float epsilon = 0.15f;
    // extensions
    int games = 1, concurrency = 256, gpus = 1, blocks = 10, channels = 256;
    bool fp8 = false;  // --fp8: run the convs in e4m3 (an SCW2 checkpoint selects it by itself)
    int groups = 1;  // handles per GPU on separate HIP streams: one group's tree work hides under another's network launch
                     // (e.g. --concurrency 512 --groups 2: +28 % simulations/s on one MI355X)
    unsigned long long seed = 0xC0FFEEULL, first_game = 0;
};

static void usage() {
    fprintf(stderr,
            "usage: sc-selfplay [-d cuda] [-r|--rollout-factor F | --rollout-num N] [-n|--num-steps 100] [-t|--trace-file trace.json]\n"
            "                   [-c|--checkpoint weights.scw] [--temperature 0] [--cpuct 1] [--temperature-switch 30] [--epsilon 0.15]\n"
            "                   [--games 1] [--concurrency 256] [--groups 1] [--gpus 1] [--seed S] [--first-game K] [--blocks 10] [--channels 256] [--fp8]\n");
}

static bool parse(int argc, char** argv, Args& a) {
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i];
        auto val = [&](const char* name) -> const char* {
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", name);
                exit(2);
            }
            return argv[++i];
        };
        if (k == "-d" || k == "--device") a.device = val("device");
        else if (k == "-r" || k == "--rollout-factor") a.rollout_factor = (float)atof(val("rollout-factor"));
        else if (k == "--rollout-num") a.rollout_num = atoi(val("rollout-num"));
        else if (k == "-n" || k == "--num-steps") a.num_steps = atoi(val("num-steps"));
        else if (k == "-t" || k == "--trace-file") a.trace_file = val("trace-file");
        else if (k == "-c" || k == "--checkpoint") a.checkpoint = val("checkpoint");
        else if (k == "--endpoint") a.endpoint = val("endpoint");
        else if (k == "--temperature") a.temperature = (float)atof(val("temperature"));
        else if (k == "--cpuct") a.cpuct = (float)atof(val("cpuct"));
        else if (k == "--temperature-switch") a.temperature_switch = atoi(val("temperature-switch"));
        else if (k == "--epsilon") a.epsilon = (float)atof(val("epsilon"));
        else if (k == "--games") a.games = atoi(val("games"));
        else if (k == "--concurrency") a.concurrency = atoi(val("concurrency"));
        else if (k == "--gpus") a.gpus = atoi(val("gpus"));
        else if (k == "--groups") a.groups = atoi(val("groups"));
        else if (k == "--seed") a.seed = strtoull(val("seed"), nullptr, 0);
        else if (k == "--first-game") a.first_game = strtoull(val("first-game"), nullptr, 0);
        else if (k == "--blocks") a.blocks = atoi(val("blocks"));
        else if (k == "--channels") a.channels = atoi(val("channels"));
        else if (k == "--fp8") a.fp8 = true;
        else if (k == "-h" || k == "--help") { usage(); exit(0); }
        else { fprintf(stderr, "unknown argument %s\n", k.c_str()); usage(); return false; }
    }
    return true;
}

static std::string trace_name(const Args& a, unsigned long long game_number) {
    std::string t = a.trace_file;
    size_t p = t.find("{}");
    if (p != std::string::npos) return t.substr(0, p) + std::to_string(game_number) + t.substr(p + 2);
    if (a.games == 1) return t;
    size_t dot = t.rfind('.');
    if (dot == std::string::npos) return t + std::to_string(game_number);
    return t.substr(0, dot) + std::to_string(game_number) + t.substr(dot);
}

static int run_gpu(const Args& a, int gpu, int first, int count, int* finished_with_outcome) {
    sc_engine* eng = nullptr;
    sc_net_config nc{a.blocks, a.channels, a.seed, a.fp8 ? SC_PREC_FP8 : SC_PREC_BF16, 0};
    const char* w = a.checkpoint == "__no_checkpoint__" ? nullptr : a.checkpoint.c_str();
    if (sc_engine_create(&nc, w, gpu, &eng)) {
        fprintf(stderr, "gpu %d: %s\n", gpu, sc_last_error());
        return 1;
    }
    const int K = a.groups < 1 ? 1 : (a.groups > count ? count : a.groups);
    std::vector<sc_selfplay*> sps;
    int rc = 0, off = 0, total_slots = 0;
    for (int k = 0; k < K && !rc; k++) {
        const int cnt = count / K + (k < count % K ? 1 : 0);
        sc_selfplay_config c{};
        const int slots = a.concurrency / K > 0 ? a.concurrency / K : 1;
        c.n_slots = cnt < slots ? cnt : slots;
        c.n_games = cnt;
        // main.rs:175-180: --rollout-num N, or --rollout-factor F = min(300, n_legal * F) per ply (chosen on the device at
        // the first simulation of each ply), or 300
        c.rollout_num = a.rollout_num > 0 ? a.rollout_num : 300;
        c.rollout_factor = a.rollout_factor > 0 ? a.rollout_factor : 0.f;
        c.num_steps = a.num_steps;
        c.cpuct = a.cpuct;
        c.temperature = a.temperature;
        c.temperature_switch = a.temperature_switch;
        c.epsilon = a.epsilon;
        c.with_noise = 1;      // main.rs:195
        c.outcome_gate = 100;  // main.rs:223
        c.evaluator = SC_EVAL_NET;
        c.seed = a.seed;
        c.first_game_id = a.first_game + (unsigned long long)(first + off);
        c.own_stream = K > 1;
        // bounded trace ring, drained as games end; a finished trace is held until its file is written
        c.trace_capacity = 2 * c.n_slots + 64;
        c.trace_hold = 1;
        sc_selfplay* sp = nullptr;
        if (sc_selfplay_create(eng, gpu, &c, &sp)) {
            fprintf(stderr, "gpu %d: %s\n", gpu, sc_last_error());
            rc = 1;
            break;
        }
        sps.push_back(sp);
        total_slots += c.n_slots;
        off += cnt;
    }
    // interleave the groups simulation step by simulation step; after every chunk (one ply's worth of steps) the games that
    // have ended are collected (sc_selfplay_poll: their trace rows stay untouched until the next poll), the NEXT chunk is
    // enqueued, and only then are their traces fetched and written -- the GPU searches while the host formats JSON
    int with_outcome = 0, finished = 0, errs = 0, written = 0;
    long long sims = 0, evals = 0, plies = 0;
    // rates: over the whole run (engine creation excluded), and over the chunks during which every slot had a game (the ragged
    // tail -- the last games finishing beside empty slots -- is what a finite job pays, not what the pipeline sustains)
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    auto t_prev = t_start;
    long long sims_prev = 0, full_sims = 0;
    double full_s = 0.0;

    const int chunk = a.rollout_num > 0 ? a.rollout_num : 300;
    std::vector<int32_t> fin(4096);
    std::vector<std::pair<size_t, int32_t>> todo;   // (group, game) reported by the polls of this round
    bool was_full = false;
    rc = sc_selfplay_enqueue_interleaved(sps.data(), (int)sps.size(), chunk);
    while (!rc) {
        int active = 0;
        long long sims_now = 0;
        bool more = false;   // a poll filled its buffer: poll again before anything else is enqueued
        todo.clear();
        for (size_t k = 0; k < sps.size() && !rc; k++) {
            sc_selfplay* sp = sps[k];
            const int n = sc_selfplay_poll(sp, fin.data(), (int)fin.size());
            if (n < 0) {
                rc = 1;
                break;
            }
            for (int i = 0; i < n; i++) todo.emplace_back(k, fin[(size_t)i]);
            more = more || n == (int)fin.size();
            sc_selfplay_stats st{};
            if (sc_selfplay_get_stats(sp, &st)) rc = 1;
            active += st.games_active;
            sims_now += (long long)st.sims_done;
            if (st.error_flags) rc = 1;   // invalid games must not reach training: stop, exit non-zero
        }
        {
            const auto t_now = clk::now();
            if (was_full && active >= total_slots) {
                full_sims += sims_now - sims_prev;
                full_s += std::chrono::duration<double>(t_now - t_prev).count();
            }
            was_full = active >= total_slots;
            t_prev = t_now;
            sims_prev = sims_now;
        }
        if (rc) break;
        if (active > 0 && !more) rc = sc_selfplay_enqueue_interleaved(sps.data(), (int)sps.size(), chunk);
        for (size_t i = 0; i < todo.size() && !rc; i++) {
            sc_selfplay* sp = sps[todo[i].first];
            sc_trace_info info{};
            if (sc_selfplay_get_trace(sp, todo[i].second, &info, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) {
                rc = 1;
                break;
            }
            with_outcome += info.has_outcome;
            std::string path = trace_name(a, info.game_id + 1);
            if (sc_selfplay_write_trace_json(sp, todo[i].second, path.c_str())) rc = 1;
            else written++;
        }
        if (rc || (active == 0 && !more)) break;
    }
    if (rc) fprintf(stderr, "gpu %d: %s\n", gpu, sc_last_error());
    for (size_t k = 0; k < sps.size(); k++) {
        sc_selfplay_stats st{};
        sc_selfplay_get_stats(sps[k], &st);
        finished += st.games_finished;
        sims += (long long)st.sims_done;
        evals += (long long)st.nn_evals;
        plies += (long long)st.plies_done;
        errs |= st.error_flags;
    }
    const double wall = std::chrono::duration<double>(clk::now() - t_start).count();
    if (errs) {
        fprintf(stderr, "gpu %d: error_flags %d -- the games of this run are not to be trusted (include/sc_engine.h: sc_selfplay_stats)\n", gpu, errs);
        rc = 1;
    }
    if (!rc && written != count) {
        fprintf(stderr, "gpu %d: %d of %d traces written\n", gpu, written, count);
        rc = 1;
    }
    printf("gpu %d: games %d finished %d with-outcome %d simulations %lld error_flags %d\n", gpu, count, finished, with_outcome, sims, errs);
    // one JSON object per GPU (sims/s, games/s, occupancy -> JSON)
    printf("{\"gpu\": %d, \"games\": %d, \"finished\": %d, \"with_outcome\": %d, \"traces_written\": %d, \"simulations\": %lld, "
           "\"nn_evals\": %lld, \"plies\": %lld, \"wall_s\": %.3f, \"sims_per_s\": %.1f, \"games_per_s\": %.3f, "
           "\"full_occupancy_sims_per_s\": %.1f, \"full_occupancy_s\": %.3f, \"slots\": %d, \"groups\": %d, \"launches_per_step\": %d, "
           "\"error_flags\": %d, \"ok\": %s}\n",
           gpu, count, finished, with_outcome, written, sims, evals, plies, wall, wall > 0 ? sims / wall : 0.0, wall > 0 ? finished / wall : 0.0,
           full_s > 0 ? full_sims / full_s : 0.0, full_s, total_slots, (int)sps.size(), sps.empty() ? 0 : sc_selfplay_launches_per_step(sps[0]),
           errs, rc ? "false" : "true");
    *finished_with_outcome = with_outcome;
    for (sc_selfplay* sp : sps) sc_selfplay_destroy(sp);
    sc_engine_destroy(eng);
    return rc;
}

int main(int argc, char** argv) {
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);   // kernel arguments in device memory; before any HIP call
    Args a;
    if (!parse(argc, argv, a)) return 2;
    if (a.rollout_factor != -1.f && a.rollout_num > 0) {  // main.rs:74
        fprintf(stderr, "both --rollout-factor and --rollout-num are specified.\n");
        return 2;
    }
    if (a.device != "cuda") {
        fprintf(stderr, "device '%s' is not supported: this launcher has no CPU path (use the reference binary for -d cpu)\n",
                a.device.c_str());
        return 2;
    }
    if (a.rollout_factor != -1.f && !(a.rollout_factor > 0.f)) {
        fprintf(stderr, "--rollout-factor must be positive\n");
        return 2;
    }
    int ndev = sc_device_count();
    if (ndev <= 0) {
        fprintf(stderr, "no MI355X visible\n");
        return 1;
    }
    int gpus = a.gpus < ndev ? a.gpus : ndev;
    std::vector<std::thread> th;
    std::vector<int> rcs((size_t)gpus, 0), outc((size_t)gpus, 0);
    int base = 0;
    for (int g = 0; g < gpus; g++) {
        int count = a.games / gpus + (g < a.games % gpus ? 1 : 0);
        if (count == 0) continue;
        th.emplace_back([&, g, base, count]() { rcs[(size_t)g] = run_gpu(a, g, base, count, &outc[(size_t)g]); });
        base += count;
    }
    for (auto& t : th) t.join();
    int rc = 0;
    for (int r : rcs) rc |= r;
    return rc;
}
