// sc-play -- batched evaluation matches on MI355X with the flags of the reference's `play` binary
// (reference src/play.rs:36-84 Args, :318-343 play_loop, :440-472 main; driven by scripts/leader-board:4-14):
//
//     sc-play --white-device cuda --white-checkpoint new.scw --black-device cuda --black-checkpoint old.scw \
//             -o "replay/w_{}.json" --rollout=100 --temperature 0 --temperature-switch 0 --cpuct 1.5 --games 100
//
// One process plays `--games` games of the pairing concurrently on one GPU (the reference plays one game per process
// under GNU parallel); "{}" in -o is replaced by the 1-based game number.  Every game is the reference loop: the two
// players alternate by ply on one shared tree cursor, no Dirichlet noise, temperature 1 below --temperature-switch,
// at temperature 0 a random child among the most visited, outcome(claim_draw=True) after every ply, at most 200 plies,
// the outcome (or null) stored in the trace.  Stockfish opponents (--black-type stockfish) are out of scope.
// Extra flags (no reference counterpart): --games, --seed, --blocks / --channels / --white-seed / --black-seed (random
// networks when no checkpoint is given).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/sc_engine.h"

struct Args {
    std::string white_device, black_device = "<not-specified>", black_type = "stockfish";
    std::string black_checkpoint = "<not-specified>", white_checkpoint = "<not-specified>", output = "01.json";
    int rollout = 60, temperature_switch = 0, games = 1, blocks = 10, channels = 256;
    float temperature = 0.0f, cpuct = 0.0f;
    unsigned long long seed = 0xC0FFEEULL, white_seed = 1, black_seed = 2;
};

static void usage() {
    fprintf(stderr,
            "usage: sc-play --white-device cuda -w|--white-checkpoint W.scw [--black-device cuda] [--black-type nn]\n"
            "               [--black-checkpoint B.scw] [-r|--rollout 60] [--temperature 0] [--temperature-switch 0] [--cpuct 0]\n"
            "               [-o|--output 01.json] [--games 1] [--seed S] [--blocks 10] [--channels 256] [--white-seed 1] [--black-seed 2]\n");
}

static bool parse(int argc, char** argv, Args& a) {
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i], v;
        size_t eq = k.find('=');
        bool has = false;
        if (k.rfind("--", 0) == 0 && eq != std::string::npos) {   // --rollout=100 form (scripts/leader-board:46)
            v = k.substr(eq + 1);
            k = k.substr(0, eq);
            has = true;
        }
        auto val = [&]() -> const char* {
            if (has) return v.c_str();
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", k.c_str());
                exit(2);
            }
            return argv[++i];
        };
        if (k == "--white-device") a.white_device = val();
        else if (k == "--black-device") a.black_device = val();
        else if (k == "--black-type") a.black_type = val();
        else if (k == "--stockfish-bin" || k == "--stockfish-level") (void)val();
        else if (k == "--black-checkpoint") a.black_checkpoint = val();
        else if (k == "-w" || k == "--white-checkpoint") a.white_checkpoint = val();
        else if (k == "-r" || k == "--rollout") a.rollout = atoi(val());
        else if (k == "--temperature") a.temperature = (float)atof(val());
        else if (k == "--temperature-switch") a.temperature_switch = atoi(val());
        else if (k == "--cpuct") a.cpuct = (float)atof(val());
        else if (k == "-o" || k == "--output") a.output = val();
        else if (k == "--games") a.games = atoi(val());
        else if (k == "--seed") a.seed = strtoull(val(), nullptr, 0);
        else if (k == "--blocks") a.blocks = atoi(val());
        else if (k == "--channels") a.channels = atoi(val());
        else if (k == "--white-seed") a.white_seed = strtoull(val(), nullptr, 0);
        else if (k == "--black-seed") a.black_seed = strtoull(val(), nullptr, 0);
        else if (k == "-h" || k == "--help") { usage(); exit(0); }
        else { fprintf(stderr, "unknown argument %s\n", k.c_str()); usage(); return false; }
    }
    return true;
}

static std::string out_name(const Args& a, int game_number) {
    std::string t = a.output;
    size_t p = t.find("{}");
    if (p != std::string::npos) return t.substr(0, p) + std::to_string(game_number) + t.substr(p + 2);
    if (a.games == 1) return t;
    size_t dot = t.rfind('.');
    if (dot == std::string::npos) return t + std::to_string(game_number);
    return t.substr(0, dot) + "_" + std::to_string(game_number) + t.substr(dot);
}

int main(int argc, char** argv) {
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);   // kernel arguments in device memory; before any HIP call
    Args a;
    if (!parse(argc, argv, a)) return 2;
    if (a.white_device.empty()) {   // clap: required argument
        fprintf(stderr, "--white-device is required\n");
        return 2;
    }
    if (a.black_checkpoint != "<not-specified>") a.black_type = "nn";   // play.rs:405-408
    if (a.black_type != "nn" && a.black_type != "NN") {
        fprintf(stderr, "black-type '%s' is not supported: only network-vs-network matches are built (Stockfish/UCI is out of scope)\n",
                a.black_type.c_str());
        return 2;
    }
    if (a.white_device != "cuda") {
        fprintf(stderr, "device '%s' is not supported: this launcher has no CPU path\n", a.white_device.c_str());
        return 2;
    }
    if (a.games < 1 || a.rollout < 1) {
        fprintf(stderr, "--games and --rollout must be positive\n");
        return 2;
    }
    if (sc_device_count() <= 0) {
        fprintf(stderr, "no MI355X visible\n");
        return 1;
    }
    sc_engine *w = nullptr, *b = nullptr;
    sc_net_config wc{a.blocks, a.channels, a.white_seed}, bc{a.blocks, a.channels, a.black_seed};
    if (sc_engine_create(&wc, a.white_checkpoint == "<not-specified>" ? nullptr : a.white_checkpoint.c_str(), 0, &w) ||
        sc_engine_create(&bc, a.black_checkpoint == "<not-specified>" ? nullptr : a.black_checkpoint.c_str(), 0, &b)) {
        fprintf(stderr, "%s\n", sc_last_error());
        return 1;
    }
    printf("Players loaded.\n");   // play.rs:455
    sc_selfplay_config c{};
    c.n_slots = c.n_games = a.games;
    c.rollout_num = a.rollout;
    c.num_steps = 200;             // play.rs:325
    c.cpuct = a.cpuct;
    c.temperature = a.temperature;
    c.temperature_switch = a.temperature_switch;
    c.epsilon = 0.15f;
    c.with_noise = 0;              // play.rs:250
    c.outcome_gate = -1;           // play.rs:335: after every ply
    c.evaluator = SC_EVAL_NET;
    c.seed = a.seed;
    c.tie_random = 1;              // play.rs:268-277
    sc_selfplay* sp = nullptr;
    int rc = sc_selfplay_create(w, 0, &c, &sp);
    if (!rc) rc = sc_selfplay_set_players(sp, w, b, 0, 0);
    if (!rc) rc = sc_selfplay_run(sp, 0);
    if (rc) {
        fprintf(stderr, "%s\n", sc_last_error());
        return 1;
    }
    int white = 0, black = 0, draw = 0, none = 0;
    for (int g = 0; g < a.games; g++) {
        sc_trace_info info{};
        if (sc_selfplay_get_trace(sp, g, &info, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)) continue;
        if (!info.has_outcome) none++;
        else if (info.winner == 1) white++;
        else if (info.winner == 0) black++;
        else draw++;
        std::string path = out_name(a, (int)info.game_id + 1);
        if (sc_selfplay_write_trace_json(sp, g, path.c_str())) {
            fprintf(stderr, "%s\n", sc_last_error());
            rc = 1;
        }
    }
    // Total/WhiteWin/BlackWin is the input format of scripts/elo.py
    printf("games %d white-wins %d black-wins %d draws %d unfinished %d   (elo.py input: %d/%d/%d)\n", a.games, white, black, draw, none,
           a.games, white, black);
    sc_selfplay_destroy(sp);
    sc_engine_destroy(w);
    sc_engine_destroy(b);
    return rc;
}
