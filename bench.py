#!/usr/bin/env python3
"""bench.py -- MCTS simulations/sec of the self-play hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one ply of self-play for every concurrent game on a GPU = `rollout` (180) simulation steps of
the whole hot path (select -> encode -> network -> expand -> backup, mcts.rs:261-288) over `games` (256)
boards, plus the per-ply move choice (mcts::step).  Workload (BASELINE.json configs[1]): 256 concurrent
self-play games per GPU from the start position, rollout = 180, 10-block ResNet bf16, Dirichlet noise on,
cpuct 2.5, temperature switch 4 (README.md:39 of the reference).  Everything is resident in HBM before the
timed region; games shard across ranks with no collective on the data path (weak scaling).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the step launch `k_step` -- search wave, network tower
and value-FC tile of every game in one launch, 96 % of GPU time -- timed with HIP events around sampled launches on the
launch stream and priced with the FULL forward's FLOP against the dense MFMA peak) and, at N=1, `cpu_baseline` (the CPU
oracle = port of the reference algorithm, timed on the host cores on a bounded sample).  `also_tower` keeps the tower-only
figure (separate launches), `phases_us` the split of a step launch, `also_encode_steps` / `also_match` the widened rows
(SURVEY.md 8f rank 1 and 2).  Exits non-zero when the engine reports error flags.
"""
import argparse
import json
import os
import sys
import time

# Kernel arguments in device memory instead of host-coherent memory: every launch otherwise starts with a PCIe round trip
# for its argument block (+6 % simulations/s measured in round 1, with three launches per simulation step).  Must be set before the HIP
# runtime initialises, i.e. before torch / the engine library are loaded.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 (block-scaled e4m3) MFMA peak, same table
HBM_PEAK_GBS = 8000.0


def macs_per_position(n_blocks, C, tower_only=False):
    """MACs of one forward of one position (BASELINE.md section 2 / SURVEY.md section 8d).  tower_only: without the value
    head's two Linear layers, which run in k_value_fc1 / the search kernel's fused tail, not in the tower launch that the
    roofline figure times."""
    H = 256
    stem = 64 * 112 * 9 * C
    block = 2 * 64 * C * 9 * C + 2 * C * (C // 2)
    policy = 64 * C * H + 64 * H * 73
    value_fc = (64 * H + 7) * 128 + 128
    value = 64 * C * H + (0 if tower_only else value_fc)
    return stem + n_blocks * block + policy + value


def shard(total_games_per_rank, rank):
    """game-id range of a rank: ids are globally unique so traces from different GPUs never collide"""
    return rank * total_games_per_rank


def cpu_baseline(n_blocks, C, budget_s, rollout):
    """Reference algorithm on the host cores: the CPU oracle (oracle/, a port -- the Rust binary cannot be
    built here) in reference-faithful mode (network re-evaluated at every node of every descent, batch 1,
    fp32, src/mcts.rs:152), one game per core."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py as orc

    orc.build()
    # every core this process may run on (the reference's convention is one single-threaded game per core,
    # dockerfile:14 OMP_NUM_THREADS=1 + scripts/run_batch's `parallel -j`)
    total_cores = os.cpu_count() or 1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else total_cores
    try:   # a container's CPU quota (cgroup v2 cpu.max) is the real share: more threads than that only time-slice
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = max(1, cores)
    net = orc.Net(n_blocks, C, seed=1)
    ev = orc.eval_fn("orc_eval_net")

    def worker(i, faithful, budget):
        # plays on from the start position: `rollout` simulations per ply, then the most visited move
        st = orc.State()
        t0 = time.time()
        n = evals = 0
        while time.time() - t0 < budget:
            s = orc.Search(st)
            k = 0
            while k < rollout and time.time() - t0 < budget:
                s.sim(evaluator=ev, user=net.h, cpuct=2.5, epsilon=0.15, with_noise=True, faithful=faithful)
                k += 1
            n += k
            evals += s.num_evals()
            d = s.dump()
            if k < rollout or d["n_child"][0] == 0:
                break
            kids = d["n"][d["first_child"][0]:d["first_child"][0] + d["n_child"][0]]
            st.push(int(d["move"][d["first_child"][0] + int(kids.argmax())]))
        return n, evals, time.time() - t0

    out = {}
    for name, faithful, budget in (("faithful", True, budget_s * 0.7), ("cached", False, budget_s * 0.3)):
        t0 = time.time()
        with ThreadPoolExecutor(cores) as ex:
            res = list(ex.map(lambda i: worker(i, faithful, budget), range(cores)))
        wall = time.time() - t0
        out[name] = (sum(r[0] for r in res) / wall, sum(r[1] for r in res), wall)
    return {
        "value": round(out["faithful"][0], 2),
        "unit": "simulations/s",
        "cores": cores,
        "cores_total": total_cores,
        "kind": "port",
        "sample": (f"{cores} self-play games from the start position (one per core, 1 thread each, rollout={rollout} per ply) "
                   f"for {out['faithful'][2]:.1f} s wall, fp32 {n_blocks}x{C} net re-evaluated at every path node "
                   f"(reference-faithful, src/mcts.rs:152; {out['faithful'][1]} net calls); CPU oracle = restatement of the "
                   "reference algorithm, not the Rust binary"),
        "cached_prior_value": round(out["cached"][0], 2),
    }


def run_gpu(args, rank, world, local_rank):
    import scamd

    R = args.rollout
    res = {}
    # main = the BASELINE configuration; the optional extras (single GPU only, short) are reported under "also":
    # the other trunk width, and twice the games in two interleaved groups (one group's search and value FC1 run under
    # the other group's network launch -- what a 512-games-per-GPU deployment gets from the same kernels)
    runs = [("main", args.channels, args.games, max(1, args.groups))]
    if args.alt and world == 1:
        runs += [("steady", args.channels, args.games, 1),
                 ("alt", 256 if args.channels == 128 else 128, args.games, max(1, args.groups)), ("x2", args.channels, 2 * args.games, 2),
                 # BASELINE configs[4]: fp8 (e4m3) network, 4096 games over 8 GPUs = 512 per GPU; and at the headline's 256
                 ("fp8", args.channels, args.games, 1), ("fp8_512", args.channels, 2 * args.games, 1),
                 # a random-init net has nearly flat priors (shallow trees); a trained one is sharp.  Same weights with the
                 # policy head's last LayerNorm gain x 8: the search descends deeper, the network cost is unchanged
                 ("sharp", args.channels, args.games, 1),
                 # BASELINE configs[3]'s per-GPU slice: 20 blocks x 256 channels, rollout 800 (one timed ply)
                 ("deep", 256, args.games, 1)]
    for tag, C, G, K in runs:
        progress(f"run {tag}: {G} games, {K} group(s)")
        prec = "fp8" if tag.startswith("fp8") else args.precision
        R = 800 if tag == "deep" else args.rollout
        blocks = 20 if tag == "deep" else args.blocks
        if tag == "sharp":
            eng = sharp_prior_engine(scamd, args.blocks, C, local_rank, prec)
        else:
            eng = scamd.Engine(blocks, C, seed=1, device=local_rank, precision=prec)
        assert G % K == 0
        sps = [scamd.SelfPlay(eng, n_slots=G // K, n_games=10 ** 7 // K, trace_capacity=4 * G // K, rollout_num=R, num_steps=150,
                              cpuct=2.5, temperature=0.0, temperature_switch=4, epsilon=0.15, with_noise=True, seed=1234,
                              first_game_id=shard(10 ** 7, rank) + k * (10 ** 7 // K), own_stream=K > 1, device=local_rank)
               for k in range(K)]

        def enqueue(n):
            if K == 1:
                sps[0].enqueue(n)
            else:
                scamd.enqueue_interleaved(sps, n)

        def stats():
            tot = {}
            for sp in sps:
                for k, v in sp.stats().items():
                    tot[k] = tot.get(k, 0) + v if k != "error_flags" else tot.get(k, 0) | v
            return tot

        steps = args.steps if tag == "main" else 1 if tag == "deep" else max(2, args.steps // 4)
        if tag == "steady":
            seed_mid_game_positions(scamd, eng, sps[0], G, R)
        enqueue((1 if tag == "deep" else args.warmup) * R)
        for sp in sps:
            sp.enable_timing(-args.timing_stride)   # every n-th step launch bracketed as a whole, launch form unchanged
            sp.timing(reset=True)
        # `repeats` timed regions of exactly `steps` plies each, every one bracketed by barrier + synchronise on both sides;
        # the reported region is the median one (median of 3)
        regions = []
        nn0 = stats()
        for _ in range(args.repeats if tag == "main" else 1):
            s0 = stats()
            for sp in sps:
                sp.sync()
            cuda_sync()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                enqueue(R)
            for sp in sps:
                sp.sync()
            cuda_sync()
            t1 = time.perf_counter()
            barrier()
            s1 = stats()
            regions.append((t1 - t0, s1["sims_done"] - s0["sims_done"]))
        tms = [sp.timing() for sp in sps]
        s1 = stats()
        nl = sum(t["tower_launches"] for t in tms)
        res[tag] = dict(C=C, precision=prec, regions=regions, steps=steps, blocks=blocks, rollout=R, nn_evals=s1["nn_evals"] - nn0["nn_evals"],
                        sims_all=s1["sims_done"] - nn0["sims_done"], err=s1["error_flags"],
                        step_ms=sum(t["ms_tower_sum"] for t in tms) / max(nl, 1), steps_timed=nl,
                        span_ms=max(t["ms_total"] for t in tms), groups=K, games=G, launches_per_step=sps[0].launches_per_step())
        if tag == "main" and K == 1:
            progress("phase stamps")
            res[tag]["phases_us"] = step_phases(sps[0], R)
            progress("tower-only sample")
            # the tower launch alone: a few plies with every 4th step as separate launches, the tower bracketed by events
            sps[0].enable_timing(4)
            sps[0].timing(reset=True)
            enqueue(2 * R)
            tt = sps[0].timing()
            res[tag]["tower_ms"] = tt["ms_tower_sum"] / max(tt["tower_launches"], 1)
            res[tag]["tower_launches"] = tt["tower_launches"]
            res[tag]["err"] |= stats()["error_flags"]
        if tag == "steady":
            plies = [sps[0].slot(g)["ply"] for g in range(0, G, 8)]
            res[tag]["ply_min_max"] = (min(plies), max(plies))
        if tag in ("main", "sharp", "steady"):   # tree levels walked by the last descent of a sample of games
            lens = [len(sps[0].slot(g)["path"]) for g in range(0, G, 4)]
            res[tag]["mean_path_len"] = sum(lens) / len(lens)
        if tag == "main" and args.alt and world == 1:
            progress("encode_steps")
            res["encode_steps"] = encode_steps_rate(scamd, eng, local_rank)
        for sp in sps:
            sp.close()
        eng.close()
    if args.alt and world == 1:
        progress("match")
        res["match"] = match_rate(scamd, args.blocks, args.channels, local_rank)
    return res


def step_phases(sp, R):
    """per-phase split of the one-launch step from the kernel's own wall-clock stamps (100 MHz; csrc/step_kernels.hip
    PHASE_STAMP): medians over the workgroups of several launches, microseconds"""
    import numpy as np
    if sp.launches_per_step() != 1:
        return None
    sp.debug_cycles(True)
    acc = []
    for _ in range(12):
        sp.enqueue(7)
        a = sp.debug_cycles(True, read=True).astype(np.int64)   # (plain stream sync: the stamps of the last step launch)
        acc.append(a[:, 24:28])
    a = np.concatenate(acc)
    a = a[(a[:, 1] > a[:, 0]) & (a[:, 2] > a[:, 1]) & (a[:, 3] >= a[:, 2])]    # workgroups that ran all phases
    if len(a) == 0:
        return None
    us = lambda x: round(float(np.median(x)) / 100.0, 2)
    return {"search": us(a[:, 1] - a[:, 0]), "tower": us(a[:, 2] - a[:, 1]), "value_fc_tile": us(a[:, 3] - a[:, 2]),
            "workgroup_total": us(a[:, 3] - a[:, 0]), "workgroups_sampled": int(len(a)),
            "note": "medians over workgroups of 12 step launches: kernel entry -> leaf selected (expand + backup of the previous simulation, "
                    "descent, move generation, plane encoding) -> network done -> value_head.ffn.0 tile done; the launch also pays its ramp "
                    "and the slowest workgroup"}


def encode_steps_rate(scamd, eng, device):
    """SURVEY.md 8f rank 1 (libsmartchess.chess_encode_steps, reference src/lib.rs:46-128): traces of 256 quick self-play games
    -> training tensors through sc_encode_steps.  The ABI hands over host buffers, so the call includes the PCIe copy-out of
    26.3 KB per ply; the kernels' own time (HIP events) prices the HBM write rate."""
    quick = scamd.SelfPlay(eng, n_slots=256, n_games=256, rollout_num=8, num_steps=100, cpuct=2.5, temperature=0.0, temperature_switch=8,
                           epsilon=0.15, with_noise=True, seed=5, outcome_gate=10 ** 6, device=device)
    quick.run()
    games = []
    for g in range(256):
        tr = quick.trace(g)
        games.append([(s[0], [(c[0], c[1]) for c in s[2]]) for s in tr["steps"]])
    quick.close()
    scamd.encode_steps_batch(games[:8], device=device, engine=eng)   # warm-up (allocations, first launch)
    t0 = time.perf_counter()
    r = scamd.encode_steps_batch(games, device=device, engine=eng)
    wall = time.perf_counter() - t0
    k_ms, call_ms = scamd.binding.encode_steps_last_timing()
    plies = int(r["ply_off"][-1])
    assert int((r["status"] != 0).sum()) == 0
    out_bytes = plies * (7168 + 28 + 4672 * 4 + 224 * 2 + 4)
    return {"plies": plies, "games": len(games), "plies_per_s": round(plies / (call_ms * 1e-3), 1), "call_ms": round(call_ms, 2),
            "kernels_ms": round(k_ms, 3), "python_wall_ms": round(wall * 1e3, 2), "kernels_plies_per_s": round(plies / (k_ms * 1e-3), 1),
            "bytes_written_per_ply": out_bytes // plies,
            "roofline": {"bound": "hbm", "achieved": round(out_bytes / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(out_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernel": "k_replay_raw + k_ply_keys + k_ply_rep + k_encode_plies + k_steps_dist",
                         "note": "algorithmic bytes written (planes 7168 + meta 28 + dist 18688 + move indices 448 + count 4 per ply) / HIP-event "
                                 "time of the five kernels"},
            "pcie_share_of_call": round(1.0 - k_ms / call_ms, 4),
            "note": "host-pointer ABI: the call = H2D of the traces + kernels + D2H of the tensors (pageable host memory) + host bookkeeping"}


def match_rate(scamd, blocks, C, device):
    """SURVEY.md 8f rank 2 (`play`, reference src/play.rs:318-343; scripts/leader-board:44-54): 2 x 100 games between two 10x128 networks
    -- both colour assignments, played at the same time on two streams -- rollout 100, noise off, outcome after every ply, <= 200 plies"""
    a, b = scamd.Engine(blocks, C, seed=1, device=device), scamd.Engine(blocks, C, seed=2, device=device)
    t0 = time.perf_counter()
    r = scamd.play_match(a, b, n_games=100, rollout=100, cpuct=1.5, temperature=0.0, temperature_switch=0, num_steps=200, seed=3, swap=True)
    wall = time.perf_counter() - t0
    plies = sum(len(t["steps"]) for k in ("as_white", "as_black") for t in r[k]["traces"] if t)
    a.close()
    b.close()
    return {"games": 200, "rollout": 100, "nets": f"two {blocks}x{C} bf16 (seeds 1, 2)", "wall_s": round(wall, 3), "games_per_s": round(200 / wall, 2),
            "plies": plies, "simulations_per_s": round(plies * 100 / wall, 1), "results_as_white": r["as_white"]["results"],
            "results_as_black": r["as_black"]["results"], "elo_a_minus_b": r["elo_a_minus_b"] if abs(r["elo_a_minus_b"]) != float("inf") else None,
            "note": "the reference's leader-board match: 100 games per colour assignment, each assignment in lockstep on its own handle and "
                    "stream (sc_selfplay_set_players), the two running side by side; games end at different plies, so late plies run with few "
                    "live games"}


def sharp_prior_engine(scamd, n_blocks, C, device, precision):
    """the seed-1 network with policy_head.model.3.weight (the gain of the policy head's last LayerNorm) multiplied by 8:
    logits 8x larger, priors concentrated on a few moves like a trained network's"""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scw
    sd = scw.prng_state_dict(n_blocks, C, seed=1)
    sd["policy_head.model.3.weight"] = sd["policy_head.model.3.weight"] * 8.0
    with tempfile.NamedTemporaryFile(suffix=".scw", delete=False) as f:
        path = f.name
    try:
        scw.write_scw(path, sd, n_blocks, C)
        return scamd.Engine(weights=path, device=device, precision=precision)
    finally:
        os.unlink(path)


def seed_mid_game_positions(scamd, eng, sp, G, R):
    """Steady-state stand-in: a self-play node does not march 256 games in lockstep from the start position -- its slots
    hold games at every stage.  A few quick games (rollout 16) are played to 150 plies with the same network, and slot g
    of the measured handle starts from the first g*150/G moves of one of them: plies 0..149 are spread evenly over the
    slots (deeper histories, mid-game branching factors, repetition planes set)."""
    quick = scamd.SelfPlay(eng, n_slots=8, n_games=8, rollout_num=16, num_steps=150, cpuct=2.5, temperature=0.0,
                           temperature_switch=8, epsilon=0.15, with_noise=True, seed=77, outcome_gate=10 ** 6)
    quick.run()
    lines = [[s[0] for s in quick.trace(g)["steps"]] for g in range(8)]
    quick.close()
    for g in range(G):
        line = lines[g % 8]
        sp.set_position(g, line[:min(len(line), (g * 150) // G)])


_dist = None


def progress(msg):
    """stderr only (stdout carries the one JSON line); rank 0"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def barrier():
    if _dist is not None:
        _dist.barrier()


def cuda_sync():
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass


def main():
    global _dist
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps plies each; the median one is reported")
    ap.add_argument("--games", type=int, default=256, help="concurrent games per GPU")
    ap.add_argument("--rollout", type=int, default=180)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--channels", type=int, default=128, help="trunk width: 128 = BASELINE configs[1]; 256 = reference module")
    ap.add_argument("--no-alt", dest="alt", action="store_false", help="skip the short extra runs (steady state, other width, 2x games, fp8)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp8"], help="network precision of the main run (BASELINE configs[1]: bf16)")
    ap.add_argument("--groups", type=int, default=1, help="split the games of a GPU into K groups on K HIP streams (overlap)")
    ap.add_argument("--timing-stride", type=int, default=8,
                    help="every n-th simulation step's launch(es) are bracketed by a HIP event pair on the launch stream for the roofline "
                         "figure (the launch form is not changed)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="HARNESS TEST ONLY (gloo, no GPU): exercises sharding/timing/aggregation with the oracle's "
                         "synthetic self-play as stand-in workload; its numbers are meaningless")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: start one rank per GPU as a child (nothing has touched the GPU in this process)
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    # under a launcher (torch.distributed.run sets RANK / MASTER_ADDR) the process group is used even with one rank, so the
    # RCCL path -- rendezvous, barrier, all-reduce of the timing scalars on device tensors -- is the same code at every N
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    torch = None
    try:
        import torch  # plumbing only: rendezvous, barrier, max-over-ranks
    except Exception:
        if use_dist:
            raise
    if use_dist:
        import torch.distributed as dist
        backend = "gloo" if args.cpu_dry_run else "nccl"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        _dist = dist

    if args.cpu_dry_run:
        from oracle import oracle_py as orc
        regions = []
        for rep_i in range(args.repeats):
            barrier()
            t0 = time.perf_counter()
            sims = 0
            for k in range(args.steps):
                g = orc.selfplay_game(rollout_num=8, num_steps=2, seed=1234, game_id=shard(10 ** 7, rank) + rep_i * args.steps + k)
                sims += g["n_sims"]
            t1 = time.perf_counter()
            barrier()
            regions.append((t1 - t0, sims))
        res = {"main": dict(C=args.channels, regions=regions, steps=args.steps, nn_evals=0, sims_all=sum(r[1] for r in regions), err=0,
                            step_ms=0.0, steps_timed=0, span_ms=0.0)}
    else:
        res = run_gpu(args, rank, world, local_rank)

    m = res["main"]
    # per region: MAX of the time over ranks, SUM of the simulations; the median region (by throughput) is reported
    secs = [r[0] for r in m["regions"]]
    simc = [float(r[1]) for r in m["regions"]]
    if _dist is not None:
        t = torch.tensor(secs, dtype=torch.float64)
        sv = torch.tensor(simc, dtype=torch.float64)
        if not args.cpu_dry_run:
            t, sv = t.cuda(), sv.cuda()
        _dist.all_reduce(t, op=_dist.ReduceOp.MAX)   # timing only -- no collective on the data path
        _dist.all_reduce(sv, op=_dist.ReduceOp.SUM)
        secs, simc = [float(x) for x in t.tolist()], [float(x) for x in sv.tolist()]
    rates = [n / t_ for n, t_ in zip(simc, secs)]
    mid = sorted(range(len(rates)), key=lambda i: rates[i])[len(rates) // 2]
    seconds, sims = secs[mid], simc[mid]

    def rate(r):
        t_, n_ = r["regions"][0]
        return n_ / t_, 1e3 * t_ / r["steps"]

    if rank == 0:
        flop_pos = 2.0 * macs_per_position(args.blocks, m["C"])
        flop_tower = 2.0 * macs_per_position(args.blocks, m["C"], tower_only=True)
        out = {
            # BASELINE.json's metric string, verbatim, for the configuration it is quoted on
            "metric": ("MCTS simulations/sec (whole node), self-play rollout=180, at 1/2/4/8 MI355X" if args.rollout == 180
                       else f"MCTS simulations/sec (whole node), self-play rollout={args.rollout}"),
            "value": round(sims / seconds, 1),
            "unit": "simulations/s",
            "n_gpus": world,
            "steps": m["steps"],
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * seconds / m["steps"], 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": m.get("precision", "bf16"),
            "data": "synthetic" if not args.cpu_dry_run else "cpu-dry-run (oracle stand-in; harness test only)",
            "config": {
                "workload": (f"{config_name(args, m)}: {args.games} concurrent self-play games per GPU from the start position, "
                             f"rollout={args.rollout}, {args.blocks}-block/{m['C']}-ch SE-ResNet {m.get('precision', 'bf16')} (random-init), "
                             "cpuct 2.5, Dirichlet(0.3) eps 0.15, temperature switch 4"),
                "games_per_gpu": args.games, "rollout": args.rollout, "net": f"{args.blocks}x{m['C']}",
                "step": "one ply = rollout simulation steps over all games", "parallelism": f"games sharded over {world} GPU(s), no collective",
            },
            "repeats": {"regions": len(rates), "reported": "median", "values": [round(x, 1) for x in rates]},
        }
        if not args.cpu_dry_run:
            pos_per_launch = args.games // m.get("groups", 1)
            peak = PEAK_FP8_TFLOPS if m.get("precision") == "fp8" else PEAK_BF16_TFLOPS
            prec = m.get("precision", "bf16")
            form = {1: f"k_step<{prec}, {m['C']}>", 2: f"k_step<{prec}, {m['C']}> + k_value_fc1", 3: "k_mcts + k_tower32 + k_value_fc1"}[m["launches_per_step"]]
            tf = pos_per_launch * flop_pos / (m["step_ms"] * 1e-3) / 1e12 if m["step_ms"] > 0 else 0.0
            out["roofline"] = {
                "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                "traffic": load_traffic(m["C"], prec, m["launches_per_step"]),
                "kernel": form, "avg_launch_ms": round(m["step_ms"], 4), "launches_timed": m["steps_timed"],
                "flop_per_launch": pos_per_launch * flop_pos, "positions_per_launch": pos_per_launch, "concurrent_groups": m.get("groups", 1),
                "launches_per_step": m["launches_per_step"],
                "algorithmic_bytes_per_launch": algorithmic_bytes(args.blocks, m["C"], prec, pos_per_launch),
                "flop_note": "2 x MACs of one whole forward (stem, blocks, both heads incl. value_head.ffn: SURVEY.md 8d) per game and launch; "
                             "the launch also runs every game's search step (expand + backup, PUCT descent, move generation, plane encoding), "
                             "which is not matrix work and is priced as zero FLOP",
                "timing_note": f"HIP events on the launch stream around every {args.timing_stride}th step launch of the timed regions",
                "chip_sustains_note": "a register-resident loop of independent bf16 32x32x16 MFMAs with non-zero operands holds 1989 TFLOP/s "
                                      "(0.80 of peak) at this pool's 1400 W cap (profiles/r02_exp_power_and_clock.txt)",
                "end_to_end_frac": round(sims / seconds / world * flop_pos / (peak * 1e12), 4),
            }
            if m.get("phases_us"):
                out["roofline"]["phases_us"] = m["phases_us"]
            if m.get("tower_ms"):
                tft = pos_per_launch * flop_tower / (m["tower_ms"] * 1e-3) / 1e12
                out["also_tower"] = {"kernel": f"k_tower32<{prec}, {m['C']}>", "avg_launch_ms": round(m["tower_ms"], 4), "launches_timed": m["tower_launches"],
                                     "achieved": round(tft, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tft / peak, 4),
                                     "flop_per_launch": pos_per_launch * flop_tower, "traffic": load_traffic(m["C"], prec, 0),
                                     "note": "the network tower as its own launch (stem, blocks, head convs; no value_head.ffn, no search): production "
                                             "runs it only inside k_step -- sampled here on separately launched steps after the timed regions"}
            out["nn_evals_per_sim"] = round(m["nn_evals"] / max(m["sims_all"], 1), 4)
            out["error_flags"] = m["err"]
            if "steady" in res:
                st = res["steady"]
                v, ms = rate(st)
                out["also_steady"] = {"value": round(v, 1), "ms_per_step": round(ms, 3), "plies_min_max": list(st["ply_min_max"]),
                                      "nn_evals_per_sim": round(st["nn_evals"] / max(st["sims_all"], 1), 4), "error_flags": st["err"],
                                      "note": ("same configuration, but the slots hold games at every stage: slot g starts from the first "
                                               "g*150/G plies of a pre-played game (mid-game branching factors, full 8-board histories)")}
            if "sharp" in res:
                st = res["sharp"]
                v, ms = rate(st)
                out["also_sharp_priors"] = {"value": round(v, 1), "ms_per_step": round(ms, 3), "mean_path_len": round(st["mean_path_len"], 2),
                                            "mean_path_len_main": round(m.get("mean_path_len", 0.0), 2), "error_flags": st["err"],
                                            "note": ("same network with the policy head's last LayerNorm gain x 8 (priors concentrated like a trained "
                                                     "net's): deeper descents, same network cost")}

            def step_frac(x, pk):
                f = 2.0 * macs_per_position(x["blocks"], x["C"])
                return round(x["games"] // x["groups"] * f / (x["step_ms"] * 1e-3) / 1e12 / pk, 4) if x["step_ms"] > 0 else None
            if "alt" in res:
                a = res["alt"]
                v, ms = rate(a)
                out["also"] = {"net": f"{args.blocks}x{a['C']}", "value": round(v, 1), "ms_per_step": round(ms, 3), "error_flags": a["err"],
                               "step_avg_ms": round(a["step_ms"], 4), "launches_per_step": a["launches_per_step"],
                               "roofline_frac": step_frac(a, PEAK_BF16_TFLOPS)}
            if "deep" in res:
                x = res["deep"]
                v, ms = rate(x)
                out["also_deep"] = {"config": "BASELINE configs[3], one GPU's slice: 20 blocks x 256 channels bf16, rollout 800, "
                                              f"{x['games']} concurrent games (one timed ply)",
                                    "net": f"{x['blocks']}x{x['C']}", "rollout": x["rollout"], "value": round(v, 1), "ms_per_step": round(ms, 3),
                                    "step_avg_ms": round(x["step_ms"], 4), "launches_per_step": x["launches_per_step"], "error_flags": x["err"],
                                    "roofline_frac": step_frac(x, PEAK_BF16_TFLOPS)}
            for tag in ("fp8", "fp8_512"):
                if tag in res:
                    x = res[tag]
                    v, ms = rate(x)
                    f8 = 2.0 * macs_per_position(args.blocks, x["C"])
                    tf8 = x["games"] * f8 / (x["step_ms"] * 1e-3) / 1e12 if x["step_ms"] > 0 else 0.0
                    out["also_" + tag] = {
                        "config": ("BASELINE configs[4] sizing: fp8 (OCP e4m3) policy/value net on the CDNA4 fp8 matrix cores, "
                                   f"{x['games']} concurrent games per GPU" + ("" if tag == "fp8_512" else " (the headline's game count)")),
                        "dtype": "fp8", "games_per_gpu": x["games"], "net": f"{args.blocks}x{x['C']}", "value": round(v, 1), "ms_per_step": round(ms, 3),
                        "error_flags": x["err"],
                        "roofline": {"bound": "mfma", "achieved": round(tf8, 2), "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(tf8 / PEAK_FP8_TFLOPS, 4), "kernel": f"k_step<fp8, {x['C']}>",
                                     "avg_launch_ms": round(x["step_ms"], 4), "launches_timed": x["steps_timed"], "launches_per_step": x["launches_per_step"],
                                     "flop_per_launch": x["games"] * f8, "positions_per_launch": x["games"],
                                     "traffic": load_traffic(x["C"], "fp8", x["launches_per_step"], x["games"]),
                                     "end_to_end_frac": round(v * f8 / (PEAK_FP8_TFLOPS * 1e12), 4)}}
            if "x2" in res:
                x = res["x2"]
                v, ms = rate(x)
                out["also_2x_games"] = {"games_per_gpu": x["games"], "groups": x["groups"], "net": f"{args.blocks}x{x['C']}",
                                        "value": round(v, 1), "ms_per_step": round(ms, 3), "error_flags": x["err"],
                                        "note": "two interleaved groups of games on two HIP streams; not the BASELINE configuration"}
            if "encode_steps" in res:
                out["also_encode_steps"] = res["encode_steps"]
            if "match" in res:
                out["also_match"] = res["match"]
            if world == 1 and args.cpu_budget > 0:
                progress("cpu_baseline")
                out["cpu_baseline"] = cpu_baseline(args.blocks, m["C"], args.cpu_budget, args.rollout)
        print(json.dumps(out), flush=True)
    if _dist is not None:
        _dist.barrier()
        _dist.destroy_process_group()
    # invalid games must not pass for a measurement (include/sc_engine.h: sc_selfplay_stats.error_flags)
    bad = sorted(k for k, v in res.items() if isinstance(v, dict) and v.get("err"))
    if bad:
        raise SystemExit(f"error flags set in run(s) {bad}: the numbers above are not valid")


def config_name(args, m):
    """which BASELINE.json configuration the arguments are"""
    prec = m.get("precision", "bf16")
    if (args.games, args.rollout, args.blocks, m["C"], prec) == (256, 180, 10, 128, "bf16"):
        return "BASELINE configs[1]"
    if (args.games, args.rollout, args.blocks, m["C"], prec) == (256, 800, 20, 256, "bf16"):
        return "BASELINE configs[3], per-GPU slice (20x256, rollout 800)"
    if prec == "fp8":
        return "BASELINE configs[4] variant (fp8 network)"
    return "custom configuration"


def load_traffic(C, precision="bf16", launches_per_step=1, games=256):
    """HBM-side bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes: tools/profile_summarize.py); null if that configuration was not collected.
    launches_per_step 0 = the stand-alone tower launch."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if launches_per_step == 0:
        key = f"k_tower32<{C}>" if precision == "bf16" else f"k_tower32<{precision},{C}>"
    else:
        key = f"k_step<{precision},{C}>" + ("+fc1" if launches_per_step == 1 else "") + (f"@{games}" if games != 256 else "")
    try:
        return json.load(open(p)).get(key)
    except Exception:
        return None


def algorithmic_bytes(n_blocks, C, precision, positions):
    """HBM bytes one step launch HAS to move: the weights once (conv operands at 2 or 1 B per element, SE and value FC
    bf16, parameters fp32) + ~20 KB per simulation for the search (tree statistics, position records, NN input, priors).  What the
    launch moves on top of that -- value-head feature rows and split-K partials written and read back (128 KB per position), one copy
    of the weights per XCD L2 -- is the implementation's own traffic: `roofline.traffic` minus this figure."""
    wb = 2 if precision == "bf16" else 1
    H = 256
    convs = 112 * 9 * C + n_blocks * 2 * 9 * C * C + C * H + C * H + H * 73
    se = n_blocks * 2 * C * (C // 2)
    fc = (64 * H + 7) * 128 + 128
    params = 4 * (8 * C * n_blocks + 8 * C + 6 * H + 2 * 73 + 256)
    return int(convs * wb + (se + fc) * 2 + params + positions * 20 * 1024)


if __name__ == "__main__":
    main()
