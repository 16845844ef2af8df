"""developer tool: same-box A/B of the headline configuration between two builds of the library (default: this tree's lib/ against a
build of the round-2 tree in lib_r02/), with a minimal loader that binds only what it needs (the ABI grew since round 2).
    python tools/ab_r02.py [libA] [libB]"""
import ctypes as C, os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    if "@" in sys.argv[2]:   # lib@VAR=value[,VAR=value]: environment of this child (A/B of a runtime switch within one build)
        sys.argv[2], envs = sys.argv[2].split("@", 1)
        for kv in envs.split(","):
            os.environ[kv.split("=")[0]] = kv.split("=", 1)[1]
    L = C.CDLL(sys.argv[2])
    prec, games = int(sys.argv[3]), int(sys.argv[4])
    CH = int(os.environ.get("SC_AB_CHANNELS", "128"))
    class NetConfig(C.Structure):
        _fields_ = [("n_res_blocks", C.c_int32), ("channels", C.c_int32), ("seed", C.c_uint64), ("precision", C.c_int32), ("reserved", C.c_int32)]
    class SpCfg(C.Structure):
        _fields_ = [("n_slots", C.c_int32), ("n_games", C.c_int32), ("rollout_num", C.c_int32), ("num_steps", C.c_int32), ("cpuct", C.c_float), ("temperature", C.c_float),
                    ("temperature_switch", C.c_int32), ("epsilon", C.c_float), ("with_noise", C.c_int32), ("outcome_gate", C.c_int32), ("evaluator", C.c_int32),
                    ("external_noise", C.c_int32), ("seed", C.c_uint64), ("first_game_id", C.c_uint64), ("trace_capacity", C.c_int32), ("own_stream", C.c_int32),
                    ("tie_random", C.c_int32), ("trace_hold", C.c_int32), ("rollout_factor", C.c_float)]
    class Stats(C.Structure):
        _fields_ = [("sims_done", C.c_int64), ("nn_evals", C.c_int64), ("games_finished", C.c_int32), ("games_active", C.c_int32), ("error_flags", C.c_int32), ("plies_done", C.c_int32)]
    eng, sp = C.c_void_p(), C.c_void_p()
    nc = NetConfig(10, CH, 1, prec, 0)
    assert L.sc_engine_create(C.byref(nc), None, 0, C.byref(eng)) == 0
    cfg = SpCfg(games, 10 ** 7, 180, 150, 2.5, 0.0, 4, 0.15, 1, 100, 0, 0, 1234, 0, 4 * games, 0, 0, 0, 0.0)
    assert L.sc_selfplay_create(eng, 0, C.byref(cfg), C.byref(sp)) == 0
    L.sc_selfplay_enqueue_sims(sp, 360); L.sc_selfplay_synchronize(sp)
    vals = []
    for rep in range(3):
        st0, st1 = Stats(), Stats()
        L.sc_selfplay_get_stats(sp, C.byref(st0))
        t0 = time.perf_counter()
        L.sc_selfplay_enqueue_sims(sp, 20 * 180); L.sc_selfplay_synchronize(sp)
        dt = time.perf_counter() - t0
        L.sc_selfplay_get_stats(sp, C.byref(st1))
        vals.append((st1.sims_done - st0.sims_done) / dt)
    print(json.dumps({"lib": sys.argv[2], "median_sims_per_s": sorted(vals)[1], "values": vals, "err": st1.error_flags}))
    sys.exit(0)
libs = sys.argv[1:] if len(sys.argv) > 2 else [os.path.join(ROOT, "smart-chess-rust_amd", d, "libsc_engine.so") for d in ("lib", "lib_r02")]
for prec, games in (((0, 256),) if os.environ.get("SC_AB_BF16_ONLY") else ((0, 256), (1, 256), (1, 512))):
    for rnd in range(2):          # A B A B: drift of the box shows as a difference between the two passes
        for lib in libs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, str(prec), str(games)], capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if line else {"median_sims_per_s": 0, "err": r.stderr[-300:]}
            tag = os.path.basename(os.path.dirname(lib.split("@")[0])) + ("@" + lib.split("@", 1)[1] if "@" in lib else "")
            print(f"{'fp8' if prec else 'bf16'} {games} games  {tag:8s} {d['median_sims_per_s'] / 1e6:.4f} M sims/s  err {d['err']}", flush=True)
