"""N>1 path of bench.py on CPU: world_size-2 gloo run of the harness (game-id sharding, barrier, max-over-ranks
timing, summed simulations, single JSON line from rank 0).  --cpu-dry-run swaps the GPU engine for the oracle's
tiny synthetic self-play: harness test only, never a product path."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_gloo():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0",
           "--cpu-dry-run"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["unit"] == "simulations/s" and out["value"] > 0 and out["vs_baseline"] is None
    # each rank played 3 stand-in games of 2 plies x 8 simulations: the sum over both ranks is reported
    assert abs(out["value"] * out["ms_per_step"] * 3 / 1e3 - 2 * 3 * 16) < 1e-3 * 2 * 3 * 16 + 1


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (no torchrun around it) starts one rank per GPU itself and prints ONE line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--cpu-dry-run"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["n_gpus"] == 2


def test_game_id_sharding_is_disjoint():
    sys.path.insert(0, ROOT)
    import bench
    ids = [bench.shard(10 ** 7, r) for r in range(8)]
    assert len(set(ids)) == 8 and all(b - a >= 10 ** 7 for a, b in zip(ids, ids[1:]))


def test_flop_model_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.macs_per_position(10, 256) == 783827968
    assert bench.macs_per_position(20, 256) == 1539458048
    assert bench.macs_per_position(10, 128) == 204653568
