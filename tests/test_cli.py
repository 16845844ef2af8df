"""sc-selfplay: the launcher with the reference's `selfplay` flags (src/main.rs:25-60)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "smart-chess-rust_amd", "lib", "sc-selfplay")


def _run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=600)


def test_cli_argument_checks():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
    import build as scbuild
    scbuild.build()
    assert os.path.exists(CLI)
    r = _run("--rollout-factor", "2", "--rollout-num", "10")          # main.rs:74 assert
    assert r.returncode == 2 and "both" in r.stderr
    r = _run("-d", "cpu")
    assert r.returncode == 2 and "no CPU path" in r.stderr
    r = _run("--bogus")
    assert r.returncode == 2


@pytest.mark.gpu
def test_cli_writes_reference_traces(tmp_path, orc):
    pat = str(tmp_path / "trace{}.json")
    r = _run("-d", "cuda", "--rollout-num", "24", "-n", "12", "--temperature", "0", "--cpuct", "2", "--temperature-switch", "4",
             "-t", pat, "--games", "6", "--concurrency", "4", "--blocks", "2", "--channels", "128", "--seed", "7")
    assert r.returncode == 0, r.stderr
    # one JSON statistics object per GPU (sims/s, games/s, occupancy -> JSON)
    stats = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(stats) == 1 and stats[0]["gpu"] == 0 and stats[0]["games"] == 6 and stats[0]["traces_written"] == 6
    assert stats[0]["simulations"] == 6 * 12 * 24 and stats[0]["error_flags"] == 0 and stats[0]["ok"] is True
    assert stats[0]["sims_per_s"] > 0 and stats[0]["launches_per_step"] in (1, 2, 3) and stats[0]["slots"] == 4
    for k in range(1, 7):
        js = json.load(open(str(tmp_path / f"trace{k}.json")))
        assert list(js.keys()) == ["outcome", "steps"] and len(js["steps"]) == 12 and js["outcome"] is None
        st = orc.State()
        for mv, q, kids in js["steps"]:
            assert [c[0] for c in kids] == st.legal_uci() and sum(c[1] for c in kids) == 23
            assert len(kids[0]) == 4
            st.push(mv)


@pytest.mark.gpu
def test_cli_groups_write_the_same_games(tmp_path):
    """--groups 2 (two interleaved handles on two HIP streams) plays the same games as one handle: a game's moves depend
    only on its id and the seed, not on the slot or group it lands in"""
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    common = ["-d", "cuda", "--rollout-num", "16", "-n", "8", "--temperature", "0", "--cpuct", "2", "--temperature-switch", "2",
              "--games", "8", "--blocks", "1", "--channels", "128", "--seed", "11"]
    r1 = _run(*common, "-t", str(a / "trace{}.json"), "--concurrency", "8")
    r2 = _run(*common, "-t", str(b / "trace{}.json"), "--concurrency", "8", "--groups", "2")
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr, r2.stderr)
    for k in range(1, 9):
        assert open(str(a / f"trace{k}.json")).read() == open(str(b / f"trace{k}.json")).read(), k


PLAY = os.path.join(ROOT, "smart-chess-rust_amd", "lib", "sc-play")


def test_play_cli_argument_checks():
    assert os.path.exists(PLAY)
    r = subprocess.run([PLAY, "--white-device", "cuda", "-w", "x.scw"], capture_output=True, text=True)   # default opponent: Stockfish
    assert r.returncode == 2 and "Stockfish" in r.stderr
    r = subprocess.run([PLAY, "-w", "x.scw"], capture_output=True, text=True)
    assert r.returncode == 2 and "white-device" in r.stderr


@pytest.mark.gpu
def test_play_cli_writes_match_traces(tmp_path, orc):
    """`play` flags (src/play.rs:36-84; scripts/leader-board:9-14): two random networks, 6 games, traces with outcome"""
    pat = str(tmp_path / "w_{}.json")
    r = subprocess.run([PLAY, "--white-device", "cuda", "--black-device", "cuda", "--black-type", "nn", "--rollout=12",
                        "--temperature", "0", "--temperature-switch", "0", "--cpuct", "1.5", "-o", pat, "--games", "6",
                        "--blocks", "1", "--channels", "128", "--white-seed", "3", "--black-seed", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Players loaded." in r.stdout and "elo.py input: 6/" in r.stdout
    for k in range(1, 7):
        js = json.load(open(str(tmp_path / f"w_{k}.json")))
        assert list(js.keys()) == ["outcome", "steps"] and 1 <= len(js["steps"]) <= 200
        st = orc.State()
        for mv, q, kids in js["steps"]:
            assert sorted(c[0] for c in kids) == sorted(st.legal_uci()) and sum(c[1] for c in kids) == 11
            st.push(mv)
        assert (js["outcome"] is None) == (st.outcome() is None)
        if js["outcome"] is not None:
            assert js["outcome"] == st.outcome()
