"""CPU unit tests of the ENGINE's rules source (csrc/chess_rules.hpp, chess_history.hpp) compiled for the
host (lib/libsc_rules_host.so): perft known answers and cross-checks against the oracle, which uses a
different method (mailbox + make/test) -- so the two implementations pin each other."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import random_games
from test_oracle_rules import PERFT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "smart-chess-rust_amd")


@pytest.fixture(scope="module")
def H():
    so = os.path.join(PKG, "lib", "libsc_rules_host.so")
    src = os.path.join(PKG, "csrc", "rules_host_api.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(PKG, "csrc", f))
                                                           for f in ("rules_host_api.cpp", "chess_rules.hpp", "chess_history.hpp")):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    L = C.CDLL(so)
    L.sct_new.restype = C.c_void_p
    L.sct_perft.restype = C.c_uint64
    L.sct_pos_hash.restype = C.c_uint64
    L.sct_rng.restype = C.c_uint64
    L.sct_key.restype = C.c_uint64
    L.sct_key_full.restype = C.c_uint64
    L.sct_key.argtypes = [C.c_void_p]
    L.sct_key_full.argtypes = [C.c_void_p]
    L.sct_rng.argtypes = [C.c_uint64] * 5
    for n in ("sct_free", "sct_reset", "sct_push", "sct_pop", "sct_encode", "sct_synth_eval"):
        getattr(L, n).restype = None
    L.sct_set_fen.argtypes = [C.c_void_p, C.c_char_p]
    L.sct_push.argtypes = [C.c_void_p, C.c_uint16]
    L.sct_move_index.argtypes = [C.c_uint16, C.c_int]
    for n in ("sct_free", "sct_reset", "sct_pop", "sct_pos_hash", "sct_turn"):
        getattr(L, n).argtypes = [C.c_void_p]
    L.sct_legal_moves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.sct_perft.argtypes = [C.c_void_p, C.c_int]
    L.sct_is_repetition.argtypes = [C.c_void_p, C.c_int]
    L.sct_outcome.argtypes = [C.c_void_p, C.c_void_p]
    L.sct_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.sct_synth_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return L


DEEP = {
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1": (4, 4085603),
    "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1": (5, 674624),
    "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1": (4, 422333),
    "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8": (4, 2103487),
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1": (5, 4865609),
}


def test_perft_known_answers(H):
    s = H.sct_new()
    for fen, exp in PERFT.items():
        assert H.sct_set_fen(s, fen.encode()) == 0
        for d, want in enumerate(exp, 1):
            assert H.sct_perft(s, d) == want, (fen, d)
    for fen, (d, want) in DEEP.items():
        assert H.sct_set_fen(s, fen.encode()) == 0
        assert H.sct_perft(s, d) == want, fen
    H.sct_free(s)


def test_cross_check_against_oracle(H, orc):
    """move order, check flag, planes, meta, repetition, outcome(claim_draw), hashes, action indices
    along 120 random games (~25k positions)"""
    import random
    rnd = random.Random(5)
    s = H.sct_new()
    buf = (C.c_uint16 * 256)()
    chk = C.c_int(0)
    seen_terms = set()
    n_pos = 0
    for g in range(120):
        H.sct_reset(s)
        o = orc.State()
        for ply in range(300):
            n = H.sct_legal_moves(s, buf, C.byref(chk))
            mine = [buf[i] for i in range(n)]
            assert mine == o.legal_moves(), o.fen()
            assert bool(chk.value) == o.is_check()
            b = np.zeros((8, 8, 112), np.int8)
            m = np.zeros(7, np.int32)
            H.sct_encode(s, b.ctypes.data, m.ctypes.data)
            ob, om = o.encode()
            assert np.array_equal(b, ob) and np.array_equal(m, om), o.fen()
            for c in (2, 3, 5):
                assert bool(H.sct_is_repetition(s, c)) == o.is_repetition(c)
            w = C.c_int(0)
            t = H.sct_outcome(s, C.byref(w))
            oo = o.outcome()
            assert (t == 0) == (oo is None), (o.fen(), t, oo)
            if oo:
                assert orc.TERMINATION[t] == oo["termination"]
                assert {1: "White", 0: "Black", -1: None}[w.value] == oo["winner"]
                seen_terms.add(oo["termination"])
            assert H.sct_pos_hash(s) == o.pos_hash()
            assert H.sct_key(s) == H.sct_key_full(s)      # incremental transposition key == full recompute
            for mv in mine:
                assert H.sct_move_index(mv, o.turn) == orc.move_index(mv, o.turn) >= 0
            n_pos += 1
            if n == 0 or (oo and oo["termination"] in ("SeventyfiveMoves", "FivefoldRepetition", "InsufficientMaterial")):
                break
            if g % 3 == 0:
                pref = [x for x in mine if abs(o.piece_at(x & 63)) == 2]
                mv = rnd.choice(pref) if pref and rnd.random() < 0.8 else rnd.choice(mine)
            else:
                mv = rnd.choice(mine)
            H.sct_push(s, mv)
            o.push(mv)
    H.sct_free(s)
    assert n_pos > 15000
    assert {"Checkmate", "ThreefoldRepetition"} <= seen_terms and len(seen_terms) >= 3


def test_clock_and_repetition_terminations(H, orc):
    s = H.sct_new()
    w = C.c_int(0)
    for fen, want in [("7k/8/6K1/8/8/8/8/R7 w - - 100 80", "FiftyMoves"), ("7k/8/6K1/8/8/8/8/R7 w - - 150 100", "SeventyfiveMoves"),
                      ("7k/8/6K1/8/8/8/8/R7 w - - 99 80", "FiftyMoves"), ("8/8/8/8/8/5k2/8/5K2 w - - 0 1", "InsufficientMaterial"),
                      ("7k/5Q2/6K1/8/8/8/8/8 b - - 0 1", "Stalemate")]:
        assert H.sct_set_fen(s, fen.encode()) == 0
        assert orc.TERMINATION[H.sct_outcome(s, C.byref(w))] == want == orc.State(fen).outcome()["termination"]
    H.sct_reset(s)
    o = orc.State()
    seq = ["g1f3", "g8f6", "f3g1", "f6g8"]
    for i in range(16):
        H.sct_push(s, orc.from_uci(seq[i % 4]))
        o.push(seq[i % 4])
        assert orc.TERMINATION.get(H.sct_outcome(s, C.byref(w))) == (o.outcome() or {}).get("termination")
    assert o.outcome()["termination"] == "FivefoldRepetition"
    H.sct_free(s)


def test_synth_evaluator_and_rng_match_oracle(H, orc):
    import ctypes
    L = orc.lib()
    s = H.sct_new()
    o = orc.State()
    for mv in ["e2e4", "c7c5", "g1f3"]:
        H.sct_push(s, orc.from_uci(mv))
        o.push(mv)
    pri = np.zeros(256, np.float32)
    val = np.zeros(1, np.float32)
    H.sct_synth_eval(s, pri.ctypes.data, val.ctypes.data)
    lm = o.legal_moves()
    legal = (C.c_uint16 * len(lm))(*lm)
    idx = (C.c_int * len(lm))()
    opri = (C.c_float * len(lm))()
    oval = C.c_float(0)
    L.orc_eval_synth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_eval_synth.restype = None
    L.orc_eval_synth(None, o.h, len(lm), legal, idx, opri, C.byref(oval))
    assert np.array_equal(pri[:len(lm)], np.array(list(opri), np.float32))
    assert val[0] == oval.value and abs(pri[:len(lm)].sum() - 1) < 1e-5
    for args in [(0, 0, 0, 1, 0), (123, 7, 33, 3, 99), (2 ** 63, 5, 1, 2, 0)]:
        assert H.sct_rng(*args) == L.orc_rng(*args)
    H.sct_free(s)
