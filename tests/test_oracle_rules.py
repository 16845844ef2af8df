"""Pins the CPU oracle's rules/encoder restatement (oracle/chess.c) against
  * public perft known answers (python-chess itself is absent from the reference tree and the image), and
  * every fixture the reference holds for this path (tests/golden/ref_fixtures.json, ref_sample_games.csv;
    provenance in tools/gen_golden_ref_fixtures.py)."""
import csv
import json
import os

import numpy as np
import pytest

from helpers import san_to_move

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FX = json.load(open(os.path.join(GOLD, "ref_fixtures.json")))

PERFT = {
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1": [20, 400, 8902, 197281],
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1": [48, 2039, 97862],
    "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1": [14, 191, 2812, 43238],
    "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1": [6, 264, 9467],
    "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8": [44, 1486, 62379],
    "r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10": [46, 2079, 89890],
}


@pytest.mark.parametrize("fen", list(PERFT))
def test_perft_known_answers(orc, fen):
    st = orc.State(fen)
    for d, want in enumerate(PERFT[fen], 1):
        assert st.perft(d) == want


def test_reference_move_order_fixtures(orc):
    st = orc.State()
    assert st.legal_uci() == FX["legal_moves_start"]          # notebooks/visualize_mcts.ipynb cell 7
    st.push("g2g3")
    assert st.legal_uci() == FX["legal_moves_after_g2g3"]     # notebooks/verify_model.ipynb cell 12
    assert st.fen() == FX["fen_after_g2g3"]
    st2 = orc.State()
    st2.push("e2e4")
    assert st2.fen() == FX["fen_after_e2e4"]


def test_reference_trace_child_order(orc):
    """children of the first 10 plies of a reference-produced trace = python-chess order at 10 positions"""
    st = orc.State()
    for step in FX["trace_first10"]:
        assert [c[0] for c in step[2]] == st.legal_uci()
        st.push(step[0])


def test_reference_selfplay_game_is_legal(orc):
    st = orc.State()
    for m in FX["selfplay_moves_41"]:
        assert orc.from_uci(m) in st.legal_moves(), m
        st.push(m)
    assert st.ply == 41


def test_reference_sample_games(orc):
    """60 real games (py/validation/sample.csv): every SAN move resolves to exactly one legal move,
    '+' and '#' agree with the oracle's check / checkmate detection."""
    n_moves = 0
    with open(os.path.join(GOLD, "ref_sample_games.csv")) as f:
        for row in csv.DictReader(f):
            st = orc.State()
            toks = row["moves"].split()
            for san in toks:
                m, chk, mate = san_to_move(st, san, orc)
                st.push(m)
                n_moves += 1
                assert st.is_check() == chk, (san, st.fen())
                if mate:
                    oc = st.outcome()
                    assert oc and oc["termination"] == "Checkmate"
                else:
                    assert len(st.legal_moves()) > 0 or not chk
            if row["victory_status"] == "mate":
                oc = st.outcome()
                assert oc["termination"] == "Checkmate" and oc["winner"].lower() == row["winner"]
    assert n_moves > 3000


def test_reference_unit_test_fen(orc):
    """src/chess_fast.rs:89: the only FEN the reference's own unit test checks (legal-move set)."""
    import ctypes as C
    st = orc.State(FX["chess_fast_test_fen"])
    legal = st.legal_uci()
    assert len(legal) == len(set(legal)) > 0
    assert st.perft(2) == sum(_n_after(orc, st, m) for m in st.legal_moves())


def _n_after(orc, st, m):
    st.push(m)
    n = len(st.legal_moves())
    st.pop()
    return n


def test_action_index_known_answers(orc):
    ex = FX["action_index_example"]      # visualize_mcts.ipynb cell 18: 751 -> c2, type 21
    assert orc.move_index("c2d1", 1) is not None
    # type 21 = direction 3 (-1,+1), distance 1 from c2 -> d1
    assert orc.move_index("c2d1", 1) == ex["index"]
    assert orc.move_index("e2e4", 1) == 1 * 584 + 4 * 73 + 1          # N, distance 2
    assert orc.move_index("g1f3", 1) == 0 * 584 + 6 * 73 + 56 + 7     # knight (2,-1)
    assert orc.move_index("e1g1", 1) == 4 * 73 + 2 * 7 + 1            # castling = king two squares east
    assert orc.move_index("e7e5", 0) == orc.move_index("e2e4", 1)     # Black is rotated (rank flip)
    assert orc.move_index("a7a8q", 1) == 6 * 584 + 0 + 0              # queen promotion is a queen move
    assert orc.move_index("a7a8n", 1) == 6 * 584 + 64 + 3             # under-promotion straight, knight
    assert orc.move_index("b7a8r", 1) == 6 * 584 + 73 + 64 + 0 + 2    # capture to the west, rook
    assert orc.move_index("h2g1b", 0) == 6 * 584 + 7 * 73 + 64 + 0 + 1  # Black under-promotion, rotated
    idx = set()
    st = orc.State()
    for m in st.legal_moves():
        idx.add(orc.move_index(m, 1))
    assert len(idx) == 20 and all(0 <= i < 4672 for i in idx)


def test_encoder_known_answers(orc):
    st = orc.State()
    b, m = st.encode()
    assert list(m) == [1, 1, 1, 1, 1, 1, 0]
    assert b.shape == (8, 8, 112) and b[..., 14:].sum() == 0             # no history yet
    assert b[1, :, 0].sum() == 8 and b[6, :, 6].sum() == 8               # pawns: plane 0 mover, 6 opponent
    assert b[0, 4, 5] == 1 and b[7, 4, 11] == 1 and b[0, 3, 4] == 1      # kings, queen
    st.push("e2e4")
    b, m = st.encode()
    assert list(m) == [0, 1, 1, 1, 1, 1, 0]
    # Black to move: board rotated (rank flipped, colours swapped): Black's pawns now on rank index 1 in plane 0
    assert b[1, :, 0].sum() == 8 and b[0, 4, 5] == 1
    assert b[4, 4, 6] == 1                                               # White's e4 pawn seen at flipped rank 4 (7-3)
    assert b[..., 14:28].sum() == 32 and b[..., 28:].sum() == 0          # one history board
    # 50 plies without repetition-relevant state: meta example of notebooks/verify_model.ipynb cell 2
    st = orc.State()
    seq = ["g1f3", "g8f6", "f3g1", "f6g8"]
    for i in range(50):
        st.push(seq[i % 4])
    _, m = st.encode()
    assert m[0] == FX["meta_example"]["meta"][0] and m[1] == FX["meta_example"]["meta"][1]


def test_repetition_and_outcome(orc):
    st = orc.State()
    seq = ["g1f3", "g8f6", "f3g1", "f6g8"]
    st.push(seq[0]); st.push(seq[1]); st.push(seq[2])
    assert not st.is_repetition(2)
    st.push(seq[3])
    assert st.is_repetition(2) and not st.is_repetition(3)
    b, _ = st.encode()
    assert b[0, 0, 12] == 1 and b[0, 0, 13] == 0 and b[0, 0, 14 * 4 + 12] == 0
    assert st.outcome() is None or st.outcome()["termination"] == "ThreefoldRepetition"
    for m in seq:
        st.push(m)
    assert st.is_repetition(3)
    assert st.outcome()["termination"] == "ThreefoldRepetition"
    b, _ = st.encode()
    assert b[0, 0, 12] == 1 and b[0, 0, 13] == 1
    # fool's mate
    st = orc.State()
    for m in ["f2f3", "e7e5", "g2g4", "d8h4"]:
        st.push(m)
    assert st.outcome() == {"termination": "Checkmate", "winner": "Black"}
    assert orc.State("8/8/8/8/8/5k2/8/5K2 w - - 0 1").outcome()["termination"] == "InsufficientMaterial"
    assert orc.State("7k/5Q2/6K1/8/8/8/8/8 b - - 0 1").outcome()["termination"] == "Stalemate"
    assert orc.State("7k/8/6K1/8/8/8/8/R7 w - - 100 80").outcome()["termination"] == "FiftyMoves"
    assert orc.State("7k/8/6K1/8/8/8/8/R7 w - - 150 100").outcome()["termination"] == "SeventyfiveMoves"
