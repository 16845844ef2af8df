"""CPU: the oracle's literal restatement of libsmartchess.chess_encode_steps (reference src/lib.rs:46-128).

No file of the reference holds an output of that function (notebooks/verify_dataset.ipynb only constructs the
dataset), so its parity is UNPINNED by fixtures; what pins it here are values derivable by hand from the reference
source: the start-position planes and meta, Board::rotate()'s meta, the action indices recorded in the reference's
notebooks (e2e4 -> 877, g1f3 -> 501), the dist arithmetic, and the structural identities the code implies.
"""
import random

import numpy as np
import pytest

from helpers import random_games


def _steps_from_moves(orc, moves, rnd):
    st = orc.State()
    steps = []
    for m in moves:
        lm = st.legal_moves()
        cnt = [rnd.randint(0, 60) for _ in lm]
        order = list(range(len(lm)))
        rnd.shuffle(order)          # the reference collects the children into a HashMap: order must not matter
        steps.append((m, [(lm[i], cnt[i]) for i in order]))
        st.push(m)
    return steps


def test_start_position_by_hand(orc):
    st = orc.State()
    lm = st.legal_moves()
    e2e4 = orc.from_uci("e2e4")
    steps = [(e2e4, [(m, 3 if m == e2e4 else 1) for m in lm])]
    for mirror in (False, True):
        rc, boards, meta, dist, idx = orc.encode_steps(steps, mirror)
        assert rc == 0
        b = boards[0]
        # White pawns on rank index 1 -> plane 0, White king e1 -> plane 5, Black pawns rank 6 -> plane 6 (chess.rs:623-650)
        assert b[1, :, 0].tolist() == [1] * 8 and b[0, 4, 5] == 1 and b[6, :, 6].tolist() == [1] * 8 and b[7, 4, 11] == 1
        assert b[:, :, 12:].sum() == 0 and b[:, :, 14:].sum() == 0      # no repetition, no history yet
        # encode_meta (chess.rs:652-662); rotate(): turn flipped, fullmove + 1 because White was to move (:594-621)
        assert meta[0].tolist() == ([0, 2, 1, 1, 1, 1, 0] if mirror else [1, 1, 1, 1, 1, 1, 0])
        # action index of e2e4 = 1*584 + 4*73 + (0*7 + 1) = 877; g1f3 = 6*73 + 56 + 7 = 501
        total = np.float32(3 + 19)
        assert dist[0][877] == np.float32(3) / (total + np.float32(1e-5))
        assert dist[0][501] == np.float32(1) / (total + np.float32(1e-5))
        assert np.count_nonzero(dist[0]) == 20
        assert sorted(idx[0].tolist()) == sorted(int(orc.move_index(m, 1)) for m in lm)
        assert idx[0].tolist() == [int(orc.move_index(m, 1)) for m in lm]   # python-chess generation order


def test_identities_on_random_games(orc):
    rnd = random.Random(5)
    for moves, _ in random_games(orc, 12, 70, seed=21):
        if not moves:
            continue
        steps = _steps_from_moves(orc, moves, rnd)
        rc, b0, m0, d0, i0 = orc.encode_steps(steps, False)
        rc1, b1, m1, d1, i1 = orc.encode_steps(steps, True)
        assert rc == 0 and rc1 == 0
        # the mirror rotates every stored board at push time and once more at view time: planes and dist are unchanged
        assert (b0 == b1).all() and (d0 == d1).all() and all((x == y).all() for x, y in zip(i0, i1))
        assert (m1[:, 0] == 1 - m0[:, 0]).all() and (m1[:, 6] == m0[:, 6]).all()
        assert (m1[:, 1] == m0[:, 1] + m0[:, 0]).all()
        assert (m1[:, 2:4] == m0[:, 4:6]).all() and (m1[:, 4:6] == m0[:, 2:4]).all()
        # without mirror the planes / meta are the hot path's _encode of the same position (chess.rs:845-877)
        st = orc.State()
        for k, s in enumerate(steps):
            bb, mm = st.encode()
            assert (bb == b0[k]).all() and (mm == m0[k]).all()
            tot = sum(c for _, c in s[1])
            assert abs(float(d0[k].sum()) - tot / (tot + 1e-5)) < 1e-5
            st.push(s[0])


def test_reference_panics_are_reported(orc):
    st = orc.State()
    lm = st.legal_moves()
    good = [(m, 1) for m in lm]
    e2e4, e7e5 = orc.from_uci("e2e4"), orc.from_uci("e7e5")
    assert orc.encode_steps([(e2e4, good[:-1])], False)[0] == 1000            # a legal move is missing
    assert orc.encode_steps([(e2e4, good + [(e7e5, 1)])], False)[0] == 1000   # a child that is not legal
    assert orc.encode_steps([(e7e5, good)], False)[0] == -1                   # the played move is not legal
    st.push(e2e4)
    good2 = [(m, 1) for m in st.legal_moves()]
    assert orc.encode_steps([(e2e4, good), (e2e4, good2)], False)[0] == -2
