"""Finds a short move list from the start position to a position with MANY legal moves (default: more than 64, the
width of a wavefront: the root of a search from there takes the four-round arg-max of the descent).  Beam search on the
CPU oracle's move generator; test tooling only -- the result is pasted into tests/test_gpu_parity2.py (WIDE).

    python tools/find_wide_position.py [target_legal_moves=80] [max_plies=60]
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as orc
# beam search over move sequences maximizing White's legal move count (Black plays quiet shuffles)
def score(st):
    return len(st.legal_moves())
best = None
random.seed(1)
beam = [([], orc.State())]
TARGET = int(sys.argv[1]) if len(sys.argv) > 1 else 80
MAXPLY = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for ply in range(MAXPLY):
    cand = []
    for mv, st in beam:
        lm = st.legal_moves()
        if not lm: continue
        random.shuffle(lm)
        for m in lm[:40]:
            s2 = st.copy(); s2.push(m)
            if s2.outcome() is not None: continue
            # evaluate: white mobility (if white to move now count, else count after a null-ish estimate)
            if s2.turn == 1:
                sc = score(s2)
            else:
                # black to move: white's mobility unknown; use previous white mobility proxy: count white moves by trying each black reply? cheap proxy: -black mobility small
                sc = None
            cand.append((mv + [m], s2, sc))
    # for black-to-move candidates compute proxy lazily: best white mobility over few black replies
    scored = []
    for mv, s2, sc in cand:
        if sc is None:
            lm = s2.legal_moves()
            random.shuffle(lm)
            sc = 0
            for m in lm[:3]:
                s3 = s2.copy(); s3.push(m)
                sc = max(sc, score(s3))
        scored.append((sc, mv, s2))
    scored.sort(key=lambda x: -x[0])
    beam = [(mv, s) for sc, mv, s in scored[:30]]
    top = scored[0]
    if beam[0][1].turn == 1 and (best is None or top[0] > best[0]):
        best = (top[0], top[1])
        print(ply, top[0], ' '.join(orc.uci(m) for m in top[1]), flush=True)
    if best and best[0] >= TARGET: break
