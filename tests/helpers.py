"""Test helpers (SAN parsing on top of the oracle's legal-move list)."""
import random

PT = {"N": 2, "B": 3, "R": 4, "Q": 5, "K": 6}


def san_to_move(st, san, orc):
    """Resolve a SAN token against the oracle's legal moves; returns (move, gives_check_flag, mate_flag)."""
    s = san.strip()
    mate = s.endswith("#")
    check = s.endswith("+") or mate
    s = s.rstrip("+#")
    legal = st.legal_moves()
    turn = st.turn
    if s in ("O-O", "O-O-O"):
        base = 0 if turn else 56
        to = base + (6 if s == "O-O" else 2)
        cands = [m for m in legal if (m & 63) == base + 4 and ((m >> 6) & 63) == to and abs(st.piece_at(base + 4)) == 6]
        assert len(cands) == 1, (san, st.fen())
        return cands[0], check, mate
    promo = 0
    if "=" in s:
        s, pr = s.split("=")
        promo = PT[pr]
    piece = 1
    if s[0] in PT:
        piece = PT[s[0]]
        s = s[1:]
    s = s.replace("x", "")
    to = (ord(s[-1]) - 49) * 8 + (ord(s[-2]) - 97)
    dis = s[:-2]
    cands = []
    for m in legal:
        f, t, p = m & 63, (m >> 6) & 63, (m >> 12) & 7
        if t != to or p != promo or abs(st.piece_at(f)) != piece:
            continue
        ok = True
        for ch in dis:
            if ch in "abcdefgh" and (f & 7) != ord(ch) - 97:
                ok = False
            if ch in "12345678" and (f >> 3) != ord(ch) - 49:
                ok = False
        if ok:
            cands.append(m)
    assert len(cands) == 1, (san, st.fen(), [orc.uci(c) for c in cands])
    return cands[0], check, mate


def random_games(orc, n, maxlen, seed):
    """[(moves, State)] with a bias towards knight shuffles (repetitions) in every third game."""
    rnd = random.Random(seed)
    games = []
    for g in range(n):
        st = orc.State()
        mv = []
        for _ in range(rnd.randint(0, maxlen)):
            lm = st.legal_moves()
            if not lm:
                break
            if g % 3 == 0:
                pref = [x for x in lm if abs(st.piece_at(x & 63)) == 2]
                m = rnd.choice(pref) if pref and rnd.random() < 0.8 else rnd.choice(lm)
            else:
                m = rnd.choice(lm)
            st.push(m)
            mv.append(m)
        games.append((mv, st))
    return games
