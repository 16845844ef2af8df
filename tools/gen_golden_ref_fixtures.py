"""Collects the golden vectors the REFERENCE already holds for this path into tests/golden/
(run in the build container only; outputs are data, not source):

  * py/validation/sample.csv           -> tests/golden/ref_sample_games.csv (60 real games in SAN:
                                           every move must be legal and uniquely resolvable, '+'/'#' must agree)
  * notebooks/*.ipynb recorded outputs -> tests/golden/ref_fixtures.json:
      legal-move order at the start position (visualize_mcts.ipynb cell 7) and after 1.g3
      (verify_model.ipynb cell 12), FENs printed by python-chess, a 41-ply self-play move list,
      the trace excerpt with serde_json float formatting (verify_model.ipynb cell 8), the decoded
      action index 751 (visualize_mcts.ipynb cell 18), a recorded meta vector, and the FEN of the
      reference's only unit test (src/chess_fast.rs:89).
"""
import ast
import json
import os
import re
import shutil

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def cell_out(nb, idx):
    d = json.load(open(os.path.join(REF, "notebooks", nb + ".ipynb")))
    c = d["cells"][idx]
    outs = []
    for o in c.get("outputs", []):
        if "text" in o:
            outs.append("".join(o["text"]))
        elif "data" in o and "text/plain" in o["data"]:
            outs.append("".join(o["data"]["text/plain"]))
    return "".join(c["source"]), "\n".join(outs)


def main():
    shutil.copyfile(os.path.join(REF, "py", "validation", "sample.csv"), os.path.join(OUT, "ref_sample_games.csv"))
    fx = {}
    _, o = cell_out("visualize_mcts", 7)
    fx["legal_moves_start"] = re.findall(r"from_uci\('([a-h1-8qrbn]+)'\)", o)
    _, o = cell_out("verify_model", 12)
    fx["legal_moves_after_g2g3"] = re.findall(r"from_uci\('([a-h1-8qrbn]+)'\)", o)
    _, o = cell_out("verify_model", 11)
    fx["fen_after_g2g3"] = re.search(r"Board\('([^']+)'\)", o).group(1)
    _, o = cell_out("visualize_mcts", 19)
    fx["fen_after_e2e4"] = re.search(r"Board\('([^']+)'\)", o).group(1)
    _, o = cell_out("visualize_mcts", 33)
    moves = ast.literal_eval(o.strip())
    _, o = cell_out("visualize_mcts", 34)
    moves.append(ast.literal_eval(o.strip()))
    fx["selfplay_moves_41"] = moves
    src, o = cell_out("verify_model", 8)
    # steps["steps"][:10] of a reference-produced trace (older 3-tuple children: move, N, Q_sum):
    # child order = python-chess legal-move order at 10 consecutive positions of a real game,
    # floats as serde_json printed them (f32 widened to f64, shortest repr)
    steps10 = ast.literal_eval(o.strip())
    assert len(steps10) == 10 and steps10[0][0] == "g2g3"
    fx["trace_first10"] = steps10
    fx["trace_first10_text"] = o.strip()
    _, o = cell_out("verify_model", 7)
    fx["trace_outcome"] = ast.literal_eval(o.strip())
    src, o = cell_out("visualize_mcts", 18)
    fx["action_index_example"] = {"index": 751, "square": "c2", "type": 21}
    assert "751" in src and "('c2', 21)" in o
    _, o = cell_out("verify_model", 2)
    fx["meta_example"] = {"ply": 50, "meta": [1, 26, 0, 0, 0, 0, 0]}
    assert "1, 26,  0,  0,  0,  0,  0" in o
    fx["chess_fast_test_fen"] = "1k1r4/1r5p/p4n1P/1ppP1P2/PP6/4PP1b/3B4/R1N1K3 b - - 0 39"
    json.dump(fx, open(os.path.join(OUT, "ref_fixtures.json"), "w"), indent=1)
    print({k: (len(v) if hasattr(v, "__len__") else v) for k, v in fx.items()})


if __name__ == "__main__":
    main()
