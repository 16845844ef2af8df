"""scamd -- Python (ctypes) host side of libsc_engine.so, the MI355X-native engine for the
MCTS + NN self-play hot path of pierric/smart-chess-rust.

The classes mirror the reference's own interfaces for this path so that tests read like the
reference's usage:

  ChessHip.predict / reverse_q   <->  trait Game<S> (src/game.rs:3-15) as implemented by
                                       ChessTS / ChessOnnx (src/backends/torch.rs:89-146)
  SelfPlay                       <->  bin `selfplay` (src/main.rs:155-238): same flag names
  encode_positions               <->  BoardState + _encode (src/chess.rs:665-877)
  encode_steps                   <->  libsmartchess.chess_encode_steps (src/lib.rs:46-128), the trace -> training-tensor step

There is NO CPU fallback: importing works anywhere (so the C ABI can be checked), but every
compute entry point raises EngineError when the HIP library or a GPU is missing.
"""
from .binding import (ChessHip, Engine, EngineError, Play, SelfPlay, encode_move, encode_positions, encode_steps, encode_steps_batch,  # noqa: F401
                      enqueue_interleaved, elo, find_max, lib, lib_path, play_match, runtime_flags, search,
                      move_uci, uci_move, write_trace_json, TERMINATION)
from . import binding  # noqa: F401
