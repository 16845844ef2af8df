// src/backends/hip.rs -- the reference-side binding of libsc_engine.so (include/sc_engine.h).
//
// What a maintainer of pierric/smart-chess-rust adds to the tree:
//   * this file as `src/backends/hip.rs` and `pub mod hip;` in `src/backends/mod.rs`
//   * `build.rs`:  println!("cargo:rustc-link-search=native=<repo>/smart-chess-rust_amd/lib");
//                  println!("cargo:rustc-link-lib=dylib=sc_engine");
//   * one arm in the backend selection of `src/main.rs:83-128` / `src/play.rs:347-384`:
//         Some("scw") => Box::new(backends::hip::ChessHip::new(&args.checkpoint, 0, false)),
//
// No new crates.  Level 1 (`ChessHip`) implements `trait Game<BoardState>` (src/game.rs:3-15) exactly like ChessTS
// (src/backends/torch.rs:34-53), so `mcts::mcts`, `main.rs` and `play.rs` stay untouched.  Level 2 (`SelfPlay`) is the
// batched driver: the whole loop of src/main.rs:155-238 for many concurrent games on one GPU.
//
// Rust is not installed in the image this engine is built in, so this file is NOT compiled there.  Its `#[repr(C)]`
// structs are checked field by field (name, order, type, size, offset) against include/sc_engine.h and against the ctypes
// structs of scamd/binding.py by tests/test_abi.py::test_rust_binding_structs_match_the_header, and the header's layout
// is pinned by static_asserts in csrc/engine.hip.
#![allow(dead_code)]
use crate::chess::{BoardState, Color, Move, PieceType, Square, Step};
use crate::game::Game;
use crate::mcts::ArcRefNode;
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int};

pub const SC_MAX_MOVES: usize = 224;
pub const SC_POLICY_SIZE: usize = 4672;
pub const SC_BOARD_BYTES: usize = 7168;
pub const SC_PREC_BF16: i32 = 0;
pub const SC_PREC_FP8: i32 = 1;
pub const SC_EVAL_NET: i32 = 0;

// ---------------------------------------------------------------------------------------------- C structs
#[repr(C)]
pub struct ScNetConfig {
    pub n_res_blocks: i32,
    pub channels: i32,
    pub seed: u64,
    pub precision: i32,
    pub reserved: i32,
}

#[repr(C)]
pub struct ScSelfplayConfig {
    pub n_slots: i32,
    pub n_games: i32,
    pub rollout_num: i32,
    pub num_steps: i32,
    pub cpuct: f32,
    pub temperature: f32,
    pub temperature_switch: i32,
    pub epsilon: f32,
    pub with_noise: i32,
    pub outcome_gate: i32,
    pub evaluator: i32,
    pub external_noise: i32,
    pub seed: u64,
    pub first_game_id: u64,
    pub trace_capacity: i32,
    pub own_stream: i32,
    pub tie_random: i32,
    pub trace_hold: i32,
    pub rollout_factor: f32,
}

#[repr(C)]
pub struct ScSelfplayStats {
    pub sims_done: i64,
    pub nn_evals: i64,
    pub games_finished: i32,
    pub games_active: i32,
    pub error_flags: i32,
    pub plies_done: i32,
}

#[repr(C)]
pub struct ScTraceInfo {
    pub n_steps: i32,
    pub n_children_total: i32,
    pub has_outcome: i32,
    pub termination: i32,
    pub winner: i32,
    pub game_id: u64,
}

#[repr(C)]
pub struct ScEngine {
    _p: [u8; 0],
}
#[repr(C)]
pub struct ScSelfplay {
    _p: [u8; 0],
}

extern "C" {
    fn sc_last_error() -> *const c_char;
    fn sc_device_count() -> c_int;
    fn sc_engine_create(cfg: *const ScNetConfig, weights_path: *const c_char, device_id: c_int, out: *mut *mut ScEngine) -> c_int;
    fn sc_engine_destroy(e: *mut ScEngine);
    // rules + _encode on the GPU (replaces BoardState::legal_moves / outcome + _encode: chess.rs:719-759, 845-877)
    fn sc_encode_positions(e: *mut ScEngine, device_id: c_int, n: c_int, moves: *const u16, move_off: *const u32,
                           boards: *mut i8, meta: *mut i32, legal_moves: *mut u16, legal_idx: *mut u16, n_legal: *mut i32,
                           outcome: *mut i32) -> c_int;
    // forward + gather + exp + renormalise (replaces torch.rs:115-138)
    fn sc_predict_batch(e: *mut ScEngine, n: c_int, boards: *const i8, meta: *const i32, legal_idx: *const u16,
                        legal_off: *const u32, priors: *mut f32, value: *mut f32) -> c_int;
    // the same with post_process_distr(argmax = true) (chess.rs:880-889): one-hot at the LAST maximal prior
    fn sc_predict_batch_argmax(e: *mut ScEngine, n: c_int, boards: *const i8, meta: *const i32, legal_idx: *const u16,
                               legal_off: *const u32, priors: *mut f32, value: *mut f32) -> c_int;
    // L-search
    fn sc_selfplay_create(e: *mut ScEngine, device_id: c_int, cfg: *const ScSelfplayConfig, out: *mut *mut ScSelfplay) -> c_int;
    fn sc_selfplay_destroy(sp: *mut ScSelfplay);
    fn sc_selfplay_enqueue_sims(sp: *mut ScSelfplay, n: c_int) -> c_int;
    fn sc_selfplay_run(sp: *mut ScSelfplay, max_sim_steps: i64) -> c_int;
    fn sc_selfplay_poll(sp: *mut ScSelfplay, finished_games: *mut i32, cap: c_int) -> c_int;
    fn sc_selfplay_get_stats(sp: *mut ScSelfplay, out: *mut ScSelfplayStats) -> c_int;
    fn sc_selfplay_get_trace(sp: *mut ScSelfplay, game: c_int, info: *mut ScTraceInfo, step_move: *mut u16, step_q: *mut f32,
                             child_off: *mut i32, child_move: *mut u16, child_n: *mut i32, child_q: *mut f32,
                             child_uct: *mut f32) -> c_int;
    fn sc_selfplay_write_trace_json(sp: *mut ScSelfplay, game: c_int, path: *const c_char) -> c_int;
    // NNPlayer::bestmove's search as one call (src/play.rs:241-252)
    fn sc_search(e: *mut ScEngine, moves: *const u16, n_moves: c_int, rollout: c_int, cpuct: f32, with_noise: c_int, seed: u64,
                 cap: c_int, child_move: *mut u16, child_n: *mut i32, child_q: *mut f32, child_prior: *mut f32,
                 root_q: *mut f32) -> c_int;
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(sc_last_error()).to_string_lossy().into_owned() }
}
fn check(rc: c_int, what: &str) {
    // the reference unwrap()s / panics on backend errors (torch.rs:27-31, 100, 111); the library itself never aborts
    if rc != 0 {
        panic!("{}: libsc_engine error {}: {}", what, rc, last_error());
    }
}

// include/sc_engine.h: move = from | to << 6 | promo << 12, squares a1 = 0 .. h8 = 63, promo = python-chess piece type
pub fn to_u16(m: &Move) -> u16 {
    let sq = |s: &Square| (s.rank * 8 + s.file) as u16;
    let p = match m.promotion {
        None => 0,
        Some(pt) => pt as u16, // PieceType: Knight = 2 .. Queen = 5 (chess.rs:52-60)
    };
    sq(&m.from) | (sq(&m.to) << 6) | (p << 12)
}
pub fn from_u16(v: u16) -> Move {
    let sq = |x: u16| Square { rank: (x >> 3) as i32, file: (x & 7) as i32 };
    let p = (v >> 12) & 7;
    Move {
        from: sq(v & 63),
        to: sq((v >> 6) & 63),
        promotion: if p == 0 { None } else { Some(PieceType::from(p as i32)) },
        drop: None,
    }
}

// ---------------------------------------------------------------------------------------------- Level 1
pub struct ChessHip {
    engine: *mut ScEngine,
}

impl ChessHip {
    /// `checkpoint`: an `.scw` blob written by tools/ckpt_to_scw.py from a reference `.ckpt` (the blob carries depth, width
    /// and -- for the fp8 export -- precision).  Replaces the backend construction of src/main.rs:83-128.
    pub fn new(checkpoint: &str, device_id: i32, fp8: bool) -> Self {
        let cfg = ScNetConfig {
            n_res_blocks: 0, // ignored when a blob is given
            channels: 256,
            seed: 0,
            precision: if fp8 { SC_PREC_FP8 } else { SC_PREC_BF16 },
            reserved: 0,
        };
        let path = CString::new(checkpoint).unwrap();
        let mut engine: *mut ScEngine = std::ptr::null_mut();
        check(unsafe { sc_engine_create(&cfg, path.as_ptr(), device_id as c_int, &mut engine) }, "sc_engine_create");
        ChessHip { engine }
    }

    /// Random-initialised network of the given shape (benchmarks, plumbing): `load_model` of py/module.py:184-201.
    pub fn with_random_init(n_res_blocks: i32, channels: i32, seed: u64, device_id: i32, fp8: bool) -> Self {
        let cfg = ScNetConfig { n_res_blocks, channels, seed, precision: if fp8 { SC_PREC_FP8 } else { SC_PREC_BF16 }, reserved: 0 };
        let mut engine: *mut ScEngine = std::ptr::null_mut();
        check(unsafe { sc_engine_create(&cfg, std::ptr::null(), device_id as c_int, &mut engine) }, "sc_engine_create");
        ChessHip { engine }
    }

    pub fn raw(&self) -> *mut ScEngine {
        self.engine
    }
}

impl Drop for ChessHip {
    fn drop(&mut self) {
        unsafe { sc_engine_destroy(self.engine) };
    }
}

impl Game<BoardState> for ChessHip {
    fn predict(&self, node: &ArcRefNode<Step>, state: &BoardState, argmax: bool) -> (Vec<Step>, Vec<f32>, f32) {
        let turn = node.borrow().step.1;
        // what _encode reconstructs from the tree's parent chain and the board's move stack (chess.rs:845-867)
        let moves: Vec<u16> = state.move_stack().iter().map(to_u16).collect(); // chess.rs:761-769
        let off = [0u32, moves.len() as u32];
        let mut boards = vec![0i8; SC_BOARD_BYTES];
        let mut meta = [0i32; 7];
        let mut lm = vec![0u16; SC_MAX_MOVES];
        let mut li = vec![0u16; SC_MAX_MOVES];
        let mut nl = 0i32;
        let mut oc = [0i32; 4];
        check(
            unsafe {
                sc_encode_positions(self.engine, 0, 1, moves.as_ptr(), off.as_ptr(), boards.as_mut_ptr(), meta.as_mut_ptr(),
                                    lm.as_mut_ptr(), li.as_mut_ptr(), &mut nl, oc.as_mut_ptr())
            },
            "sc_encode_positions",
        );
        assert!(oc[3] == 0, "illegal move in the move stack");
        assert!((meta[0] == 1) == (turn == Color::White)); // torch.rs:109-111
        if nl == 0 {
            // torch.rs:98-106: +1 White won / -1 Black won / 0
            return (vec![], vec![], match oc[1] { 1 => 1.0, 0 => -1.0, _ => 0.0 });
        }
        let loff = [0u32, nl as u32];
        let mut pri = vec![0f32; nl as usize];
        let mut val = 0f32;
        let rc = unsafe {
            if argmax {
                sc_predict_batch_argmax(self.engine, 1, boards.as_ptr(), meta.as_ptr(), li.as_ptr(), loff.as_ptr(), pri.as_mut_ptr(), &mut val)
            } else {
                sc_predict_batch(self.engine, 1, boards.as_ptr(), meta.as_ptr(), li.as_ptr(), loff.as_ptr(), pri.as_mut_ptr(), &mut val)
            }
        };
        check(rc, "sc_predict_batch");
        if !val.is_finite() {
            println!("Warning: value is not finite: {}", val); // torch.rs:129-135
        }
        let steps = lm[..nl as usize].iter().map(|&m| Step(Some(from_u16(m)), !turn)).collect(); // torch.rs:140-143
        (steps, pri, val)
    }

    fn reverse_q(&self, node: &ArcRefNode<Step>) -> bool {
        node.borrow().step.1 == Color::Black // torch.rs:49-52
    }
}

// ---------------------------------------------------------------------------------------------- Level 2
/// One handle = many concurrent games on one GPU (`selfplay`'s loop, src/main.rs:155-238, batched).  One handle per GPU,
/// one host thread per handle; handles of different GPUs are independent (how games shard over a node).
pub struct SelfPlay<'a> {
    handle: *mut ScSelfplay,
    _engine: &'a ChessHip,
}

impl<'a> SelfPlay<'a> {
    /// The reference's flags (src/main.rs:25-60) with the canonical self-play setting of README.md:39 as defaults.
    pub fn default_config(n_slots: i32, n_games: i32) -> ScSelfplayConfig {
        ScSelfplayConfig {
            n_slots,
            n_games,
            rollout_num: 180,
            num_steps: 150,
            cpuct: 2.5,
            temperature: 0.0,
            temperature_switch: 4,
            epsilon: 0.15,
            with_noise: 1,
            outcome_gate: 100,
            evaluator: SC_EVAL_NET,
            external_noise: 0,
            seed: 0,
            first_game_id: 0,
            trace_capacity: 0,
            own_stream: 0,
            tie_random: 0,
            trace_hold: 0,
            rollout_factor: 0.0,
        }
    }

    pub fn new(engine: &'a ChessHip, cfg: &ScSelfplayConfig) -> Self {
        let mut handle: *mut ScSelfplay = std::ptr::null_mut();
        check(unsafe { sc_selfplay_create(engine.raw(), 0, cfg, &mut handle) }, "sc_selfplay_create");
        SelfPlay { handle, _engine: engine }
    }

    /// Plays every game; afterwards `write_trace(g, path)` writes the reference's trace JSON (src/trace.rs:23-32).
    pub fn run(&mut self) -> ScSelfplayStats {
        check(unsafe { sc_selfplay_run(self.handle, 0) }, "sc_selfplay_run");
        self.stats()
    }

    pub fn stats(&mut self) -> ScSelfplayStats {
        let mut st = ScSelfplayStats { sims_done: 0, nn_evals: 0, games_finished: 0, games_active: 0, error_flags: 0, plies_done: 0 };
        check(unsafe { sc_selfplay_get_stats(self.handle, &mut st) }, "sc_selfplay_get_stats");
        st
    }

    pub fn write_trace(&mut self, game: i32, path: &str) {
        let p = CString::new(path).unwrap();
        check(unsafe { sc_selfplay_write_trace_json(self.handle, game as c_int, p.as_ptr()) }, "sc_selfplay_write_trace_json");
    }
}

impl<'a> Drop for SelfPlay<'a> {
    fn drop(&mut self) {
        unsafe { sc_selfplay_destroy(self.handle) };
    }
}
