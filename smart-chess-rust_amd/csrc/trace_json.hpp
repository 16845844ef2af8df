// trace_json.hpp -- writer for the reference's trace file (src/trace.rs:5-42).
//
// Trace::save serialises json!({"steps": ..., "outcome": ...}) with serde_json::to_string_pretty:
//   * object keys in BTreeMap order: "outcome" before "steps" (and "termination" before "winner");
//   * 2-space indentation, every array element on its own line;
//   * steps[i] = [uci, q_root, [[uci, N, Q_sum, uct], ...]] (src/main.rs:198-218);
//   * f32 values are widened to f64 and printed with the shortest round-trip representation in
//     ryu's "pretty" style (e.g. 11.045379638671875, 0.0, 1e-7);  non-finite -> null;
//   * Move -> UCI string (src/chess.rs:233-240, 513-519); Outcome {termination, winner}
//     with enum variant names (src/chess.rs:87-105); Option::None -> null.
// Consumers: py/dataset.py:60-76, scripts/sample.py, the jq calls in scripts/run_batch:23-28.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <charconv>
#include <string>

namespace sctrace {

inline int move_uci(uint16_t m, char* buf) {
    static const char PCH[] = " pnbrqk";
    int f = m & 63, t = (m >> 6) & 63, pr = (m >> 12) & 7;
    int n = 0;
    buf[n++] = (char)('a' + (f & 7));
    buf[n++] = (char)('1' + (f >> 3));
    buf[n++] = (char)('a' + (t & 7));
    buf[n++] = (char)('1' + (t >> 3));
    if (pr >= 2 && pr <= 5) buf[n++] = PCH[pr];
    buf[n] = 0;
    return n;
}

// shortest round-trip decimal of a double, formatted like ryu::Buffer::format (serde_json floats)
inline std::string fmt_f64(double v) {
    if (!isfinite(v)) return "null";
    if (v == 0.0) return signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    // shortest digits that read back as v (std::to_chars is that by definition; trying printf precisions 1..17 against
    // strtod gave the same text at 27x the cost: an f32 widened to f64 needs 15-17 digits, 9 us per number, most of the
    // time the self-play CLI spent between two plies)
    const auto r = std::to_chars(buf, buf + sizeof buf - 1, v, std::chars_format::scientific);
    *r.ptr = 0;
    // buf = [-]d.ddddde[+-]XX
    std::string s(buf);
    bool neg = s[0] == '-';
    size_t epos = s.find('e');
    std::string mant = s.substr(neg ? 1 : 0, epos - (neg ? 1 : 0));
    int exp10 = atoi(s.c_str() + epos + 1);
    std::string digits;
    for (char c : mant)
        if (c != '.') digits.push_back(c);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    int length = (int)digits.size();
    int kk = exp10 + 1;          // decimal point position: value = 0.DIGITS * 10^kk
    int k = kk - length;         // value = DIGITS * 10^k
    std::string out = neg ? "-" : "";
    if (0 <= k && kk <= 16) {
        out += digits;
        out.append((size_t)k, '0');
        out += ".0";
    } else if (0 < kk && kk <= 16) {
        out += digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    } else if (-5 < kk && kk <= 0) {
        out += "0.";
        out.append((size_t)(-kk), '0');
        out += digits;
    } else if (length == 1) {
        out += digits + "e" + std::to_string(kk - 1);
    } else {
        out += digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
    }
    return out;
}

static const char* TERMINATION_NAMES[] = {"", "Checkmate", "Stalemate", "InsufficientMaterial", "SeventyfiveMoves",
                                          "FivefoldRepetition", "FiftyMoves", "ThreefoldRepetition", "VariantWin",
                                          "VariantLoss", "VariantDraw"};

inline std::string trace_to_json(int n_steps, int has_outcome, int termination, int winner, const uint16_t* step_move,
                                 const float* step_q, const int32_t* child_off, const uint16_t* child_move,
                                 const int32_t* child_n, const float* child_q, const float* child_uct) {
    std::string o;
    o.reserve((size_t)n_steps * 4096 + 256);
    char mv[8];
    o += "{\n  \"outcome\": ";
    if (!has_outcome) {
        o += "null";
    } else {
        o += "{\n    \"termination\": \"";
        o += TERMINATION_NAMES[termination >= 0 && termination <= 10 ? termination : 0];
        o += "\",\n    \"winner\": ";
        o += winner == 1 ? "\"White\"" : winner == 0 ? "\"Black\"" : "null";
        o += "\n  }";
    }
    o += ",\n  \"steps\": ";
    if (n_steps == 0) {
        o += "[]";
    } else {
        o += "[\n";
        for (int i = 0; i < n_steps; i++) {
            move_uci(step_move[i], mv);
            o += "    [\n      \"";
            o += mv;
            o += "\",\n      ";
            o += fmt_f64((double)step_q[i]);
            o += ",\n      ";
            int a = child_off[i], b = child_off[i + 1];
            if (a == b) {
                o += "[]";
            } else {
                o += "[\n";
                for (int c = a; c < b; c++) {
                    move_uci(child_move[c], mv);
                    o += "        [\n          \"";
                    o += mv;
                    o += "\",\n          ";
                    o += std::to_string(child_n[c]);
                    o += ",\n          ";
                    o += fmt_f64((double)child_q[c]);
                    o += ",\n          ";
                    o += fmt_f64((double)child_uct[c]);
                    o += "\n        ]";
                    o += c + 1 < b ? ",\n" : "\n";
                }
                o += "      ]";
            }
            o += "\n    ]";
            o += i + 1 < n_steps ? ",\n" : "\n";
        }
        o += "  ]";
    }
    o += "\n}";
    return o;
}

}  // namespace sctrace
