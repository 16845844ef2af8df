"""developer tool: the shader clock the chip holds while the tower runs, at one and at two tower workgroups
per CU -- in-kernel stamps of s_memtime (shader clock) and s_memrealtime (100 MHz wall clock) around every workgroup's tower,
read from the LAST of SC_EXP_REPEAT back-to-back launches (>= 2 s of the same work: the clock has settled).
Needs the experiment build (tools/build_exp.sh; SC_EXP_DEFS=-DSC_T32_OCC=2 for two workgroups per CU):
    SC_ENGINE_LIB=smart-chess-rust_amd/lib_exp/libsc_engine.so SC_DBG_N=512 python tools/dbg_clock.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
g = np.load(os.path.join(ROOT, "tests", "golden", "nn_ref_b10_c256.npz"))
N = int(os.environ.get("SC_DBG_N", "256"))
C = int(os.environ.get("SC_DBG_C", "128"))
boards = np.repeat(g["boards"], (N + 7) // 8, axis=0)[:N]
meta = np.repeat(g["meta"], (N + 7) // 8, axis=0)[:N]
eng = scamd.Engine(10, C, seed=1, precision=os.environ.get("SC_PREC", "bf16"))
eng.debug(boards, meta, 2001)                    # warm-up
os.environ["SC_EXP_REPEAT"] = os.environ.get("SC_DBG_REPEAT", "12000")
t0 = time.time()
d = eng.debug(boards, meta, 2001)
wall = time.time() - t0
d = d.reshape(N, -1)[:, :3].astype(np.float64)
cyc, rt, start = d[:, 0], d[:, 1], d[:, 2]
ghz = cyc / rt * 0.1
rep = int(os.environ["SC_EXP_REPEAT"]) + 1
print(f"{eng.precision} 10x{C}, {N} positions per launch, {rep} launches back to back in {wall:.2f} s ({wall / rep * 1e6:.1f} us per launch incl. value FC)")
print(f"  tower per workgroup: median {np.median(cyc):9.0f} shader cycles (s_memtime) in {np.median(rt) / 100:7.2f} us (s_memrealtime)")
print(f"  shader clock = d(s_memtime) / d(s_memrealtime): median {np.median(ghz):.3f} GHz, p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}")
span = (start.max() - start.min()) / 100.0
print(f"  workgroup start times spread over {span:.1f} us; launch extent (first start -> last end) {((start + rt).max() - start.min()) / 100.0:.1f} us")
eng.close()
