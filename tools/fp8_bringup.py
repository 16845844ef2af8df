"""developer tool: fp8 tower bring-up -- HIP fp8 forward vs the oracle's fp8-emulating mode and vs the fp32 goldens"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd")); sys.path.insert(0, ROOT)
import scamd
from oracle import oracle_py as orc
g = np.load(os.path.join(ROOT, "tests", "golden", "nn_ref_b1_c256.npz"))
for C, nb in [(128, 0), (256, 0), (128, 1), (256, 1), (128, 3), (256, 2), (128, 10), (256, 10)]:
    eng = scamd.Engine(nb, C, seed=9, precision="fp8")
    net = orc.Net(nb, C, seed=9, emulate_fp8=True)
    logp, val = eng.forward(g["boards"][:4], g["meta"][:4])
    for stage in ([0, 1] if nb else [0]):
        lat = eng.debug(g["boards"][:1], g["meta"][:1], stage if stage == 0 else 1000)[0]
    dl, dv = [], []
    for k in range(4):
        ol, ov, olat = net.forward(g["boards"][k], g["meta"][k], latent=True)
        dl.append(np.abs(logp[k] - ol).max()); dv.append(abs(val[k] - ov))
        if k == 0:
            dlat = np.abs(lat - olat).max() / max(1.0, np.abs(olat).max())
    print(f"fp8 {nb}x{C}: vs fp8 oracle max|dlogp|={max(dl):.4f} |dvalue|={max(dv):.5f} latent rel {dlat:.4f}  finite={np.isfinite(logp).all()}", flush=True)
    eng.close()
for nb in (1, 10, 20):
    gg = np.load(os.path.join(ROOT, "tests", "golden", f"nn_ref_b{nb}_c256.npz"))
    eng = scamd.Engine(nb, 256, seed=int(gg["seed"]), precision="fp8")
    logp, val = eng.forward(gg["boards"], gg["meta"])
    tv = 0.5 * np.abs(np.exp(logp.astype(np.float64)) - np.exp(gg["logp"].astype(np.float64))).sum(axis=1)
    print(f"fp8 {nb}x256 vs fp32 reference goldens: max|dlogp|={np.abs(logp - gg['logp']).max():.4f} max|dvalue|={np.abs(val - gg['value']).max():.4f} max TVD={tv.max():.4f}", flush=True)
    eng.close()
