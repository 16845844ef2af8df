"""CPU: the fp8 export (tools/scw.py SCW2, tools/ckpt_to_scw.py --fp8) and the e4m3 quantisation rules, which exist three
times (numpy here, C in the oracle, C++ in the engine's weights.hpp) and must agree bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scw  # noqa: E402


def test_e4m3_rules_numpy_equals_oracle_c(orc):
    L = orc.lib()
    rnd = np.random.RandomState(0)
    x = np.concatenate([rnd.standard_normal(20000).astype(np.float32) * np.float32(10) ** rnd.uniform(-4, 3, 20000).astype(np.float32),
                        np.array([0, -0.0, 448, 449, 464, 480, 1000, -1e30, 2 ** -9, 2 ** -10, 1.5 * 2 ** -10, 0.0155, 2 ** -6, 1.0625, 1.1875], np.float32)])
    r = scw.e4m3_round(x)
    assert np.array_equal(r, np.array([L.orc_e4m3_round(float(v)) for v in x], np.float32))
    # known answers measured on the MI355X's v_cvt_pk_fp8_f32 (tools/experiments/mfma_fp8_layout.hip), below the clamp
    known = {0.0: 0.0, 1.0: 1.0, 1.0625: 1.0, 1.125: 1.125, 1.1875: 1.25, 447.0: 448.0, 448.0: 448.0, 464.0: 448.0, 0.001: 0.001953125,
             0.0009765625: 0.0, 0.00146: 0.001953125}
    for k, v in known.items():
        assert float(scw.e4m3_round(np.float32(k))) == v, k
    assert float(scw.e4m3_round(np.float32(1000.0))) == 448.0 and float(scw.e4m3_round(np.float32(-1e30))) == -448.0   # clamp (the convert alone gives NaN)
    b = scw.e4m3_encode(r)
    assert np.array_equal(scw.e4m3_decode(b), r) and b.max() <= 0xFE and not np.any((b & 0x7F) == 0x7F)   # never the NaN code
    assert sorted(set(scw.e4m3_decode(np.arange(256, dtype=np.uint8))[np.arange(256) & 0x7F != 0x7F].tolist()))[-1] == 448.0
    m = np.abs(x[:3000]) + np.float32(1e-12)
    assert np.array_equal(scw.channel_exps(m.reshape(-1, 1)), np.array([L.orc_fp8_channel_exp(float(v)) for v in m], np.int8))
    for mx, e in ((448.0, 0), (449.0, 1), (224.0, -1), (224.5, 0), (1.0, -8), (0.0, 0)):
        assert int(scw.channel_exps(np.array([[mx]], np.float32))[0]) == e, mx


def test_scw2_round_trip_and_conv_selection(tmp_path):
    sd = scw.prng_state_dict(2, 128, seed=4)
    p = str(tmp_path / "w8.scw")
    scw.write_scw(p, sd, 2, 128, fp8=True)
    nb, C, back = scw.read_scw(p)
    assert (nb, C) == (2, 128) and set(back) == set(sd)
    n8 = 0
    for name, w in sd.items():
        if scw.is_fp8_conv(name):
            e, q, deq = scw.quantize_fp8(w)
            assert np.array_equal(back[name], deq) and w.ndim == 4 and e.shape == (w.shape[0],)
            assert np.abs(deq - w).max() <= np.abs(w).max() * 2 ** -4          # 3 mantissa bits: half a step of the top binade
            assert np.array_equal(scw.quantize_fp8(deq)[1], q)                   # re-quantising the export is lossless
            n8 += 1
        else:
            assert np.array_equal(back[name], w), name
    assert n8 == 1 + 2 * 2 + 3                                                   # stem, 2 convs per block, 3 head convs
    want = 20 + sum(32 + (v.shape[0] + v.size if scw.is_fp8_conv(k) else 4 * v.size) for k, v in sd.items())
    assert os.path.getsize(p) == want                                            # conv weights take one byte each


def test_ckpt_to_scw_fp8(tmp_path):
    torch = pytest.importorskip("torch")
    import ckpt_to_scw
    sd = scw.prng_state_dict(1, 256, seed=6)
    torch.save({"pytorch-lightning_version": "2.5.0", "state_dict": {"model." + k: torch.from_numpy(v.copy()) for k, v in sd.items()}},
               str(tmp_path / "x.ckpt"))
    assert ckpt_to_scw.convert(str(tmp_path / "x.ckpt"), str(tmp_path / "x8.scw"), fp8=True)[:2] == (1, 256)
    assert open(str(tmp_path / "x8.scw"), "rb").read(4) == b"SCW2"
    _, _, back = scw.read_scw(str(tmp_path / "x8.scw"))
    assert np.array_equal(back["res_blocks.0.conv1.weight"], scw.quantize_fp8(sd["res_blocks.0.conv1.weight"])[2])
    assert np.array_equal(back["res_blocks.0.se.fc1.weight"], sd["res_blocks.0.se.fc1.weight"])


def test_fp8_oracle_mode_meets_the_stated_tolerance(orc):
    """the checker itself: e4m3 convs stay within the stated fp8 tolerance (prior total variation < 0.05, |value| error
    < 0.05) of the vectors produced by the reference module"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "nn_ref_b10_c256.npz"))
    net = orc.Net(10, 256, seed=int(g["seed"]), emulate_fp8=True)
    for k in range(4):
        lp, v = net.forward(g["boards"][k], g["meta"][k])
        tv = 0.5 * np.abs(np.exp(lp.astype(np.float64)) - np.exp(g["logp"][k].astype(np.float64))).sum()
        assert tv < 0.05 and abs(v - g["value"][k]) < 0.05
