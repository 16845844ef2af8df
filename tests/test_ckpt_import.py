"""CPU: reference checkpoint -> SCW1 blob (tools/ckpt_to_scw.py), with the key handling of py/module.py:157-181."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("lightning", [False, True])
@pytest.mark.parametrize("nb,C", [(2, 256), (3, 128)])
def test_checkpoint_round_trip(tmp_path, lightning, nb, C):
    torch = pytest.importorskip("torch")
    import ckpt_to_scw
    import scw
    sd = scw.prng_state_dict(nb, C, seed=5)
    tsd = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    if lightning:   # what lightning's ModelCheckpoint writes for a module holding the net as `self.model`
        obj = {"pytorch-lightning_version": "2.5.0", "epoch": 3, "global_step": 100,
               "state_dict": {"model." + k: v for k, v in tsd.items()}}
    else:
        obj = tsd
    src, dst = str(tmp_path / "x.ckpt"), str(tmp_path / "x.scw")
    torch.save(obj, src)
    got = ckpt_to_scw.convert(src, dst)
    assert got[:2] == (nb, C) and got[2] == []
    nb2, C2, back = scw.read_scw(dst)
    assert (nb2, C2) == (nb, C) and set(back) == set(sd)
    for k in sd:
        assert np.array_equal(back[k], sd[k]), k


def test_checkpoint_with_missing_tensor_is_rejected(tmp_path):
    torch = pytest.importorskip("torch")
    import ckpt_to_scw
    import scw
    sd = scw.prng_state_dict(1, 256, seed=1)
    sd.pop("value_head.ffn.2.bias")
    torch.save({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, str(tmp_path / "bad.ckpt"))
    with pytest.raises(ValueError):
        ckpt_to_scw.convert(str(tmp_path / "bad.ckpt"), str(tmp_path / "bad.scw"))
