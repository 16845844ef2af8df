"""Checkpoint import: a reference `.ckpt` -> the engine's SCW1 weight blob.

Accepts what the reference's own loader accepts (py/module.py:157-181 `_load_ckpt`): a plain `state_dict` saved with
`torch.save`, or a Lightning checkpoint (`"pytorch-lightning_version"` present: the weights are under `"state_dict"`
and every key loses its first dotted component, e.g. `model.`).  The file is read with
`torch.load(..., weights_only=True)` only -- nothing in it is executed.  Network depth and trunk width are inferred
from the tensors (`res_blocks.<i>.*`, `conv_block.0.weight`).

    python tools/ckpt_to_scw.py last.ckpt last.scw      ->   sc-selfplay -c last.scw ... / scamd.Engine(weights="last.scw")
    python tools/ckpt_to_scw.py --fp8 last.ckpt last8.scw   the fp8 export (SCW2: e4m3 conv weights + one power-of-two scale per
                                                            output channel, tools/scw.py); the engine then runs its fp8 tower
"""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scw  # noqa: E402


def state_dict_from_checkpoint(obj):
    """the reference's key handling (py/module.py:168-175)"""
    if "pytorch-lightning_version" in obj:
        return {k.split(".", 1)[1]: v for k, v in obj["state_dict"].items()}
    return dict(obj)


def infer_shape(sd):
    blocks = {int(m.group(1)) for k in sd for m in [re.match(r"res_blocks\.(\d+)\.", k)] if m}
    n_blocks = max(blocks) + 1 if blocks else 0
    if blocks != set(range(n_blocks)):
        raise ValueError(f"res_blocks indices are not contiguous: {sorted(blocks)}")
    C = int(sd["conv_block.0.weight"].shape[0])
    if C not in (128, 256):
        raise ValueError(f"unsupported trunk width {C} (the engine builds 128 and 256)")
    return n_blocks, C


def convert(src, dst, fp8=False):
    import torch
    obj = torch.load(src, map_location="cpu", weights_only=True)
    sd = state_dict_from_checkpoint(obj)
    sd = {k: (v.detach().to(torch.float32).numpy() if hasattr(v, "detach") else np.asarray(v, np.float32)) for k, v in sd.items()}
    n_blocks, C = infer_shape(sd)
    names = {n for n, _, _, _ in scw.tensor_table(n_blocks, C)}
    missing = sorted(names - set(sd))
    if missing:
        raise ValueError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
    scw.write_scw(dst, sd, n_blocks, C, fp8=fp8)
    return n_blocks, C, sorted(set(sd) - names)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--fp8"]
    if len(args) != 2:
        raise SystemExit(__doc__)
    nb, C, extra = convert(args[0], args[1], fp8="--fp8" in sys.argv)
    print(f"{args[1]}: {nb} blocks x {C} channels" + (" (fp8 export)" if "--fp8" in sys.argv else "") + (f"; ignored keys: {extra[:5]}" if extra else ""))
