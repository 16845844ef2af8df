"""Build-owned deterministic weights + the SCW1 weight-blob format (shared by oracle, engine, tools).

Tensor table = ChessModule(n_res_blocks).state_dict() order (reference py/module.py:109-133,
names listed in SURVEY.md section 8 row a19).  `channels` is 256 in the reference; 128 is the
build-defined trunk width of BASELINE.json configs[1] (heads stay 256 wide).

PRNG (identical in oracle/nn.c:orc_prng_weight and csrc/weights.cpp):
    h = mix64(seed*GOLD + tensor*K + idx);  u = h >> 40;  x = (u + 0.5) / 2^23 - 1   (float64)
    value = float32(shift + x*scale)
    conv/linear weight, bias: scale = 1/sqrt(fan_in), shift 0;  LN weight: 1 +- 0.25;  LN bias: +-0.25

SCW1 blob (little endian): b"SCW1", u32 n_blocks, u32 channels, u32 n_tensors, then per tensor
u32 ndim, u32 shape[4], u64 numel, float32 data[numel]  (state_dict order, PyTorch layout).
"""
import struct

import numpy as np

GOLD = np.uint64(0x9E3779B97F4A7C15)
K2 = np.uint64(0xD1B54A32D192ED03)


def mix64(z):
    z = (z + GOLD).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
    return z ^ (z >> np.uint64(31))


def tensor_table(n_blocks, C=256):
    """[(name, shape, kind, fan_in)]; kind 0 weight, 1 bias, 2 LN weight, 3 LN bias"""
    H = 256
    t = [("conv_block.0.weight", (C, 112, 3, 3), 0, 112 * 9), ("conv_block.0.bias", (C,), 1, 112 * 9),
         ("conv_block.1.weight", (C,), 2, 0), ("conv_block.1.bias", (C,), 3, 0)]
    for i in range(n_blocks):
        p = f"res_blocks.{i}."
        t += [(p + "conv1.weight", (C, C, 3, 3), 0, C * 9), (p + "conv1.bias", (C,), 1, C * 9),
              (p + "bn1.weight", (C,), 2, 0), (p + "bn1.bias", (C,), 3, 0),
              (p + "conv2.weight", (C, C, 3, 3), 0, C * 9), (p + "conv2.bias", (C,), 1, C * 9),
              (p + "bn2.weight", (C,), 2, 0), (p + "bn2.bias", (C,), 3, 0),
              (p + "se.fc1.weight", (C // 2, C, 1, 1), 0, C), (p + "se.fc1.bias", (C // 2,), 1, C),
              (p + "se.fc2.weight", (C, C // 2, 1, 1), 0, C // 2), (p + "se.fc2.bias", (C,), 1, C // 2)]
    t += [("value_head.conv.0.weight", (H, C, 1, 1), 0, C), ("value_head.conv.0.bias", (H,), 1, C),
          ("value_head.conv.1.weight", (H,), 2, 0), ("value_head.conv.1.bias", (H,), 3, 0),
          ("value_head.ffn.0.weight", (128, 64 * H + 7), 0, 64 * H + 7), ("value_head.ffn.0.bias", (128,), 1, 64 * H + 7),
          ("value_head.ffn.2.weight", (1, 128), 0, 128), ("value_head.ffn.2.bias", (1,), 1, 128),
          ("policy_head.model.0.weight", (H, C, 1, 1), 0, C), ("policy_head.model.0.bias", (H,), 1, C),
          ("policy_head.model.1.weight", (H,), 2, 0), ("policy_head.model.1.bias", (H,), 3, 0),
          ("policy_head.model.2.weight", (73, H, 1, 1), 0, H), ("policy_head.model.2.bias", (73,), 1, H),
          ("policy_head.model.3.weight", (73,), 2, 0), ("policy_head.model.3.bias", (73,), 3, 0)]
    return t


def prng_tensor(seed, t_index, shape, kind, fan_in):
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * GOLD + np.uint64(t_index) * K2
        h = mix64((base + np.arange(n, dtype=np.uint64)).astype(np.uint64))
    u = (h >> np.uint64(40)).astype(np.float64)
    x = (u + 0.5) / 8388608.0 - 1.0
    scale = 1.0 / np.sqrt(float(fan_in)) if kind <= 1 else 0.25
    shift = 1.0 if kind == 2 else 0.0
    return (shift + x * scale).astype(np.float32).reshape(shape)


def prng_state_dict(n_blocks, C=256, seed=0):
    return {name: prng_tensor(seed, i, shape, kind, fan)
            for i, (name, shape, kind, fan) in enumerate(tensor_table(n_blocks, C))}


def write_scw(path, state_dict, n_blocks, C=256):
    """state_dict: name -> array (names as in the reference; a leading 'model.' is stripped,
    as reference py/module.py:168-175 does for Lightning checkpoints)."""
    sd = {(k[6:] if k.startswith("model.") else k): np.asarray(v, np.float32) for k, v in state_dict.items()}
    table = tensor_table(n_blocks, C)
    with open(path, "wb") as f:
        f.write(b"SCW1" + struct.pack("<III", n_blocks, C, len(table)))
        for name, shape, _, _ in table:
            a = np.ascontiguousarray(sd[name], np.float32)
            assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
            sh = list(shape) + [1] * (4 - len(shape))
            f.write(struct.pack("<I4IQ", len(shape), *sh, a.size))
            f.write(a.tobytes())


def read_scw(path):
    with open(path, "rb") as f:
        assert f.read(4) == b"SCW1"
        n_blocks, C, nt = struct.unpack("<III", f.read(12))
        table = tensor_table(n_blocks, C)
        assert nt == len(table)
        sd = {}
        for name, shape, _, _ in table:
            nd, s0, s1, s2, s3, numel = struct.unpack("<I4IQ", f.read(28))
            a = np.frombuffer(f.read(4 * numel), np.float32).reshape([s0, s1, s2, s3][:nd])
            sd[name] = a.copy()
    return n_blocks, C, sd
