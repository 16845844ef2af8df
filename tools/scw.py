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

SCW2 blob = the fp8 export (BASELINE.json configs[4]; the reference's analogue is the bf16-autocast export of
py/export.py:47-65): b"SCW2", u32 n_blocks, u32 channels, u32 n_tensors, u32 precision (1 = fp8), then per tensor
u32 ndim, u32 shape[4], u64 numel, u32 encoding and
    encoding 0: float32 data[numel]
    encoding 1: int8 exponent[shape[0]] (one power-of-two scale per output channel), uint8 e4m3[numel]
                value = e4m3 * 2^exponent -- the conv weights that run on the fp8 matrix cores (stem, residual-block
                convs, the three head convs); everything else stays float32.
Quantisation rules (identical in smart-chess-rust_amd/csrc/weights.hpp and oracle/nn.c): OCP e4m3, round to nearest even,
subnormals kept, |x| > 448 clamps; channel exponent = the smallest e with max|w| / 2^e <= 448.
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


def is_fp8_conv(name):
    """the tensors that run in e4m3: every conv weight except the squeeze-excitation 1x1s"""
    return name.endswith(".weight") and (name == "conv_block.0.weight" or name.endswith("conv1.weight") or name.endswith("conv2.weight")
                                         or name in ("value_head.conv.0.weight", "policy_head.model.0.weight", "policy_head.model.2.weight"))


def e4m3_round(x):
    """float32 -> the nearest OCP e4m3 value (as float32): ties to even, subnormals of 2^-9, clamp to +-448"""
    x = np.asarray(x, np.float32)
    a = np.minimum(np.abs(x), np.float32(448.0)).astype(np.float32)
    _, ex = np.frexp(a)
    q = np.where(a >= np.float32(2.0 ** -6), np.ldexp(np.float32(1.0), ex - 4), np.float32(2.0 ** -9)).astype(np.float32)
    return np.copysign(np.rint(a / q).astype(np.float32) * q, x).astype(np.float32)


def e4m3_encode(r):
    """representable float32 values -> e4m3 bytes"""
    r = np.asarray(r, np.float32)
    a = np.abs(r)
    f, ex = np.frexp(a)
    sub = a < np.float32(2.0 ** -6)
    e = np.where(sub, 0, ex - 1 + 7).astype(np.int32)
    m = np.where(sub, np.rint(a * 512.0), np.rint((f * 2.0 - 1.0) * 8.0)).astype(np.int32)
    b = ((e << 3) | m).astype(np.uint8)
    b[a == 0] = 0
    return (b | np.where(np.signbit(r), 0x80, 0).astype(np.uint8)).astype(np.uint8)


def e4m3_decode(b):
    b = np.asarray(b, np.uint8).astype(np.int32)
    e, m = (b >> 3) & 15, b & 7
    v = np.where(e == 0, np.ldexp(m.astype(np.float32), -9), np.ldexp((1.0 + m / 8.0).astype(np.float32), e - 7)).astype(np.float32)
    return np.where(b & 0x80, -v, v).astype(np.float32)


def channel_exps(w):
    """per output channel (axis 0): the smallest e with max|w| / 2^e <= 448 (0 for an all-zero channel)"""
    m = np.abs(np.asarray(w, np.float32)).reshape(w.shape[0], -1).max(axis=1)
    f, ex = np.frexp((m / np.float32(448.0)).astype(np.float32))
    e = np.where(f == 0.5, ex - 1, ex)
    return np.clip(np.where(m > 0, e, 0), -100, 100).astype(np.int8)


def quantize_fp8(w):
    """conv weight -> (exponents int8[O], e4m3 bytes uint8[w.shape], dequantised float32)"""
    w = np.asarray(w, np.float32)
    e = channel_exps(w)
    sh = (-1,) + (1,) * (w.ndim - 1)
    r = e4m3_round(np.ldexp(w, -e.astype(np.int32).reshape(sh)))
    return e, e4m3_encode(r), np.ldexp(r, e.astype(np.int32).reshape(sh)).astype(np.float32)


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


def write_scw(path, state_dict, n_blocks, C=256, fp8=False):
    """state_dict: name -> array (names as in the reference; a leading 'model.' is stripped,
    as reference py/module.py:168-175 does for Lightning checkpoints).  fp8: write the SCW2 fp8 export."""
    sd = {(k[6:] if k.startswith("model.") else k): np.asarray(v, np.float32) for k, v in state_dict.items()}
    table = tensor_table(n_blocks, C)
    with open(path, "wb") as f:
        if fp8:
            f.write(b"SCW2" + struct.pack("<IIII", n_blocks, C, len(table), 1))
        else:
            f.write(b"SCW1" + struct.pack("<III", n_blocks, C, len(table)))
        for name, shape, _, _ in table:
            a = np.ascontiguousarray(sd[name], np.float32)
            assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
            sh = list(shape) + [1] * (4 - len(shape))
            f.write(struct.pack("<I4IQ", len(shape), *sh, a.size))
            if fp8 and is_fp8_conv(name):
                e, q, _ = quantize_fp8(a)
                f.write(struct.pack("<I", 1) + e.tobytes() + q.tobytes())
            else:
                if fp8:
                    f.write(struct.pack("<I", 0))
                f.write(a.tobytes())


def read_scw(path):
    """-> (n_blocks, C, state_dict); fp8 tensors of an SCW2 blob come back dequantised (e4m3 * 2^exponent)"""
    with open(path, "rb") as f:
        magic = f.read(4)
        assert magic in (b"SCW1", b"SCW2")
        n_blocks, C, nt = struct.unpack("<III", f.read(12))
        if magic == b"SCW2":
            assert struct.unpack("<I", f.read(4))[0] == 1
        table = tensor_table(n_blocks, C)
        assert nt == len(table)
        sd = {}
        for name, shape, _, _ in table:
            nd, s0, s1, s2, s3, numel = struct.unpack("<I4IQ", f.read(28))
            enc = struct.unpack("<I", f.read(4))[0] if magic == b"SCW2" else 0
            if enc == 1:
                e = np.frombuffer(f.read(s0), np.int8).astype(np.int32)
                q = np.frombuffer(f.read(numel), np.uint8).reshape([s0, s1, s2, s3][:nd])
                a = np.ldexp(e4m3_decode(q), e.reshape((-1,) + (1,) * (nd - 1))).astype(np.float32)
            else:
                a = np.frombuffer(f.read(4 * numel), np.float32).reshape([s0, s1, s2, s3][:nd])
            sd[name] = a.copy()
    return n_blocks, C, sd
