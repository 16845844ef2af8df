"""Generate tests/golden/nn_*.npz from the REFERENCE network (py/module.py), imported in place.

Runs only in the build container (needs /root/reference); the GPU box uses the committed .npz.
torchvision / timm are absent from the image, so two stand-ins restating the published third-party
modules are registered before the import:
    torchvision.ops.SqueezeExcitation(C, S): avgpool -> Conv2d(C,S,1) 'fc1' -> ReLU -> Conv2d(S,C,1) 'fc2' -> Sigmoid -> scale*x
    timm.layers.norm.LayerNorm2d(C, eps=1e-6): layer_norm over the channel dim of NCHW
No reference source is copied: only (input, output) vectors are written.

Inputs are positions reached by short move sequences, encoded by the oracle's restatement of
_encode (src/chess.rs:845-877); weights come from the build-owned PRNG (tools/scw.py) written into
the reference module's state_dict, so nothing but KB-sized vectors is committed.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference/py"


def install_standins():
    class SqueezeExcitation(torch.nn.Module):
        def __init__(self, input_channels, squeeze_channels):
            super().__init__()
            self.avgpool = torch.nn.AdaptiveAvgPool2d(1)
            self.fc1 = torch.nn.Conv2d(input_channels, squeeze_channels, 1)
            self.fc2 = torch.nn.Conv2d(squeeze_channels, input_channels, 1)
            self.activation = torch.nn.ReLU()
            self.scale_activation = torch.nn.Sigmoid()

        def forward(self, x):
            s = self.avgpool(x)
            s = self.scale_activation(self.fc2(self.activation(self.fc1(s))))
            return s * x

    class LayerNorm2d(torch.nn.LayerNorm):
        def __init__(self, num_channels, eps=1e-6, affine=True):
            super().__init__(num_channels, eps=eps, elementwise_affine=affine)

        def forward(self, x):
            x = x.permute(0, 2, 3, 1)
            x = torch.nn.functional.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
            return x.permute(0, 3, 1, 2)

    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")
    tvo.SqueezeExcitation = SqueezeExcitation
    tv.ops = tvo
    tm = types.ModuleType("timm")
    tml = types.ModuleType("timm.layers")
    tmn = types.ModuleType("timm.layers.norm")
    tmn.LayerNorm2d = LayerNorm2d
    tml.norm = tmn
    tm.layers = tml
    for k, v in {"torchvision": tv, "torchvision.ops": tvo, "timm": tm, "timm.layers": tml,
                 "timm.layers.norm": tmn}.items():
        sys.modules[k] = v


GAMES = {
    "start": [],
    "black_to_move": ["e2e4"],
    "short_history": ["d2d4", "d7d5", "c2c4"],
    "castled": ["e2e4", "e7e5", "g1f3", "b8c6", "f1c4", "f8c5", "e1g1", "g8f6", "d2d3", "e8g8"],
    "rights_lost": ["e2e4", "e7e5", "e1e2", "e8e7", "e2e1", "e7e8", "g1f3"],
    "repetition": ["g1f3", "g8f6", "f3g1", "f6g8", "g1f3", "g8f6", "f3g1", "f6g8", "g1f3"],
    "promotion_near": ["a2a4", "b7b5", "a4b5", "a7a6", "b5a6", "c8b7", "a6b7", "b8c6", "b7a8q", "d8a8"],
    "ref_selfplay_40": ["e2e4", "b7b6", "f2f4", "b8c6", "d1g4", "c8b7", "g1f3", "d7d5", "a2a4", "d5e4", "f1b5",
                        "b7a6", "g4e6", "g8f6", "b5e2", "d8c8", "b2b3", "c8e6", "a1a2", "a8c8", "e1f1", "a6b7",
                        "f1e1", "f6d7", "e1f1", "f7f5", "e2a6", "e6d6", "b1c3", "g7g5", "a6b5", "d6d4", "d2d3",
                        "e4f3", "g2g4", "b7a6", "c3d1", "d4e4", "a2b2", "f5g4", "c1e3"],
}


def main():
    install_standins()
    sys.path.insert(0, REF)
    import module as refmod  # noqa: the reference network, imported in place

    import scw
    from oracle import oracle_py as orc

    boards, metas, names = [], [], []
    for name, moves in GAMES.items():
        st = orc.State()
        for m in moves:
            assert orc.from_uci(m) in st.legal_moves(), (name, m)
            st.push(m)
        b, mt = st.encode()
        boards.append(b)
        metas.append(mt)
        names.append(name)
    boards = np.stack(boards)
    metas = np.stack(metas)

    out_dir = os.path.join(ROOT, "tests", "golden")
    # 19 = the reference's default depth (py/module.py:110), 20 = BASELINE configs[3]; existing files are kept unless --force
    for n_blocks in (1, 10, 19, 20):
        if os.path.exists(os.path.join(out_dir, f"nn_ref_b{n_blocks}_c256.npz")) and "--force" not in sys.argv:
            continue
        model = refmod.load_model(n_res_blocks=n_blocks, device="cpu", compile=False, inference=True)
        sd = model.state_dict()
        table = scw.tensor_table(n_blocks, 256)
        assert [k for k in sd.keys()] == [t[0] for t in table], "state_dict order differs from tools/scw.py"
        new = scw.prng_state_dict(n_blocks, 256, seed=20260501 + n_blocks)
        for (name, shape, _, _) in table:
            assert tuple(sd[name].shape) == tuple(shape), name
        model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()}, strict=True)
        inp = torch.from_numpy(boards.astype(np.float32)).permute(0, 3, 1, 2).contiguous()
        meta = torch.from_numpy(metas.astype(np.float32))
        with torch.no_grad():
            logp, value = model(inp, meta)
        np.savez_compressed(os.path.join(out_dir, f"nn_ref_b{n_blocks}_c256.npz"), names=np.array(names),
                            boards=boards, meta=metas, logp=logp.numpy().astype(np.float32),
                            value=value.numpy().astype(np.float32).reshape(-1), seed=np.int64(20260501 + n_blocks),
                            n_params=np.int64(sum(p.numel() for p in model.parameters())))
        print(n_blocks, "blocks:", logp.shape, value.reshape(-1)[:4], "params", sum(p.numel() for p in model.parameters()))

    # BASELINE configs[1]'s 128-channel trunk has no instantiation in the reference (ChessModule hard-codes 256, py/module.py:120-133).
    # Its building blocks do take a width: the network below is ASSEMBLED from the reference's own classes -- ResBlockSE(128, 128)
    # (py/module.py:14-46: 20 of its 23 convs, both LayerNorms and the SE of every block) and ValueHead(128, "LayerNorm")
    # (py/module.py:83-106) -- around a stem and a policy head written out layer by layer from the reference's lists with the widths
    # BASELINE gives them (py/module.py:120-126 with 128 output channels; py/module.py:70-80 with its second conv taking the first
    # one's 256 outputs: PolicyHead(din) itself only runs with din == 256), and ChessModule.forward's body (py/module.py:136-152).
    path = os.path.join(out_dir, "nn_ref_b10_c128.npz")
    if not os.path.exists(path) or "--force" in sys.argv:
        n_blocks, C, seed = 10, 128, 20260701
        LN = refmod.NormTable["LayerNorm"]

        class NarrowPolicy(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.model = torch.nn.Sequential(torch.nn.Conv2d(C, 256, kernel_size=1), LN(256), torch.nn.Conv2d(256, 73, kernel_size=1), LN(73),
                                                 torch.nn.Flatten())

            def forward(self, x):
                return torch.nn.functional.log_softmax(self.model(x), dim=1)

        class Narrow(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.conv_block = torch.nn.Sequential(torch.nn.Conv2d(112, C, kernel_size=3, stride=1, padding=1), LN(C), torch.nn.ReLU())
                self.res_blocks = torch.nn.ModuleList([refmod.ResBlockSE(C, C) for _ in range(n_blocks)])
                self.value_head = refmod.ValueHead(C, norm="LayerNorm")
                self.policy_head = NarrowPolicy()

            def forward(self, inp, meta):
                x = self.conv_block(inp)
                for block in self.res_blocks:
                    x = block(x)
                v1 = self.policy_head(x)
                v2 = self.value_head(x, meta) * (meta[:, 0].unsqueeze(-1) * 2 - 1)
                return v1, v2

        model = Narrow().eval()
        table = scw.tensor_table(n_blocks, C)
        sd = model.state_dict()
        assert [k for k in sd.keys()] == [t[0] for t in table], "state_dict order differs from tools/scw.py"
        for (name, shape, _, _) in table:
            assert tuple(sd[name].shape) == tuple(shape), name
        new = scw.prng_state_dict(n_blocks, C, seed=seed)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()}, strict=True)
        inp = torch.from_numpy(boards.astype(np.float32)).permute(0, 3, 1, 2).contiguous()
        meta = torch.from_numpy(metas.astype(np.float32))
        with torch.no_grad():
            logp, value = model(inp, meta)
        np.savez_compressed(path, names=np.array(names), boards=boards, meta=metas, logp=logp.numpy().astype(np.float32),
                            value=value.numpy().astype(np.float32).reshape(-1), seed=np.int64(seed),
                            n_params=np.int64(sum(p.numel() for p in model.parameters())))
        print("narrow 10 x 128:", logp.shape, value.reshape(-1)[:4], "params", sum(p.numel() for p in model.parameters()))

    # A trained network's outputs are not those of a U(-k, k) init: its log-probabilities reach magnitudes of 12-21
    # (reference notebooks/check_model.ipynb cells 6-8) and its value leaves the linear part of tanh.  Same PRNG weights with
    # the gain of the policy head's last LayerNorm x 4 and value_head.ffn.2.weight x 2: logits of that magnitude, through the
    # reference module, as a second set of vectors.
    path = os.path.join(out_dir, "nn_ref_b10_c256_sharp.npz")
    if not os.path.exists(path) or "--force" in sys.argv:
        n_blocks, seed = 10, 20260601
        model = refmod.load_model(n_res_blocks=n_blocks, device="cpu", compile=False, inference=True)
        new = scw.prng_state_dict(n_blocks, 256, seed=seed)
        new["policy_head.model.3.weight"] = new["policy_head.model.3.weight"] * np.float32(4.0)
        new["value_head.ffn.2.weight"] = new["value_head.ffn.2.weight"] * np.float32(2.0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()}, strict=True)
        inp = torch.from_numpy(boards.astype(np.float32)).permute(0, 3, 1, 2).contiguous()
        meta = torch.from_numpy(metas.astype(np.float32))
        with torch.no_grad():
            logp, value = model(inp, meta)
        np.savez_compressed(path, names=np.array(names), boards=boards, meta=metas, logp=logp.numpy().astype(np.float32),
                            value=value.numpy().astype(np.float32).reshape(-1), seed=np.int64(seed), policy_gain_scale=np.float32(4.0),
                            value_fc2_scale=np.float32(2.0))
        print("sharp 10 blocks: logp range", float(logp.min()), float(logp.max()), "values", value.reshape(-1))


if __name__ == "__main__":
    main()
