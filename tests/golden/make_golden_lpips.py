"""Golden vectors for the LPIPS term of the stage-2 / stage-3 loss (weight 1.0 in configs/default.yaml:97-101): outputs of the
REFERENCE's own `LPIPS(net='vgg')` (3rd_Complete_HOSNeRF/third_parties/lpips/lpips.py:22-118, pretrained_networks.py:97-135) as the
training step calls it (src/model/mipnerf360/model.py:1664-1678: `lpips_func(2 rgb - 1, 2 target - 1)` on [N, 3, 32, 32] patches, mean),
imported here (build container only).  torchvision is absent and so are the ImageNet weights of VGG-16 (a download): a stand-in
`torchvision.models.vgg16` returns the published VGG-16 `features` layer list with SEEDED random weights (`vgg16_features_state`
below, shared with the tests), so the fixture pins the algorithm -- scaling layer, the five feature taps, channel normalisation, the
reference's own learned 1x1 calibration weights (weights/v0.1/vgg.pth, stored in the fixture: 1 472 floats), spatial mean, sum --
and its gradient with respect to the prediction, not the ImageNet filters.
  python tests/golden/make_golden_lpips.py   ->  tests/golden/lpips.npz"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/3rd_Complete_HOSNeRF"
VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def vgg16_features_state(seed: int = 1234):
    """The `features.N.weight / bias` tensors of a VGG-16 with He-normal weights drawn from a seeded CPU generator (conv indices
    0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28 of torchvision's `vgg16().features`).  Biases are small and random so that
    they matter."""
    g = torch.Generator().manual_seed(seed)
    sd, cin, idx = {}, 3, 0
    for v in VGG16_CFG:
        if v == "M":
            idx += 1
            continue
        sd[f"{idx}.weight"] = torch.randn(v, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        sd[f"{idx}.bias"] = torch.randn(v, generator=g) * 0.05
        cin = v
        idx += 2
    return sd


def _stub_torchvision():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")

    def vgg16(pretrained=False):
        layers, cin = [], 3
        for v in VGG16_CFG:
            if v == "M":
                layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [torch.nn.Conv2d(cin, v, kernel_size=3, padding=1), torch.nn.ReLU(inplace=True)]
                cin = v
        feats = torch.nn.Sequential(*layers)
        feats.load_state_dict(vgg16_features_state())
        return types.SimpleNamespace(features=feats)

    models.vgg16 = vgg16
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models


def main():
    _stub_torchvision()
    sys.path.insert(0, REF)
    from third_parties.lpips import LPIPS
    torch.manual_seed(0)
    net = LPIPS(net="vgg", verbose=False)
    for p in net.parameters():
        p.requires_grad_(False)
    out = {"lin": np.concatenate([net.lins[k].model[1].weight.detach().reshape(-1).numpy() for k in range(5)]),
           "shift": net.scaling_layer.shift.reshape(-1).numpy(), "scale": net.scaling_layer.scale.reshape(-1).numpy()}
    rs = np.random.RandomState(7)
    for tag, n, P in (("a", 2, 32), ("b", 4, 32), ("c", 3, 16)):
        pred = torch.from_numpy(rs.uniform(0, 1, size=(n, P, P, 3)).astype(np.float32)).requires_grad_(True)
        targ = torch.from_numpy(np.clip(pred.detach().numpy() + 0.15 * rs.standard_normal((n, P, P, 3)), 0, 1).astype(np.float32))
        val, per_layer = net(2.0 * pred.permute(0, 3, 1, 2) - 1.0, 2.0 * targ.permute(0, 3, 1, 2) - 1.0, retPerLayer=True)
        loss = torch.mean(val)                                   # model.py:1676
        loss.backward()
        out[f"{tag}_pred"], out[f"{tag}_target"] = pred.detach().numpy(), targ.numpy()
        out[f"{tag}_val"] = val.detach().reshape(-1).numpy()
        out[f"{tag}_layers"] = np.stack([r.detach().reshape(-1).numpy() for r in per_layer], 0)
        out[f"{tag}_loss"] = np.float32(loss.item())
        out[f"{tag}_grad"] = pred.grad.numpy()
        # the same class evaluated in float64 (round 5): the yardstick for discrete events.  Fixture `b`: the float32 evaluation
        # above has ONE pre-activation within rounding of zero that takes the other ReLU branch than the exact value does -- its
        # gradient is 7.4e-3 of the largest gradient away from this one (13.8 % of the values by more than 1e-5), `a` and `c` 7e-6.
        net64 = net.double()
        p64 = pred.detach().double().requires_grad_(True)
        val64 = net64(2.0 * p64.permute(0, 3, 1, 2) - 1.0, 2.0 * targ.double().permute(0, 3, 1, 2) - 1.0)
        loss64 = torch.mean(val64)
        loss64.backward()
        out[f"{tag}_loss64"], out[f"{tag}_grad64"] = np.float64(loss64.item()), p64.grad.numpy()
        net = net64.float()
    np.savez_compressed(os.path.join(HERE, "lpips.npz"), **out)
    print("lpips.npz:", {k: (float(out[k]) if out[k].ndim == 0 else out[k].shape) for k in out if k.endswith(("_loss", "_val", "lin"))})


if __name__ == "__main__":
    main()
