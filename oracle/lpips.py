"""TEST INFRASTRUCTURE -- CPU restatement of the LPIPS term of the reference's training loss.  Imported only by tests.

The algorithm lives in the reference's vendored third-party package (3rd_Complete_HOSNeRF/third_parties/lpips, "LPIPS v0.1", `net='vgg'`):
  L:  third_parties/lpips/lpips.py            P:  third_parties/lpips/pretrained_networks.py        I:  third_parties/lpips/__init__.py
and is called by the step at src/model/mipnerf360/model.py:1664-1678 (M) on the unpacked patches.  The VGG-16 filters are torchvision's
ImageNet weights (a download, absent offline): every function here takes them as a `features.N.weight / bias` dict.
Pinned by tests/golden/lpips.npz = outputs of the reference's own class (tests/golden/make_golden_lpips.py)."""
import torch
import torch.nn.functional as F

VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
TAPS = (3, 8, 15, 22, 29)          # P:105-114: slices end behind features[3], [8], [15], [22], [29] = relu1_2 .. relu5_3
CHNS = (64, 128, 256, 512, 512)    # L:41
SHIFT = (-0.030, -0.088, -0.188)   # L:126 ScalingLayer
SCALE = (0.458, 0.448, 0.450)      # L:127


def vgg_taps(x, vgg):
    """P:117-131: the five ReLU outputs of torchvision's vgg16().features[0:30] on x [N,3,H,W]."""
    outs, idx, h = [], 0, x
    for v in VGG16_CFG:
        if idx > TAPS[-1]:
            break
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            idx += 1
            continue
        h = F.relu(F.conv2d(h, vgg[f"{idx}.weight"].to(h.dtype), vgg[f"{idx}.bias"].to(h.dtype), padding=1))
        idx += 2
        if idx - 1 in TAPS:
            outs.append(h)
    return outs


def normalize_tensor(f, eps=1e-10):
    """I:10-12."""
    return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True) + eps) + eps)


def lpips(pred, target, vgg, lin):
    """L:82-122 with lpips=True, spatial=False, version 0.1, eval mode (no dropout): pred / target [N,3,H,W] in [-1, 1];
    `lin`: the five 1x1 calibration weights concatenated (64+128+256+512+512).  Returns (val [N], per-layer [5, N])."""
    shift = torch.tensor(SHIFT, dtype=pred.dtype).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE, dtype=pred.dtype).view(1, 3, 1, 1)
    o0, o1 = vgg_taps((pred - shift) / scale, vgg), vgg_taps((target - shift) / scale, vgg)      # L:88-89
    res, off = [], 0
    for k in range(5):
        d = (normalize_tensor(o0[k]) - normalize_tensor(o1[k])) ** 2                             # L:93-94
        w = lin[off:off + CHNS[k]].to(pred.dtype).view(1, -1, 1, 1)
        off += CHNS[k]
        res.append((d * w).sum(1, keepdim=True).mean([2, 3]).reshape(-1))                        # L:100: lin (1x1 conv, no bias) + spatial mean
    return sum(res), torch.stack(res, 0)                                                          # L:107-109


def lpips_loss(pred_patches, target_patches, vgg, lin):
    """M:1673-1676: patches [N,P,P,3] in [0,1] -> mean over patches of LPIPS(2 x - 1, 2 y - 1)."""
    val, _ = lpips(2.0 * pred_patches.permute(0, 3, 1, 2) - 1.0, 2.0 * target_patches.permute(0, 3, 1, 2) - 1.0, vgg, lin)
    return val.mean()
