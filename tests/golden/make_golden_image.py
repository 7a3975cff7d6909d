"""Golden vectors for the image branch glue: the REFERENCE's CustomEfficientNet
(P/occformer/backbones/efficientnet.py, imported unmodified from /root/reference through tests/refshim) on seeded
parameters and a small odd-sized image.  Run in the build container only:

    python tests/golden/make_golden_image.py        ->  tests/golden/efficientnet.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import paramgen, refshim  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_grad_enabled(False)


def reference_backbone(arch, out_indices):
    refshim.install()
    mod = refshim.ref("occformer.backbones.efficientnet")
    return mod.CustomEfficientNet(arch=arch, drop_path_rate=0.2, out_indices=out_indices, frozen_stages=0,
                                  norm_eval=False, with_cp=True, init_cfg=None)


def main():
    arrs = {}
    for arch, oi, hw in (("b0", (2, 3, 4, 5, 6), (67, 93)), ("b2", (1, 3, 6), (48, 80))):
        m = reference_backbone(arch, oi)
        sd = paramgen.fill_state_dict(m.state_dict(), 21)
        m.load_state_dict(sd)
        m.eval()
        x = paramgen.tensor("img." + arch, (2, 3) + hw, 21)
        outs = m(x)
        arrs[f"{arch}.seed"] = 21
        arrs[f"{arch}.param_checksum"] = paramgen.checksum(sd)
        arrs[f"{arch}.x"] = x.numpy()
        for i, o in zip(oi, outs):
            arrs[f"{arch}.out{i}"] = o.numpy()
        print(arch, [tuple(o.shape) for o in outs], "params %.2f M" % (sum(p.numel() for p in m.parameters()) / 1e6))
    # key list of the b7 the SemanticKITTI config builds (shapes via a checksum of the element counts)
    m7 = reference_backbone("b7", (2, 3, 4, 5, 6))
    sd7 = m7.state_dict()
    arrs["b7.keys"] = np.array(sorted(sd7))
    arrs["b7.numel"] = np.array([sd7[k].numel() for k in sorted(sd7)])
    np.savez_compressed(os.path.join(OUT, "efficientnet.npz"), **arrs)
    print("wrote efficientnet.npz (%.0f KiB)" % (os.path.getsize(os.path.join(OUT, "efficientnet.npz")) / 1024))


if __name__ == "__main__":
    main()
