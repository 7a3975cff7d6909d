"""Golden vectors for the SemanticKITTI form of the view transformer (BASELINE configs 0-1: one camera, 4x4
intrinsics with a translation column, 4x4 BEV augmentation, cam_channels = 33), produced by the REFERENCE's
ViewTransformerLiftSplatShootVoxel imported unmodified through tests/refshim; the oracle is checked against it.

    python tests/golden/make_golden_kitti.py        ->  tests/golden/view_transformer_kitti.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import paramgen, refshim, tinycfg  # noqa: E402
from oracle import occformer_ref as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_grad_enabled(False)


def kitti_vt_cfg():
    model, meta = tinycfg.tiny_nusc(ncams=1)
    cfg = dict(model["img_view_transformer"], cam_channels=33)
    return cfg, meta


def main():
    refshim.install()
    refshim.ref("occformer.image2bev.ViewTransformerLSSVoxel")
    from mmdet.models.builder import MODELS
    cfg, meta = kitti_vt_cfg()
    vt = refshim.build_from_cfg(refshim.ConfigDict(cfg), MODELS, None)
    sd = paramgen.fill_state_dict(vt.state_dict(), 31)
    vt.load_state_dict(sd)
    vt.eval()
    B, N = 2, 1
    cams = paramgen.camera_rig(B, N, *meta["input_size"], meta["focal"], seed=31, kitti=True)
    x = paramgen.tensor("kitti_feats", (B, N, 32, meta["fH"], meta["fW"]), 31)
    mlp = vt.get_mlp_input(*cams)
    vox, depth = vt([x, *cams, mlp])
    sd_p = {"vt." + k: v for k, v in sd.items()}
    vox_o, depth_o = O.view_transformer(sd_p, "vt.", x, cams, meta["D"], meta["C"])
    err = lambda a, b: (float((a - b).abs().max()), float(b.abs().max()))
    print("oracle vs reference: mlp_input", err(O.mlp_input_from_cameras(*cams), mlp), "depth", err(depth_o, depth),
          "voxel", err(vox_o, vox))
    assert mlp.shape[-1] == 33
    assert err(O.mlp_input_from_cameras(*cams), mlp)[0] < 1e-5 and err(depth_o, depth)[0] < 1e-4 and \
        err(vox_o, vox)[0] < 1e-4
    np.savez_compressed(os.path.join(OUT, "view_transformer_kitti.npz"), x=x.numpy(), rots=cams[0].numpy(),
                        trans=cams[1].numpy(), intrins=cams[2].numpy(), post_rots=cams[3].numpy(),
                        post_trans=cams[4].numpy(), bda=cams[5].numpy(), mlp_input=mlp.numpy(), depth=depth.numpy(),
                        voxel=vox.numpy(), param_checksum=paramgen.checksum(sd), seed=31)
    print("wrote view_transformer_kitti.npz (%.0f KiB)" %
          (os.path.getsize(os.path.join(OUT, "view_transformer_kitti.npz")) / 1024))


def main_head():
    """SemanticKITTI head at test time: Mask2FormerOccHead.simple_test (20 classes, no LiDAR branch) on the pixel
    decoder vectors of tests/golden/pixel_decoder.npz"""
    from tests.golden.make_golden_train import kitti_head_cfg
    refshim.install()
    for n in ("losses.dice_loss", "assigners.match_costs.match_cost", "assigners.mask_hungarian_assigner",
              "samplers.mask_pseudo_sampler"):
        try:
            refshim.ref("occformer.mask2former." + n)
        except Exception:
            pass
    refshim.ref("occformer.mask2former.positional_encodings.positional_encoding")
    kitti = refshim.ref("occformer.mask2former.mask2former_occ")
    model, meta = tinycfg.tiny_nusc()
    kc = refshim.ConfigDict(kitti_head_cfg(model))
    args = {k: v for k, v in kc.items() if k != "type"}
    head = kitti.Mask2FormerOccHead(**args, train_cfg=None, test_cfg=None)
    sd = paramgen.fill_state_dict(head.state_dict(), 41)
    head.load_state_dict(sd)
    head.eval()
    z = np.load(os.path.join(OUT, "pixel_decoder.npz"))
    feats = [torch.from_numpy(z[f"out{i}"]) for i in range(4)]
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"]) for _ in range(feats[0].shape[0])]
    cls, masks = head(feats, metas)
    res = head.simple_test(feats, metas)
    sd_p = {"h." + k: v for k, v in sd.items()}
    res_o = O.head_simple_test(sd_p, "h.", feats, meta["occ_size"], None, None, heads=meta["heads"],
                               num_layers=meta["dec_layers"])
    e = float((res_o["output_voxels"] - res["output_voxels"][0]).abs().max())
    print("oracle vs reference: kitti output_voxels", e, "classes", res["output_voxels"][0].shape[1])
    assert e < 1e-4 and res["output_voxels"][0].shape[1] == 20 and res["output_points"] is None
    np.savez_compressed(os.path.join(OUT, "head_kitti.npz"), cls_last=cls[-1].numpy(), mask_last=masks[-1].numpy(),
                        output_voxels=res["output_voxels"][0].numpy(), param_checksum=paramgen.checksum(sd), seed=41)
    print("wrote head_kitti.npz (%.0f KiB)" % (os.path.getsize(os.path.join(OUT, "head_kitti.npz")) / 1024))


if __name__ == "__main__":
    main()
    main_head()
