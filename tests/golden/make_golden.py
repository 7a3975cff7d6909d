"""Generate golden vectors from the REFERENCE (imported unmodified from /root/reference
through tests/refshim) and check the oracle restatement against it.

Run in the build container only:  python tests/golden/make_golden.py
Writes tests/golden/*.npz (small) -- inputs, reference outputs and a checksum of the
seeded parameters (tests/paramgen.py regenerates the parameters themselves).
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


def build(cfg, **default):
    from mmdet.models.builder import MODELS
    return refshim.build_from_cfg(refshim.ConfigDict(cfg), MODELS, default or None)


def load_filled(mod, seed):
    sd = paramgen.fill_state_dict(mod.state_dict(), seed)
    mod.load_state_dict(sd)
    mod.eval()
    return sd


def maxerr(a, b):
    return float((a - b).abs().max()), float(b.abs().max())


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KiB)")


def main():
    refshim.install()
    refshim.ref("occformer.image2bev.ViewTransformerLSSVoxel")
    refshim.ref("occformer.backbones.occnet")
    refshim.ref("occformer.necks.multiscale_deformattn_3d")
    refshim.ref("occformer.mask2former.positional_encodings.positional_encoding")
    refshim.ref("occformer.mask2former.mask2former_nusc_occ")
    model, meta = tinycfg.tiny_nusc()
    B, N = 2, 3
    report = {}

    # ---------------- view transformer (rows 1-7)
    vt = build(model["img_view_transformer"])
    sd_vt = load_filled(vt, 1)
    cams = paramgen.camera_rig(B, N, *meta["input_size"], meta["focal"], seed=1)
    x = paramgen.tensor("img_feats", (B, N, 32, meta["fH"], meta["fW"]), 1)
    mlp = vt.get_mlp_input(*cams)
    vox_ref, depth_ref = vt([x, *cams, mlp])
    geom_ref = vt.get_geometry(*cams)
    sd_p = {"img_view_transformer." + k: v for k, v in sd_vt.items()}
    vox_o, depth_o = O.view_transformer(sd_p, "img_view_transformer.", x, cams, meta["D"], meta["C"])
    report["vt.mlp_input"] = maxerr(O.mlp_input_from_cameras(*cams), mlp)
    report["vt.geometry"] = maxerr(O.lss_geometry(sd_vt["frustum"], *cams), geom_ref)
    report["vt.depth"] = maxerr(depth_o, depth_ref)
    report["vt.voxel"] = maxerr(vox_o, vox_ref)
    coords, kept = O.lss_voxel_coords(geom_ref, sd_vt["dx"], sd_vt["bx"], sd_vt["nx"])
    print("  kept", int(kept.sum()), "of", kept.numel(), " nonzero voxels",
          int((vox_ref.abs().sum(1) > 0).sum()))
    save("view_transformer", x=x, rots=cams[0], trans=cams[1], intrins=cams[2], post_rots=cams[3],
         post_trans=cams[4], bda=cams[5], mlp_input=mlp, geom=geom_ref, depth=depth_ref,
         voxel=vox_ref, coords=coords.int(), kept=kept, param_checksum=paramgen.checksum(sd_vt),
         seed=1)

    # ---------------- bev_pool op alone, incl. edge cases (row 5-7)
    from mmdet3d.ops.bev_pool import bev_pool as ref_bev_pool
    feats = paramgen.tensor("bp_feats", (300, 8), 2)
    g = torch.Generator().manual_seed(5)
    cc = torch.stack((torch.randint(0, 5, (300,), generator=g), torch.randint(0, 4, (300,), generator=g),
                      torch.randint(0, 3, (300,), generator=g), torch.randint(0, 2, (300,), generator=g)), 1)
    out_ref = ref_bev_pool(feats, cc, 2, 3, 5, 4)          # [B, C, Z, X, Y]
    order, gg, st, ln = O.bev_pool_intervals(cc, 2, 3, 5, 4)
    out_o = O.bev_pool_forward(feats[order], gg, st, ln, 2, 3, 5, 4).permute(0, 4, 1, 2, 3)
    report["bev_pool"] = maxerr(out_o, out_ref)
    save("bev_pool", feats=feats, coords=cc.int(), out=out_ref, B=2, Z=3, X=5, Y=4)

    # ---------------- encoder (rows 8-11)
    enc = build(model["img_bev_encoder_backbone"])
    sd_enc = load_filled(enc, 2)
    vin = paramgen.tensor("voxel_in", (B, meta["C"], *meta["grid"]), 2, 0.5)
    enc_ref = enc(vin)
    sd_p = {"e." + k: v for k, v in sd_enc.items()}
    enc_o = O.occupancy_encoder(sd_p, "e.", vin, groups=meta["groups"])
    for i, (a, b) in enumerate(zip(enc_o, enc_ref)):
        report[f"encoder.out{i}"] = maxerr(a, b)
    # one block, non-multiple-of-7 + shifted, checked separately with small tensors
    blk = enc.layers[0][1]
    bx_in = paramgen.tensor("blk_in", (1, 32, 10, 9, 3), 3, 0.5)
    blk_ref = blk(bx_in)
    blk_o = O.dualpath_block(sd_p, "e.layers.0.1.", bx_in, 1, True, meta["groups"])
    report["encoder.block_shift"] = maxerr(blk_o, blk_ref)
    save("encoder", x=vin, **{f"out{i}": t for i, t in enumerate(enc_ref)}, blk_in=bx_in,
         blk_out=blk_ref, param_checksum=paramgen.checksum(sd_enc), seed=2)

    # ---------------- pixel decoder (rows 12-14)
    pd = build(model["img_bev_encoder_neck"])
    sd_pd = load_filled(pd, 3)
    pd_ref = pd(enc_ref)
    sd_p = {"n." + k: v for k, v in sd_pd.items()}
    pd_o = O.pixel_decoder(sd_p, "n.", enc_ref, num_layers=meta["pd_layers"], groups=meta["groups"])
    for i, (a, b) in enumerate(zip(pd_o, pd_ref)):
        report[f"pixel_decoder.out{i}"] = maxerr(a, b)
    # inputs are encoder.npz's out0..3
    save("pixel_decoder", **{f"out{i}": t for i, t in enumerate(pd_ref)}, param_checksum=paramgen.checksum(sd_pd), seed=3)

    # ---------------- head (rows 15-17)
    head = build(model["pts_bbox_head"], train_cfg=None, test_cfg=None)
    sd_h = load_filled(head, 4)
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"]) for _ in range(B)]
    lo = torch.tensor(meta["pc_range"][:3])
    hi = torch.tensor(meta["pc_range"][3:])
    pts = [paramgen.uniform(f"pts{b}", (257, 3), 4) * (hi - lo) * 1.1 + lo - 0.05 * (hi - lo) for b in range(B)]
    cls_ref, mask_ref = head(pd_ref, metas)
    res_ref = head.simple_test(pd_ref, metas, points=pts)
    sd_p = {"h." + k: v for k, v in sd_h.items()}
    cls_o, mask_o = O.mask2former_head(sd_p, "h.", pd_ref, heads=meta["heads"], num_layers=meta["dec_layers"])
    res_o = O.head_simple_test(sd_p, "h.", pd_ref, meta["occ_size"], pts, meta["pc_range"],
                               heads=meta["heads"], num_layers=meta["dec_layers"])
    for i in (0, 1, meta["dec_layers"]):
        report[f"head.cls{i}"] = maxerr(cls_o[i], cls_ref[i])
        report[f"head.mask{i}"] = maxerr(mask_o[i], mask_ref[i])
    report["head.output_voxels"] = maxerr(res_o["output_voxels"], res_ref["output_voxels"][0])
    report["head.output_points"] = maxerr(res_o["output_points"], res_ref["output_points"])
    # inputs are pixel_decoder.npz's out0..3
    save("head", cls_last=cls_ref[-1], cls0=cls_ref[0],
         mask_last=mask_ref[-1], mask0=mask_ref[0], output_voxels=res_ref["output_voxels"][0],
         output_points=res_ref["output_points"], pts0=pts[0], pts1=pts[1],
         param_checksum=paramgen.checksum(sd_h), seed=4)

    # ---------------- pos-enc / index tables (known-answer helpers)
    pe_mod = refshim.ref("occformer.mask2former.positional_encodings.positional_encoding")
    pe = pe_mod.SinePositionalEncoding3D(num_feats=32, normalize=True)
    pe_ref = pe(torch.zeros(1, 5, 4, 3, dtype=torch.bool))[0]
    report["pos_enc"] = maxerr(O.sine_pos_enc_3d((5, 4, 3), 32), pe_ref)
    wa = refshim.ref("occformer.backbones.modules.window_attention")
    w = wa.WindowMSA(32, 1, (7, 7))
    report["rel_pos_index"] = maxerr(O.rel_pos_index(7).float(), w.relative_position_index.float())
    save("tables", pos_enc_5x4x3_f32=pe_ref, rel_pos_index=w.relative_position_index.int())

    print("\noracle vs reference (max abs err, max |ref|):")
    bad = 0
    for k, (e, m) in report.items():
        flag = "" if e <= 1e-4 * max(1.0, m) else "   <-- MISMATCH"
        bad += bool(flag)
        print(f"  {k:28s} {e:.3e}  /  {m:.3e}{flag}")
    if bad:
        raise SystemExit(f"{bad} mismatches")


if __name__ == "__main__":
    main()
