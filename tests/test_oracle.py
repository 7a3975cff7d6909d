"""The oracle itself: against the golden vectors generated from the reference, and (when
the reference tree is present) live against the reference's own Python."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import occformer_ref as O
from tests import paramgen, refshim, tinycfg
from tests.conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_restatement_vs_golden_bev_pool():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "build", "libbev_pool_ref.so"))
    g = golden("bev_pool")
    B, Z, X, Y = g["B"], g["Z"], g["X"], g["Y"]
    order, geom, starts, lengths = O.bev_pool_intervals(g["coords"].long(), B, Z, X, Y)
    x = g["feats"][order].contiguous()
    n, c = x.shape
    out = torch.empty(B, Z, X, Y, c)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.bev_pool_ref_forward(B, Z, X, Y, n, c, starts.numel(), p(x), p(geom), p(starts), p(lengths), p(out))
    assert torch.allclose(out.permute(0, 4, 1, 2, 3), g["out"], atol=1e-5)
    assert torch.allclose(out, O.bev_pool_forward(x, geom, starts, lengths, B, Z, X, Y), atol=1e-5)
    og = paramgen.tensor("og", (B, Z, X, Y, c), 0)
    xg = torch.empty(n, c)
    lib.bev_pool_ref_backward(B, Z, X, Y, n, c, starts.numel(), p(og), p(geom), p(starts), p(lengths), p(xg))
    assert torch.equal(xg, O.bev_pool_backward(og, geom, starts, lengths))


def test_bev_pool_hand_computed():
    """5 points -> 3 voxels incl. the (-1,0)->0 truncation case and an out-of-range point
    (SURVEY.md §8c known-answer)."""
    dx, bx, nx = O.grid_constants([0, 2, 1.0], [0, 2, 1.0], [0, 1, 1.0])
    geom = torch.tensor([[[[[[0.5, 0.5, 0.5]]]], [[[[0.6, 0.4, 0.2]]]], [[[[-0.5, 0.5, 0.5]]]],   # trunc -> 0
                          [[[[1.5, 1.5, 0.5]]]], [[[[2.5, 0.5, 0.5]]]]]])                             # out of range
    geom = geom.view(1, 5, 1, 1, 1, 3)
    coords, kept = O.lss_voxel_coords(geom, dx, bx, nx)
    assert kept.tolist() == [True, True, True, True, False]
    depth = torch.ones(5, 1, 1, 1)
    feat = torch.tensor([1.0, 10.0, 100.0, 1000.0, 1e4]).view(5, 1, 1, 1)
    out = O.lift_splat(depth, feat, geom, dx, bx, nx)     # [1,1,2,2,1]
    assert out[0, 0, 0, 0, 0] == 111.0 and out[0, 0, 1, 1, 0] == 1000.0 and float(out.sum()) == 1111.0


def test_oracle_vs_golden_modules():
    model, meta = tinycfg.tiny_nusc()
    g = golden("view_transformer")
    cams = (g["rots"], g["trans"], g["intrins"], g["post_rots"], g["post_trans"], g["bda"])
    assert torch.allclose(O.lss_geometry(O.make_frustum((64, 176), 16, [2.0, 10.0, 0.5]), *cams), g["geom"], atol=1e-5)
    assert torch.equal(O.mlp_input_from_cameras(*cams), g["mlp_input"])
    t = golden("tables")
    assert torch.allclose(O.sine_pos_enc_3d((5, 4, 3), 32), t["pos_enc_5x4x3_f32"], atol=1e-6)


@pytest.mark.skipif(not refshim.available(), reason="reference tree not present (GPU box)")
def test_oracle_live_vs_reference_encoder_block():
    refshim.install()
    refshim.ref("occformer.backbones.occnet")
    from mmdet.models.builder import MODELS as REF
    model, meta = tinycfg.tiny_nusc()
    enc = refshim.build_from_cfg(refshim.ConfigDict(model["img_bev_encoder_backbone"]), REF)
    sd = paramgen.fill_state_dict(enc.state_dict(), 9)
    enc.load_state_dict(sd)
    enc.eval()
    x = paramgen.tensor("live", (1, 32, 9, 12, 2), 9, 0.5)
    with torch.no_grad():
        ref = enc.layers[0](x)
        out = O.dualpath_block({"e." + k: v for k, v in sd.items()}, "e.layers.0.0.", x, 1, False, 8)
        out = O.dualpath_block({"e." + k: v for k, v in sd.items()}, "e.layers.0.1.", out, 1, True, 8)
    assert torch.allclose(out, ref, atol=1e-5)


def test_forced_gates_mechanism():
    """oracle.occformer_ref.forced_gates (test infrastructure of the gradient comparisons): with a unit's OWN gates the
    forced ReLU is F.relu in value and gradient; a gate forced the other way is counted, its pre-activation reported,
    and the gradient follows the forced gate (what makes two implementations differentiate the same function)"""
    from oracle import occformer_ref as O
    z = torch.tensor([[-0.5, 2e-7, 0.25, -3e-7], [1.5, -1.0, 4e-7, 0.75]], requires_grad=True)
    own = (z.detach() > 0)
    with O.forced_gates([own]) as g:
        y = O._relu_gated(z)
    assert g.i == 1 and g.flipped == 0 and g.units == z.numel()
    assert torch.equal(y, torch.relu(z))
    (gy,) = torch.autograd.grad(y.sum(), z)
    assert torch.equal(gy, own.float())
    other = own.clone()
    other[0, 1] = False            # the other implementation saw 2e-7 as <= 0
    other[0, 3] = True             # ... and -3e-7 as > 0
    with O.forced_gates([other]) as g:
        y2 = O._relu_gated(z)
        with pytest.raises(AssertionError):
            O._relu_gated(z)       # more ReLUs evaluated than gates recorded
    assert g.flipped == 2 and abs(g.max_abs_z - 3e-7) < 1e-12 and g.max_rel_z < 1e-6
    (gy2,) = torch.autograd.grad(y2.sum(), z)
    assert torch.equal(gy2, other.float())
    assert float((y2 - torch.relu(z)).abs().max()) <= 3.1e-7      # the forward moves by the flipped pre-activations only
    assert O._GATES is None and torch.equal(O._relu_gated(z), torch.relu(z))      # outside the context: plain ReLU
    # the three classes against the three levels: heavy units are forced at every level, the BEV ASPP's maps ("bev") from
    # "heavy+bev" on, every other ReLU (False) at "all" only -- and the product's tap (noise.gates_wanted) agrees
    from occformer_amd import noise
    for level, taken in (("heavy", (True,)), ("heavy+bev", (True, "bev")), ("all", (True, "bev", False))):
        for cls in (True, "bev", False):
            with O.forced_gates([other], level=level) as g:
                O._relu_gated(z, cls)
            assert g.i == (1 if cls in taken else 0), (level, cls)
            noise.record_gates(level)
            try:
                assert noise.gates_wanted(cls) == (cls in taken), (level, cls)
            finally:
                noise.record_gates(False)


def test_synthetic_rig_has_no_noise_amplifying_camera_columns():
    """DepthNet normalises the 27 camera scalars with a train-mode BatchNorm1d over the cameras
    (ViewTransformerLSSBEVDepth.py:453,489).  A column that is constant but whose mean does not round back to the value
    ((x1 + .. + x6) / 6 != x in fp32) comes out as rounding noise / sqrt(eps): +-0.02 for fx = 557 repeated six times --
    backend-dependent, it put the camera-MLP gradients of GPU and CPU 3e-2 apart.  The
    bench / parity rig must not contain one: every column either varies or is reproduced exactly by its own mean."""
    from occformer_amd import configs
    from oracle import occformer_ref as O
    for name in ("nusc_r50_200", "nusc_r50_ref128", "nusc_r101"):
        _, meta = configs.workload(name)
        inp, _, _ = configs.synthetic_sample(meta, "cpu", seed=0)
        m = O.mlp_input_from_cameras(*inp[1:7])[0]                       # [6, 27]
        mean = m.sum(0) / m.shape[0]
        for c in range(m.shape[1]):
            col = m[:, c]
            constant = bool((col == col[0]).all())
            if constant:
                assert float(mean[c]) == float(col[0]), (name, c, float(col[0]), float(mean[c]))
            else:
                assert float(col.std()) > 1e-4 * max(1.0, float(col.abs().max())), (name, c)
