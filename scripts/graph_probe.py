"""Can the static-shape part of the forward (dual-path encoder -> pixel decoder -> occupancy decoder -> output volume)
be captured in a HIP graph, and what does replay save over eager launches?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occformer_amd  # noqa
from occformer_amd import configs
from occformer_amd.registry import build_model

device = torch.device("cuda", 0)
torch.manual_seed(0)
cfg, meta = configs.workload("nusc_r50_200")
model = build_model(cfg).to(device).eval()
img_inputs, metas, points = configs.synthetic_sample(meta, device, seed=0)

with torch.no_grad():
    x = model.image_encoder(img_inputs[0])
    rots, trans, intrins, post_rots, post_trans, bda = img_inputs[1:7]
    mlp_input = model.img_view_transformer.get_mlp_input(rots, trans, intrins, post_rots, post_trans, bda)
    vox, depth = model.img_view_transformer([x, rots, trans, intrins, post_rots, post_trans, bda, mlp_input])

def tail(v):
    feats = model.bev_encoder(v)
    if not isinstance(feats, list):
        feats = [feats]
    res = model.pts_bbox_head.simple_test(feats, metas, points=points)
    return res["output_voxels"][0], res["output_points"]

def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    ref = tail(vox)
    print("eager tail ms:", round(timeit(lambda: tail(vox)), 3), flush=True)
    static_in = vox.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            tail(static_in)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = tail(static_in)
    except Exception as e:
        print("capture failed:", type(e).__name__, str(e)[:400], flush=True)
        sys.exit(0)
    def replay():
        static_in.copy_(vox)
        g.replay()
    replay()
    torch.cuda.synchronize()
    print("graph output max abs diff vs eager:", float((out[0] - ref[0]).abs().max()), float((out[1] - ref[1]).abs().max()))
    print("graph tail ms:", round(timeit(replay), 3), flush=True)
