"""micro-benchmarks of the three heaviest backward kernels at the 200-grid sizes (HIP events, 5 repeats)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops()
g = torch.Generator().manual_seed(0)
what = sys.argv[1:] or ["wgrad", "window", "msda", "dgrad"]

def timeit(name, fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:60s} {e0.elapsed_time(e1) / n:8.3f} ms", flush=True)

if "wgrad" in what or "wgrad192" in what or "wgrad128" in what:
    for C in ((128, 192) if "wgrad" in what else (192,) if "wgrad192" in what else (128,)):
        dy = torch.randn(1, 200, 200, 16, C, generator=g).cuda()
        x = torch.randn(1, 200, 200, 16, C, generator=g).cuda()
        timeit(f"conv3d_wgrad 3^3 {C}->{C} 200x200x16", lambda: ops.conv3d_wgrad(dy, x, (3, 3, 3), 1, 1))
    if "wgrad" in what:
        dy = torch.randn(680000, 384, generator=g).cuda()
        x = torch.randn(680000, 128, generator=g).cuda()
        timeit("linear_wgrad 680000 x (384, 128)", lambda: ops.linear_wgrad(dy, x))
if "window" in what:
    B, X, Y, S, heads = 1, 200, 200, 17, 4
    C = heads * 32
    n = B * X * Y * S
    qkv = torch.randn(n, 3 * C, generator=g).cuda()
    qb = torch.randn(3 * C, generator=g).cuda()
    tab = torch.randn(169, heads, generator=g).cuda()
    out = ops.window_attention(qkv, qb, tab, B, X, Y, S, heads, 3)
    dout = torch.randn(n, C, generator=g).cuda()
    timeit("window_attention_backward stage 0 (shift 3)",
           lambda: ops.window_attention_backward(qkv, qb, tab, out, dout, B, X, Y, S, heads, 3))
if "msda" in what:
    shapes = [(25, 25, 2), (50, 50, 4), (100, 100, 8)]
    Nq = sum(a * b * c for a, b, c in shapes)
    value = torch.randn(1, Nq, 192, generator=g).cuda()
    offs = (torch.randn(1, Nq, 8 * 3 * 4 * 3, generator=g) * 2).cuda()
    lg = torch.randn(1, Nq, 96, generator=g).cuda()
    dout = torch.randn(1, Nq, 192, generator=g).cuda()
    timeit("msda3d_backward 91250 queries, 8 heads x 24",
           lambda: ops.msda3d_backward(value, offs, lg, dout, shapes, 8, 4))

if "xattn" in what:
    heads, Q = 6, 100
    for L in (80000, 10000, 1250, 100):
        qq = torch.randn(1, Q, 192, generator=g).cuda()
        kk = torch.randn(1, L, 192, generator=g).cuda()
        vv = torch.randn(1, L, 192, generator=g).cuda()
        bl = (torch.rand(1, Q, L, generator=g) < 0.5).to(torch.uint8).cuda() if L > 100 else None
        ro = torch.ones(Q, dtype=torch.int32).cuda() if L > 100 else None
        out = ops.masked_attention(qq, kk, vv, heads, bl, ro)
        do = torch.randn(1, Q, 192, generator=g).cuda()
        timeit(f"masked_attention forward Q=100 L={L}", lambda: ops.masked_attention(qq, kk, vv, heads, bl, ro))
        timeit(f"masked_attention_backward Q=100 L={L}", lambda: ops.masked_attention_backward(qq, kk, vv, heads, out, do, bl, ro))

if "dgrad" in what:
    import torch.nn.functional as F
    for (dims, Cin, Cout) in (((200, 200, 16), 128, 256), ((100, 100, 8), 256, 512), ((50, 50, 4), 512, 1024)):
        w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.02
        wt = w.permute(1, 2, 3, 4, 0).reshape(Cin, -1).contiguous().cuda()
        sp = ops.split_bf16(wt)
        dy = torch.randn(1, dims[0] // 2, dims[1] // 2, dims[2] // 2, Cout, generator=g).cuda()
        timeit(f"conv3d_dgrad 3^3 stride 2 {Cin}->{Cout} input {dims}",
               lambda: ops.conv3d_dgrad(dy, sp, (1, *dims, Cin), (3, 3, 3), 2, 1))

if "conv" in what:
    for C in (128, 192):
        x = torch.randn(1, 200, 200, 16, C, generator=g).cuda()
        w = (torch.randn(C, 27 * C, generator=g) * 0.02).cuda()
        sp = ops.split_bf16(w)
        timeit(f"conv3d 3^3 {C}->{C} 200x200x16 (halo kernel)", lambda: ops.conv3d(x, w, (3, 3, 3), 1, 1, (1, 1, 1), None, w_split=sp))
if "psb" in what:
    # the matched rows' point-logit gradient scattered voxel-major: one [V, 160] buffer shared by the ten prediction
    # sets (13 columns each) vs a [V, 16] buffer per set, vs the row-major [13, V] volumes
    X, Y, Z, n, P = 200, 200, 16, 13, 12544
    V = X * Y * Z
    pts = torch.rand(n, P, 3, generator=g).cuda()
    dout = torch.randn(n, 1, P, generator=g).cuda()
    big = torch.zeros(V, 160, device="cuda")
    small = [torch.zeros(V, 16, device="cuda") for _ in range(10)]
    k = [0]

    def f_big():
        ops.point_sample_3d_backward(dout, pts, (n, 1, X, Y, Z), False, "border", voxel_major_cols=160, out=big,
                                     col0=16 * (k[0] % 10))
        k[0] += 1

    def f_small():
        ops.point_sample_3d_backward(dout, pts, (n, 1, X, Y, Z), False, "border", voxel_major_cols=16,
                                     out=small[k[0] % 10], col0=0)
        k[0] += 1
    timeit("point_sample_3d_backward 13 x 12544 -> [V, 160] cols", f_big, n=10)
    timeit("point_sample_3d_backward 13 x 12544 -> [V, 16] per set", f_small, n=10)
    timeit("point_sample_3d_backward 13 x 12544 -> [13, V] (+ zero fill)",
           lambda: ops.point_sample_3d_backward(dout, pts, (n, 1, X, Y, Z), False, "border"), n=10)
    timeit("torch.cat of ten [V, 16] -> [V, 160]", lambda: torch.cat(small, 1), n=5)
    timeit("zero fill [V, 160]", lambda: big.zero_(), n=5)
if "gn" in what:
    # the streaming normalisation / elementwise passes at the full-resolution shapes, with the bytes they actually move
    def gbs(name, fn, nbytes, n=8):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{name:78s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.0f} GB/s", flush=True)
    for C, tokens, relu, res in ((128, True, True, False), (192, False, True, False), (192, False, False, True)):
        B, X, Y, Z, G = 1, 200, 200, 16, 32
        x = torch.randn(B, X, Y, Z, C, generator=g).cuda()
        gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
        dy = torch.randn(B, X, Y, Z + 1 if tokens else Z, C, generator=g).cuda()
        nb = x.numel() * 4
        st = ops.groupnorm_stats(x, G, 1e-5)
        gbs(f"groupnorm_stats [{X},{Y},{Z},{C}] (1 pass)", lambda: ops.groupnorm_stats(x, G, 1e-5), nb)
        r = torch.randn_like(x) if res else None
        gbs(f"groupnorm_apply relu={relu} tokens={tokens} res={res} ({3 if res else 2} passes)",
            lambda: ops.groupnorm_apply(x, st, gamma, beta, G, relu, tokens, r), nb * (3 if res else 2))
        for form in ("0", "1"):
            os.environ["OCCF_GNB_APPLY_ROWS"] = form
            gbs(f"groupnorm_backward relu={relu} tokens={tokens} dres={res} (OCCF_GNB_APPLY_ROWS={form}; {6 if res else 5} passes)",
                lambda: ops.groupnorm_backward(x, st, gamma, beta, dy, G, relu, tokens, want_residual=res),
                nb * (6 if res else 5))
    M, C = 680000, 128
    x = torch.randn(M, C, generator=g).cuda()
    dy = torch.randn(M, C, generator=g).cuda()
    w, b = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    gbs("layernorm [680000,128] (2 passes)", lambda: ops.layernorm(x, w, b, 1e-5), M * C * 8)
    gbs("layernorm_backward [680000,128] (3 passes)", lambda: ops.layernorm_backward(x, w, dy, 1e-5), M * C * 12)
    sc = torch.rand(17, generator=g).cuda()
    gbs("droppath identity + branch [680000,128] (3 passes)", lambda: ops.droppath(x, dy, sc, 40000, 17), M * C * 12)
    gbs("act_backward GELU [680000,128] (3 passes)", lambda: ops.act_backward(x, dy, 2), M * C * 12)
    tok = torch.randn(1, 200, 200, 17, C, generator=g).cuda()
    bev = torch.randn(1, 200, 200, C, generator=g).cuda()
    ident = torch.randn(1, 200, 200, 16, C, generator=g).cuda()
    cw, cb = torch.randn(C, generator=g).cuda(), torch.randn(1, generator=g).cuda()
    gbs("dualpath_combine [200,200,16,128] (3 passes)", lambda: ops.dualpath_combine(tok, bev, cw, cb, ident), ident.numel() * 12)
if "psample" in what:
    # importance sampling of the head loss at the metric's sizes: S = 10 sets x G = 17 matched masks, 150 528 candidate
    # points per set, sampled from the channel-major logits [S, G, X, Y, Z] (what runs today) against the same logits
    # voxel-major [V, 224] (20 columns per set) through the channels-last sampler
    S, G, P3 = 10, 17, 150528
    dense = torch.randn(S, G, 200, 200, 16, generator=g).cuda()
    cand = torch.rand(S, P3, 3, generator=g).cuda()
    timeit("point_sample_3d dense [10,17,200,200,16] at [10,150528] points", lambda: ops.point_sample_3d(dense, cand, False, "border"))
    vm = torch.randn(200 * 200 * 16, 224, generator=g).cuda()
    def vm_sample():
        return [ops.point_sample_tokens(vm[:, 20 * s:20 * s + 20], (200, 200, 16), cand[s], False, "border") for s in range(S)]
    timeit("point_sample_tokens x10 on voxel-major [640000, 224] column blocks of 20", vm_sample)
    feat = torch.randn(640000, 192, generator=g).cuda()
    w = torch.randn(224, 192, generator=g).cuda() * 0.05
    sp = ops.split_bf16(w)
    timeit("linear feat [640000,192] x rows [224,192] -> voxel-major logits", lambda: ops.linear(feat, w, None, w_split=sp))
    rows = torch.randn(170, 192, generator=g).cuda()
    fsp = ops.split_bf16(feat)
    timeit("linear rows [170,192] x feat [640000,192] -> channel-major logits (today)", lambda: ops.linear(rows, feat, None, w_split=fsp, allow_small=False))
    pts = torch.rand(G, 50176, 3, generator=g).cuda()
    timeit("point_sample_3d one set's rows [17,1,200,200,16] at their own [17,50176] points", lambda: ops.point_sample_3d(dense[0].unsqueeze(1).contiguous(), pts, False, "border"))
