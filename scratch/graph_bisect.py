"""Which pieces of the step survive HIP graph capture?  Each stage runs in its own subprocess (a crash in
hipStreamEndCapture kills the process)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGES = ["enc_loss", "enc_loss_opt", "step_tiny"]

def capture(fn, warm=3):
    import torch, gc
    gc.collect()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm): fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gc.collect()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = fn()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    return out

def run(stage):
    import torch
    from deepcoro_clip_amd import kernels as K, functional as Fn
    dev = "cuda"
    if stage == "gemm_ring":
        a = torch.randn(4096, 512, device=dev).bfloat16(); b = torch.randn(1024, 512, device=dev).bfloat16(); o = torch.empty(4096, 1024, device=dev, dtype=torch.bfloat16)
        capture(lambda: K.gemm(a, b, out=o))
    elif stage == "gemm_small":
        K.GEMM_VARIANT = 6 << 8
        a = torch.randn(256, 512, device=dev).bfloat16(); b = torch.randn(256, 512, device=dev).bfloat16(); o = torch.empty(256, 256, device=dev, dtype=torch.bfloat16)
        capture(lambda: K.gemm(a, b, out=o))
    elif stage == "ln":
        x = torch.randn(4096, 96, device=dev).bfloat16(); g = torch.ones(96, device=dev); b = torch.zeros(96, device=dev)
        dg = torch.zeros(96, device=dev); db = torch.zeros(96, device=dev)
        def f():
            y, m, r = K.layernorm_fwd(x, g, b, 1e-6)
            K.layernorm_bwd(y, x, m, r, g, dg, db)
        capture(f)
    elif stage == "pool":
        geo = K.PoolGeom(2, 2, 96, (4, 14, 14), (3, 3, 3), (1, 2, 2), (1, 1, 1), True)
        x = torch.randn(2, geo.n_in, 2, 96, device=dev).bfloat16(); w = torch.randn(96, 27, device=dev); dw = torch.zeros(96, 27, device=dev)
        dx = torch.empty_like(x)
        def f():
            y, _ = K.pool_fwd(x, geo, w)
            K.pool_bwd_data(y, dx, geo, w)
            K.dwconv_bwd_weight(y, x, geo, dw)
        capture(f)
    elif stage.startswith("enc_"):
        import __graft_entry__ as ge
        runner, _, (videos, ids, mask) = ge.make_tiny_setup(torch.bfloat16)
        videos, ids, mask = videos.cuda(), ids.cuda(), mask.cuda()
        runner._graph_host = torch.zeros(10, dtype=torch.float32).pin_memory()
        ve, te = runner.video_encoder, runner.text_encoder
        params = [p for p in list(ve.parameters()) + list(te.parameters()) + [runner.log_temp]]
        from deepcoro_clip_amd.optim import GradClipper
        clipper = GradClipper([p for p in params if p.requires_grad])
        def f():
            for p in params: p.grad = None
            if stage == "enc_sq_nostream":
                v = ve(videos); t = te(ids, mask)
            else:
                v, t = runner._encode(videos, ids, mask)
            if stage.startswith("enc_sq"):
                loss = v.float().pow(2).sum() + t.float().pow(2).sum()
            else:
                loss = runner._loss(v, t, None, None, None, {})
            loss.backward()
            if stage in ("enc_loss_clip", "enc_loss_opt"):
                clipper.clip_(5.0, scale=True)
            if stage == "enc_loss_opt":
                runner.optimizer.step()
        capture(f)
    elif stage in ("attn", "optim", "video_fwd", "video_fwdbwd", "text_fwdbwd", "step_tiny"):
        import __graft_entry__ as ge
        runner, _, (videos, ids, mask) = ge.make_tiny_setup(torch.bfloat16)
        videos, ids, mask = videos.cuda(), ids.cuda(), mask.cuda()
        runner._graph_host = torch.zeros(10, dtype=torch.float32).pin_memory()
        ve, te = runner.video_encoder, runner.text_encoder
        if stage == "attn" or stage == "video_fwd":
            def f():
                with torch.no_grad(): return ve(videos)
            capture(f)
        elif stage == "video_fwdbwd":
            def f():
                for p in ve.parameters(): p.grad = None
                ve(videos).float().pow(2).sum().backward()
            capture(f)
        elif stage == "text_fwdbwd":
            def f():
                for p in te.parameters(): p.grad = None
                te(ids, mask).float().pow(2).sum().backward()
            capture(f)
        elif stage == "optim":
            ve(videos).float().pow(2).sum().backward(); te(ids, mask).float().pow(2).sum().backward()
            capture(lambda: runner.optimizer.step())
        else:
            capture(lambda: runner._train_step(videos=videos, input_ids=ids, attention_mask=mask))
    print("STAGE", stage, "OK", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for s in STAGES:
            r = subprocess.run([sys.executable, __file__, s], capture_output=True, text=True, timeout=280)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l][-3:]
            print(f"== {s}: rc={r.returncode}", " | ".join(t[:160] for t in tail), flush=True)
