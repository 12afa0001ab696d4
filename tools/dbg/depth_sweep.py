import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
W, H = 3840, 2160
sc = S.Mandelbulb()
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
for depth in (1, 2, 3, 4, 6, 8):
    ctx = native.Context(0); ctx.set_stream(st.cuda_stream); ctx.set_samples_in_flight(depth)
    h = ctx.create_scene(sc)
    line = f"depth {depth}:"
    for N in (1, 2, 4, 8):
        fb = ctx.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, N, N // 2)
        h2, h3 = J.halton(2), J.halton(3)
        def step():
            u.randNoise[0], u.randNoise[1] = next(h2), next(h3)
            ctx.render_sample(h, fb, u, None, abi.RM_RENDER_FAST)
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter(); K = 60
        for _ in range(K): step()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / K * 1e3
        line += f"  N={N} {t:.3f} ms"
        fb.destroy()
    print(line)
    h.destroy(); ctx.close()
