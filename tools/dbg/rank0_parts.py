import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = native.Context(0); ctx.set_stream(st.cuda_stream)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
W, H, N = 3840, 2160, 8
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
aux = torch.cuda.Stream(); issue = torch.cuda.Stream()
counts = shard.row_counts(H, N); max_rows = max(counts)
ctx.set_samples_in_flight(3)
recv = torch.zeros((N, max_rows, W, 4), dtype=torch.uint8, device=dev)
frame = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
snaps = [torch.zeros((max_rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
planes = [torch.zeros((max_rows, W, 4), dtype=torch.float32, device=dev) for _ in range(3)]
fb = ctx.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, N, 0, *(p.data_ptr() for p in planes))
h2, h3 = J.halton(2), J.halton(3)
for mode in ("render", "render+present", "+copy(issue)", "+assemble(aux) no copy", "+copy+assemble", "+copy+assemble all on st"):
    k = [0]
    def step():
        u.randNoise[0], u.randNoise[1] = next(h2), next(h3)
        ctx.render_sample(h, fb, u, None, abi.RM_RENDER_FAST)
        if mode == "render": return
        snap = snaps[k[0] & 1]; k[0] += 1
        ctx.present_rows(fb, k[0], snap.data_ptr(), st.cuda_stream)
        if mode == "render+present": return
        if mode == "+copy+assemble all on st":
            recv[0].copy_(snap, non_blocking=True)
            ctx.assemble_striped_bytes(recv.data_ptr(), N, max_rows, W * 4, H, shard.STRIPE_ROWS, frame.data_ptr(), st.cuda_stream)
            return
        issue.wait_stream(st); issue.wait_stream(aux)
        if "copy" in mode:
            with torch.cuda.stream(issue):
                recv[0].copy_(snap, non_blocking=True)
        aux.wait_stream(issue)
        if "assemble" in mode:
            ctx.assemble_striped_bytes(recv.data_ptr(), N, max_rows, W * 4, H, shard.STRIPE_ROWS, frame.data_ptr(), aux.cuda_stream)
    for _ in range(8): step()
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 80
    for _ in range(K): step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{mode:32s} {t / K * 1e3:.3f} ms per step (host loop {th / K * 1e3:.3f})")
