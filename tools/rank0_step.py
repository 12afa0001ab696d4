"""What rank 0 of an N-GPU run does per step, without the collective: render its stripes (3 samples in flight), snapshot the
colour plane, assemble a gathered frame.  Shows whether the host loop or the extra kernels limit the step at N = 8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_cases as GC
from raymarching_engine_amd import abi, job as J, native, scene as S, shard
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if os.environ.get("OWN_STREAM", "1") == "1":  # not the legacy default stream: it synchronises with too much
    torch.cuda.set_stream(torch.cuda.Stream())
print("torch stream", torch.cuda.current_stream())
ctx = native.Context(0); ctx.set_samples_in_flight(3)
if os.environ.get("CTX_STREAM", "torch") == "torch":
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
sc = S.Mandelbulb(); h = ctx.create_scene(sc)
W, H = 3840, 2160
schema = J.make_schema(sc, W, H, counts=(256,), render_mode="full", position=(0, 0, -2.5), lights=GC.LIGHT)
u = J.uniforms_from_schema(schema, (0.5, 1 / 3))
counts = shard.row_counts(H, N); max_rows = max(counts)
planes = [torch.zeros((max_rows, W, 4), dtype=torch.float32, device=dev) for _ in range(3)]
fb = ctx.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, N, 0, *(p.data_ptr() for p in planes)) if os.environ.get("PLANES", "torch") == "torch" else ctx.create_striped_framebuffer(W, H, shard.STRIPE_ROWS, N, 0)
if os.environ.get("NO_BIG") != "1":
    pass
recv = torch.zeros((N, max_rows, W, 4), dtype=torch.float32, device=dev); frame = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
h2, h3 = J.halton(2), J.halton(3)
aux = torch.cuda.Stream()
def step(extra):
    u.randNoise[0], u.randNoise[1] = next(h2), next(h3)
    ctx.render_sample(h, fb, u, None, 1)
    if extra:
        snap = planes[0].clone()
        torch.cuda.current_stream().wait_stream(aux)
        recv[0].copy_(snap)  # stands in for the collective's local copy
        aux.wait_stream(torch.cuda.current_stream())
        ctx.assemble_striped(recv.data_ptr(), N, max_rows, W, H, shard.STRIPE_ROWS, frame.data_ptr(), aux.cuda_stream)
for extra in (False, False, False, True, True):
    for _ in range(6): step(extra)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 60
    for _ in range(K): step(extra)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"N={N} rank 0, {'render + snapshot + assemble' if extra else 'render only'}: {t/K*1e3:.3f} ms per step (host loop alone {t_host/K*1e3:.3f} ms)")
