"""Run by tests/test_gpu_parity.py in a fresh process: render into torch-owned device memory through
pt_render_device on torch's stream, single rank, both partition modes, and compare with the CPU restatement."""
import sys
from pathlib import Path

import numpy as np
import torch

assert torch.cuda.is_available(), "no GPU visible to torch"
torch.cuda.init()

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from path_tracing_amd import abi, scenes  # noqa: E402
from path_tracing_amd import dist as ptdist  # noqa: E402
from tests.oracle_lib import oracle_render  # noqa: E402

scene = scenes.cornell_box(64, 40, 2, 6)
dev = torch.device("cuda", 0)
with abi.Context() as ctx:
    ctx.upload_scene(scene)

    def render_fn(frame):
        out = torch.zeros((frame.height, frame.width, 4), dtype=torch.float32, device=dev)
        ctx.render_device(frame, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        return out

    full = ptdist.DistributedRenderer(render_fn, 0, 1, "tiles").render(lambda **kw: scene.frame(**kw)).cpu().numpy()
    parts = [render_fn(scene.frame(part_index=k, part_count=2)) for k in range(2)]  # two ranks' work on one device
    both = (parts[0] + parts[1]).cpu().numpy()
    vf = ptdist.DistributedRenderer(render_fn, 0, 1, "virtual_frames").render(lambda **kw: scene.frame(**kw)).cpu().numpy()
    half = torch.zeros((scene.height, scene.width, 4), dtype=torch.float16, device=dev)
    ctx.render_device(scene.frame(flags=abi.PT_FRAME_OUTPUT_F16), half.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    half = half.cpu().numpy()
ref = oracle_render(scene, scene.frame())["image"]
assert np.array_equal(full.view(np.uint32), ref.view(np.uint32)), "tiles mode differs"
assert np.array_equal(both.view(np.uint32), ref.view(np.uint32)), "sum of partitions differs"
assert np.array_equal(vf.view(np.uint32), ref.view(np.uint32)), "virtual frame of rank 0 differs"
assert np.array_equal(half.view(np.uint16), ref.astype(np.float16).view(np.uint16)), "binary16 output differs"

# Frames in flight: three contexts on three streams take frames in turn, as a caller that overlaps frames does. From
# their second frames on the schedule comes from hints, k_paths keeps only the workgroups the frame has work for (most
# of a 1080p launch leaves at once here) and sees the other frames running. Every image must be the one a context
# renders alone, and afterwards a single context must again find the device to itself.
big = scenes.scene_glb_textured(1920, 1080, 16, 10)
times = [0.5, 1.5, 2.5]
want = []
with abi.Context() as ctx:
    ctx.upload_scene(big)
    for t in times:
        o = torch.zeros((1080, 1920, 4), dtype=torch.float32, device=dev)
        ctx.render_device(big.frame(time=t), o.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        want.append(o.cpu().numpy())
import os
os.environ["PT_AMD_DEBUG"] = "1"     # (read when a context is created: these contexts print a line per frame on stderr)
ctxs = [abi.Context() for _ in range(3)]
del os.environ["PT_AMD_DEBUG"]
streams = [torch.cuda.Stream(dev) for _ in range(3)]
outs = [[torch.zeros((1080, 1920, 4), dtype=torch.float32, device=dev) for _ in times] for _ in range(3)]
for c in ctxs:
    c.upload_scene(big)
for rep in range(3):
    for j, t in enumerate(times):
        for k in range(3):
            ctxs[k].render_device(big.frame(time=t), outs[k][j].data_ptr(), streams[k].cuda_stream)
    torch.cuda.synchronize(dev)
    for k in range(3):
        for j in range(len(times)):
            assert np.array_equal(outs[k][j].cpu().numpy().view(np.uint32), want[j].view(np.uint32)), f"in flight: context {k}, frame {j}, pass {rep}"
ctxs[0].render_device(big.frame(time=0.5), outs[0][0].data_ptr(), streams[0].cuda_stream)   # (lets the 20 ms of the last overlap pass)
torch.cuda.synchronize(dev)
import time
time.sleep(0.05)
r, w = os.pipe()
saved = os.dup(2)
sys.stderr.flush()
os.dup2(w, 2)
try:
    ctxs[0].render_device(big.frame(time=0.5), outs[0][0].data_ptr(), streams[0].cuda_stream)
    torch.cuda.synchronize(dev)
    ctxs[0].stats()      # (takes the frame's record in, which prints what the frame saw at its start)
finally:
    os.dup2(saved, 2)
    os.close(w)
log = os.read(r, 1 << 16).decode()
assert "other frames running at its start 0\n" in log, log[-600:]
assert np.array_equal(outs[0][0].cpu().numpy().view(np.uint32), want[0].view(np.uint32))
for c in ctxs:
    c.close()
print("ok")
