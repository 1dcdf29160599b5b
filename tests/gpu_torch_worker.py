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
print("ok")
