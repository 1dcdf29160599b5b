"""pt_multi_*: several GPUs behind one handle of the C ABI. This machine has one GPU, so what is checked here:
one listed device equals the single-GPU entry points bit for bit; RCCL can be opened and has the entry points the
library uses; and, with the host-side sum that allows one GPU to be listed twice or three times, the two partitions:
tiles assemble to the single-GPU frame bit for bit, virtual frames are the mean of the per-GPU frames as the CPU
restatement defines it; and the RCCL leg with a communicator of one rank that sends its image to itself. A transfer
between two different GPUs runs only on a machine that has them."""
import numpy as np
import pytest

from path_tracing_amd import abi, scenes
from path_tracing_amd.dist import virtual_frame_time
from tests.oracle_lib import oracle_render

pytestmark = pytest.mark.gpu


def test_one_device_is_the_single_gpu_path():
    scene = scenes.boxes(160, 90, 8, 10)
    frame = scene.frame(time=1.5)
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        want = ctx.render(frame)
    for mode in (abi.PT_MULTI_TILES, abi.PT_MULTI_VIRTUAL_FRAMES):
        with abi.MultiContext([0], mode=mode) as m:
            m.upload_scene(scene)
            got = m.render(frame)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            # frame flags pass through on one device
            got16 = m.render(scene.frame(time=1.5, flags=abi.PT_FRAME_OUTPUT_F16))
            assert got16.dtype == np.float16 and np.array_equal(got16.view(np.uint16), want.astype(np.float16).view(np.uint16))


def test_rccl_can_be_opened():
    assert abi.rccl_version() >= 20000   # RCCL reports an NCCL 2.x version number


@pytest.mark.parametrize("mode", [abi.PT_MULTI_TILES, abi.PT_MULTI_VIRTUAL_FRAMES])
def test_one_rank_through_the_collective(mode):
    """The RCCL leg itself on the one GPU there is: a communicator of one rank, the grouped ncclSend / ncclRecv (to itself)
    behind the frame on its stream, the summing kernel and the copy. The image must be pt_render's, twice in a row."""
    scene = scenes.cornell_box(100, 61, 4, 10)
    frame = scene.frame(time=0.5)
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        want = ctx.render(frame)
    with abi.MultiContext([0], mode=mode, flags=abi.PT_MULTI_FLAG_FORCE_COLLECTIVE) as m:
        m.upload_scene(scene)
        for _ in range(2):
            got = m.render(frame)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("n", [2, 3])
def test_tiles_over_several_handles_assemble_the_frame(n):
    scene = scenes.cornell_box(100, 61, 4, 10)     # ragged size: partial tiles
    frame = scene.frame()
    ref = oracle_render(scene, frame)
    with abi.MultiContext([0] * n, mode=abi.PT_MULTI_TILES, flags=abi.PT_MULTI_FLAG_HOST_SUM) as m:
        m.upload_scene(scene)
        got = m.render(frame)
        rays = [m.rays(k) for k in range(n)]
    assert np.array_equal(got.view(np.uint32), ref["image"].view(np.uint32))
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        ctx.render(frame)
        assert sum(rays) == ctx.stats()["rays"] and all(r > 0 for r in rays)


def test_virtual_frames_are_the_mean_of_the_per_gpu_frames():
    scene = scenes.boxes(96, 54, 4, 10)
    n = 2
    with abi.MultiContext([0] * n, mode=abi.PT_MULTI_VIRTUAL_FRAMES, flags=abi.PT_MULTI_FLAG_HOST_SUM) as m:
        m.upload_scene(scene)
        got = m.render(scene.frame(time=0.0))
    f = [oracle_render(scene, scene.frame(time=virtual_frame_time(k)))["image"] for k in range(n)]
    want = (f[0] + f[1]) * np.float32(1.0 / n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.all(got[..., 3] == 1.0) and not np.array_equal(f[0], f[1])


def test_multi_errors():
    scene = scenes.cornell_box(32, 32, 1, 2)
    with pytest.raises(abi.PtError) as e:
        abi.MultiContext([0, 0])          # the same GPU twice needs the host-side sum
    assert e.value.code == abi.PT_ERR_INVALID_ARGUMENT
    with pytest.raises(abi.PtError) as e:
        abi.MultiContext([0, 99], flags=abi.PT_MULTI_FLAG_HOST_SUM)
    assert e.value.code == abi.PT_ERR_NO_DEVICE
    with abi.MultiContext([0, 0], flags=abi.PT_MULTI_FLAG_HOST_SUM) as m:
        with pytest.raises(abi.PtError) as e:
            m.render(scene.frame())       # nothing uploaded
        assert e.value.code == abi.PT_ERR_NOT_READY
        m.upload_scene(scene)
        with pytest.raises(abi.PtError) as e:
            m.render(scene.frame(flags=abi.PT_FRAME_ACCUMULATE))
        assert e.value.code == abi.PT_ERR_UNSUPPORTED
        with pytest.raises(abi.PtError) as e:
            m.render(scene.frame(part_index=0, part_count=2))
        assert e.value.code == abi.PT_ERR_INVALID_ARGUMENT
        assert np.isfinite(m.render(scene.frame())).all()
