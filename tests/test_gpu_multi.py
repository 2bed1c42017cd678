"""Multi-GPU inside the library (pt_multi_*) rehearsed on one GPU: several contexts on device 0, tiles
cost-balanced over them, one sum at the end.  The image must be the single-context image bit for bit -- through the C ABI
and through pth_main (PATHTRACE_HIP_DEVICES) down to the bytes of the PPM -- and the RCCL form of the reduce runs with a
communicator of one device."""
import os

import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import scene_path
from test_gpu_main import _workdir

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("scene,n_ctx", [("cornell_box", 2), ("cornell_box_with_volume", 3), ("textured_room", 2)])
def test_contexts_on_one_gpu_reassemble_the_single_context_image(scene, n_ctx, monkeypatch):
    w, h, spp = 400, 225, 6
    sc = pt.Scene(scene_path(scene), w, h)
    single = pt.Renderer(sc, seed=2)
    ref = single.render(spp)
    ref_c = single.counters()
    single.close()
    for rr in (False, True):
        if rr:
            monkeypatch.setenv("PATHTRACE_HIP_MULTI", "roundrobin")
        m = pt.MultiRenderer(sc, [0] * n_ctx, seed=2, block=64)
        owners = m.tile_owners()
        assert len(owners) == -(-w // 64) * -(-h // 64) and set(owners) == set(range(n_ctx))
        m.render_async(0, 2)
        m.render_async(2, spp)                      # a render goes on after ...
        part, acc = m.snapshot()                    # ... a look at the live framebuffers
        assert acc <= w * h * spp and np.isfinite(part[part == part]).all()
        fb = m.framebuffer()
        assert np.array_equal(bits(fb), bits(ref)), (scene, n_ctx, rr)
        assert m.counters() == ref_c
        done, samples, rays = m.poll()
        assert done and samples == w * h * spp and rays == ref_c["rays"]
        fb2 = m.framebuffer()                       # reading twice does not add twice
        assert np.array_equal(bits(fb2), bits(ref))
        # the exchange moves only the pixels the peers (every context but the first) own, 16 bytes each
        tiles = pt.spiral_tiles(w, h, 64, 64)
        peer_px = sum((t[2] - t[0]) * (t[3] - t[1]) for t, o in zip(tiles, owners) if o != 0)
        assert m.exchange_bytes() == 16 * peer_px < 16 * w * h
        per_dev = [m.device_counters(i) for i in range(n_ctx)]
        assert all(sum(d[k] for d in per_dev) == v for k, v in ref_c.items()) and all(d["rays"] > 0 for d in per_dev)
        m.clear()
        m.render_async(0, 1)
        assert m.counters()["camera_samples"] == w * h
        m.close()
        monkeypatch.delenv("PATHTRACE_HIP_MULTI", raising=False)


def test_rccl_reduce_with_one_device(monkeypatch):
    # ncclCommInitAll + ncclReduce in a group, loaded from librccl.so on demand: the only form a one-GPU box can run
    w, h, spp = 256, 144, 4
    sc = pt.Scene(scene_path("cornell_box"), w, h)
    single = pt.Renderer(sc)
    ref = single.render(spp)
    single.close()
    monkeypatch.setenv("PATHTRACE_HIP_MULTI", "rccl")
    m = pt.MultiRenderer(sc, [0])
    m.render_async(0, spp)
    assert np.array_equal(bits(m.framebuffer()), bits(ref))
    m.close()
    with pytest.raises(pt.PathtraceError, match="distinct"):
        pt.MultiRenderer(sc, [0, 0])


def test_main_on_two_contexts_writes_the_same_ppm(tmp_path, monkeypatch):
    wd, cfg = _workdir(tmp_path, "cornell_box", film={"width": 320, "height": 180, "exposure": 0.0, "gamma": 2.2}, samples=8,
                       block_width=64, block_height=64)
    monkeypatch.setenv("PATHTRACE_HIP_DEVICES", "0")
    assert pt.lib().pth_main(os.fsencode(str(wd))) == 0, pt.last_error()
    one = (wd / "output" / "render.ppm").read_bytes()
    monkeypatch.setenv("PATHTRACE_HIP_DEVICES", "0,0")
    assert pt.lib().pth_main(os.fsencode(str(wd))) == 0, pt.last_error()
    two = (wd / "output" / "render.ppm").read_bytes()
    assert one == two and len(one) == len(b"P6\n320 180\n255\n") + 320 * 180 * 3
