"""BASELINE.json's full frame sizes on the GPU (configs 2-5: 1920x1080 and 3840x2160), at a sample count the checks
finish in seconds.  The oracle cannot render whole 4K frames quickly, so parity at these sizes is
  * bit-exact agreement with the oracle (stream mode) on randomly placed 16x16 windows of the full frame,
  * conservation laws of the path counters that hold for any size,
  * invariance of the image under the tile partition the 8-GPU configuration uses (NaiveSpiral 128x128 tiles dealt to 8
    owners, every owner's list rendered as multi-rect batches into one framebuffer = the single-GPU image, bit for bit).
"""
import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import scene_path
from pathtrace_amd.distributed import tiles_for_rank

pytestmark = pytest.mark.gpu

CASES = [("cornell_box", 1920, 1080), ("cornell_box_small_lights", 1920, 1080), ("cornell_box_with_volume", 1920, 1080),
         ("cornell_box", 3840, 2160)]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("scene,w,h", CASES)
def test_full_frame_windows_counters_and_tile_partition(oracle, scene, w, h):
    spp, L = 2, 4
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc)                       # default batch size: the frame is cut into bands
    whole = r.render(spp)
    c = r.counters()
    # conservation: one termination per camera sample, light_samples shadow rays per hit, rays = extension + shadow
    assert c["camera_samples"] == w * h * spp
    assert c["rays"] == c["extension_rays"] + c["shadow_rays"] and c["shadow_rays"] == L * c["extension_hits"]
    assert c["term_miss"] + c["term_rr"] + c["term_emitter"] + c["term_pdf"] + c["term_bounce_limit"] == c["camera_samples"]
    assert c["extension_rays"] - c["extension_hits"] == c["term_miss"]
    assert not np.isnan(whole).any() and (whole >= 0).all()   # de_nan (renderer.h:670); +inf is a legitimate estimate (tiny pdf)
    # windows against the oracle
    osc = oracle.Scene.from_json(scene_path(scene))
    cfg = oracle.make_config(w, h, spp)
    rng = np.random.default_rng(w + len(scene))
    wins = [(0, 0), (w - 16, h - 16), (w // 2 - 8, h // 2 - 8)] + [(int(rng.integers(0, w - 16)), int(rng.integers(0, h - 16))) for _ in range(21)]
    for (x0, y0) in wins:
        ofb = np.zeros((h, w, 3), np.float32)      # a buffer per window: windows may overlap
        osc.render_stream(cfg, seed=0, rect=(x0, y0, x0 + 16, y0 + 16), threads=4, fb=ofb)
        g, o = whole[y0:y0 + 16, x0:x0 + 16], ofb[y0:y0 + 16, x0:x0 + 16]
        assert ((bits(g) == bits(o)) | (g == o)).all(), (scene, w, h, x0, y0)
    # the 8-owner tile partition (round robin, and cost-balanced with a synthetic cost) reassembles the same image
    for costs in (None, [1 + (k * 7919) % 13 for k in range(len(pt.spiral_tiles(w, h, 128, 128)))]):
        r.clear()
        n_tiles = 0
        for rank in range(8):
            mine = tiles_for_rank(w, h, 128, 128, rank, 8, costs)
            n_tiles += len(mine)
            r.render_tiles_async(mine, 0, spp)
        assert n_tiles == -(-w // 128) * -(-h // 128)
        assert np.array_equal(bits(r.framebuffer()), bits(whole)) and r.counters() == c
    r.close()
