"""The launch plan lives in the library (include/pathtrace_hip.h, ABI v6): the rule that cuts a render call into wavefront
batches, host side (no GPU).  The figures are the ones bench.py's own `launch_plan` produced in round 4, which the
measurements of DESIGN.md 6 were taken with."""
import pathtrace_amd as pt


def test_abi_version_and_plan_symbols():
    L = pt.lib()
    assert L.pt_abi_version() == 6
    for name in ("pt_reserve", "pt_prime", "pt_get_plan", "pt_plan_batches", "pt_render_seconds", "pt_wait_for", "pt_multi_reserve",
                 "pt_multi_render_seconds", "pt_multi_wait_for"):
        assert hasattr(L, name), name


def test_the_rule_gives_round_4s_measured_plans():
    cap = pt.PLAN_MAX_PATHS
    assert cap == 1920 * 1080 * 96
    # N = 1: K = 64 (1024 spp) -> 12 batches of 86 spp; the driver's K = 20 (320 spp) -> 6 of 54
    assert pt.plan_batches(1920 * 1080, 1024, cap) == (86, 12)
    assert pt.plan_batches(1920 * 1080, 320, cap) == (54, 6)
    # a rank of N = 8 at K = 20 (269 k pixels): TWO batches, not one and not one per lane
    assert pt.plan_batches(269_000, 320, cap) == (160, 2)
    # ranks of N = 4 / N = 2 at K = 20 fit two batches as well
    assert pt.plan_batches(1920 * 1080 // 4, 320, cap) == (160, 2) and pt.plan_batches(1920 * 1080 // 2, 320, cap) == (160, 2)
    # config 5's frame on one GPU: the cap binds (24 spp of 8.3 M pixels), a multiple of the three lanes
    spp, nb = pt.plan_batches(3840 * 2160, 8192, cap)
    assert spp == 24 and nb % 3 == 0 and spp * 3840 * 2160 <= cap


def test_the_rule_is_total_and_never_exceeds_the_slots():
    for pixels in (1, 17, 4096, 200 * 200, 1920 * 1080, 3840 * 2160):
        for samples in (1, 2, 3, 4, 16, 100, 1024, 8192):
            for slots in (64, 4096, 1 << 23, pt.PLAN_MAX_PATHS):
                for lanes in (1, 2, 3, 4):
                    spp, nb = pt.plan_batches(pixels, samples, slots, lanes)
                    assert spp >= 1 and nb >= 1
                    assert (nb - 1) * spp < samples <= nb * spp            # the batches cover the samples, none is empty
                    assert spp * pixels <= max(slots, pixels)                 # a batch fits the slots (one sample always may)
                    if samples >= 2 and slots >= pixels:
                        assert nb >= 2                                         # never one batch: its tail would overlap nothing
