"""The per-scene build of the traversal sweep (pt_spec.cpp), host side: the scene's program as a compile-time table and the
hiprtc build of pt_kernels.hip for it -- hiprtc cross-compiles gfx950 without a GPU, so that part is checked here."""
import re

import pytest

import pathtrace_amd as pt
from conftest import ALL_SCENES, scene_path


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_box_with_volume", "textured_room"])
def test_the_module_of_a_scene_compiles_for_gfx950(scene):
    sc = pt.Scene(scene_path(scene), 320, 180)
    assert pt.spec_build_check(sc, 4) > 20000       # a code object with k_extend, k_connect and three k_trace
    assert pt.spec_build_check(sc, 3) > 20000       # odd light_samples: one shadow ray per sweep


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_the_table_is_the_fast_program(scene):
    sc = pt.Scene(scene_path(scene), 320, 180)
    if scene == "cornell_box_nested_fog":   # a medium whose boundary is a medium: the general sweep carries it, nothing to specialise
        with pytest.raises(pt.PathtraceError, match="does not take the fast sweep"):
            pt.spec_header(sc)
        return
    text = pt.spec_header(sc)
    n = int(re.search(r"#define PT_SPEC_N (\d+)", text).group(1))
    rows = re.findall(r"^  \{([-0-9, ]+)\},$", text, re.M)
    assert len(rows) == n and all(len(r.split(",")) == 32 for r in rows)
    kinds = [int(r.split(",")[0]) for r in rows]
    assert kinds[0] == 0 and 1 not in kinds         # the root's ENTER first, no COMBINE ops in the fast program
    d = sc.desc
    leaf_insts = [int(r.split(",")[1]) for r, k in zip(rows, kinds) if k >= 2]
    # every instance is a leaf op (twice where bvh_node's n == 1 case made it both children, bvh.h:133-175)
    assert set(leaf_insts) == set(range(d.n_instances)) and len(leaf_insts) <= d.n_instances + d.n_nodes
    ga = int(re.search(r"#define PT_SPEC_GA (\d)", text).group(1))
    assert ga == (1 if any(k >= 6 for k in kinds) else 0)


def test_a_broken_build_reports_instead_of_raising_into_the_render(monkeypatch):
    # PATHTRACE_HIP_SPEC_BREAK makes the build fail on purpose: the check says so; a context would keep the generic kernels
    # (tests/test_gpu_spec.py renders through that fallback on the GPU)
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_BREAK", "1")
    sc = pt.Scene(scene_path("three_orbs"), 64, 64)
    with pytest.raises(pt.PathtraceError, match="fails on purpose"):
        pt.spec_build_check(sc, 4)


def test_the_module_is_built_by_this_toolchain_whatever_the_process_holds(tmp_path, monkeypatch):
    # A PyTorch wheel brings its own libhiprtc / libamd_comgr (an older LLVM) under the sonames of /opt/rocm's: with torch
    # imported, an in-process dlopen("libhiprtc.so") compiled the module with THAT compiler (24 spilled VGPRs in the generic
    # k_connect against none).  The build therefore runs in a helper process (pathtrace_amd/lib/pt_spec_cc): the code object's
    # producer string must be hipcc's.
    import os
    import subprocess
    import torch  # noqa: F401  (the point: its libraries are in the process)
    from pathtrace_amd import build as ptb

    assert os.access(ptb.SPEC_CC, os.X_OK)
    out = tmp_path / "m.co"
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_DUMP", str(out))
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_FLAGS", "-DPT_TEST_TOOLCHAIN=1")   # a key nothing has cached: this build runs now
    sc = pt.Scene(scene_path("light_test"), 48, 27)
    assert pt.spec_build_check(sc, 7) > 20000
    blob = out.read_bytes()
    hipcc = subprocess.run([ptb.HIPCC, "--version"], capture_output=True, text=True).stdout
    version = re.search(r"clang version (\S+)", hipcc).group(1)
    build_id = re.search(r"roc-[0-9.]+ \d+ [0-9a-f]+", hipcc)
    assert ("clang version " + version).encode() in blob
    assert build_id is None or build_id.group(0).encode() in blob


@pytest.mark.parametrize("helper", ["/nonexistent/pt_spec_cc", "/bin/false"])
def test_without_a_working_helper_the_build_runs_in_process(helper, monkeypatch):
    # the helper process is the preferred compiler, not a requirement: a missing file or one that ends without a result
    # leaves the in-process hiprtc (and after that, on the GPU, the generic kernels)
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_CC", helper)
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_FLAGS", "-DPT_TEST_NO_HELPER_%d=1" % len(helper))   # a key nothing has cached
    sc = pt.Scene(scene_path("three_orbs"), 40, 30)
    assert pt.spec_build_check(sc, 6) > 20000


def test_provenance_names_the_compiler_and_survives_a_hostile_environment(monkeypatch):
    # pt_spec_info / pt_spec_build_info: which process compiled, which libhiprtc file, the code object's producer string and
    # whether it is the compiler the library was built with.  The helper is started with an environment of the library's own
    # making: an LD_LIBRARY_PATH that puts PyTorch's bundled ROCm first (hiprtc opens libamd_comgr by soname), an LD_PRELOAD
    # or a profiler's tool library in the host's environment must not reach it.
    import os
    import torch

    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    monkeypatch.setenv("LD_LIBRARY_PATH", torch_lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    monkeypatch.setenv("LD_PRELOAD", "/nonexistent/libtool.so")
    monkeypatch.setenv("HSA_TOOLS_LIB", "/nonexistent/librocprofiler-sdk-tool.so")
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_FLAGS", "-DPT_TEST_PROVENANCE=1")   # a key nothing has cached
    sc = pt.Scene(scene_path("cornell_box"), 32, 32)
    info = pt.spec_build_info(sc, 4)
    assert info["built_by"] == "helper" and info["own_compiler"] is True, info
    assert "/opt/rocm" in info["rtc_lib"] and "torch" not in info["rtc_lib"], info
    assert info["producer"].startswith("AMD clang version") and info["library_producer"] in info["producer"]
    # without the helper the in-process compiler is whatever the process resolves: the record says so
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_CC", "/nonexistent/pt_spec_cc")
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_FLAGS", "-DPT_TEST_PROVENANCE=2")
    info = pt.spec_build_info(sc, 4)
    assert info["built_by"] == "in-process" and "is not there" in info.get("note", ""), info
    assert info["own_compiler"] in (True, False) and info["producer"]


def _table_survives(code_object):
    """Does the module still READ the table at run time?  A reference to it never shows by name in a disassembly (the address is
    s_getpc_b64 plus a resolved immediate, .rodata is not disassembled), so look at what changes when the table survives: an
    object symbol for it in the symbol table, and any s_getpc_b64 at all -- the per-scene kernels call nothing and address no
    constant data, a fully folded module has neither."""
    import os
    import subprocess
    from pathtrace_amd import build as ptb

    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(ptb.HIPCC))), "lib", "llvm", "bin")
    syms = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-sW", str(code_object)], capture_output=True, text=True).stdout
    text = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", str(code_object)], capture_output=True, text=True).stdout
    assert "k_connect" in text and "k_extend" in text and "FUNC" in syms
    table_symbol = any("kSpecW" in line and "OBJECT" in line for line in syms.splitlines())
    return table_symbol or "s_getpc_b64" in text


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_box_with_volume", "three_orbs", "textured_room"])
def test_the_table_folds_into_the_code(scene, tmp_path, monkeypatch):
    # The point of the per-scene build: the op loop unrolls completely and the table becomes literals.  LLVM sizes the unrolled
    # loop BEFORE it folds the per-kind dispatch and, past a threshold, quietly keeps a run-time loop over a table in constant
    # memory that carries every leaf body (round 4: the volume scene's k_connect, 18.7 against 16.2 ms).
    out = tmp_path / "m.co"
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_DUMP", str(out))
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_FLAGS", "-DPT_TEST_FOLD=1")   # a key nothing has cached: this build runs now
    sc = pt.Scene(scene_path(scene), 64, 36)
    assert pt.spec_build_check(sc, 4) > 20000
    assert not _table_survives(out)
    # ... and the register budget of the module's k_connect (round 5, pt_spec.cpp PT_CONNECT_NOHOIST): nothing spilled, six waves per
    # SIMD (<= 80 VGPRs) for rects and boxes, five (<= 96) with sphere / medium leaves -- one more resident wave each than round 4
    meta = _kernel_metadata(out, "k_connect")
    assert meta["vgpr_spill_count"] == 0 and meta["private_segment_fixed_size"] == 0, meta
    assert meta["vgpr_count"] <= (80 if scene == "cornell_box" else 96), meta


def _kernel_metadata(code_object, name):
    """The integer fields of one kernel's entry in the code object's metadata note (llvm-readelf --notes)."""
    import os
    import subprocess
    from pathtrace_amd import build as ptb

    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(ptb.HIPCC))), "lib", "llvm", "bin")
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(code_object)], capture_output=True, text=True).stdout
    entries = notes.split("  - .agpr_count:")[1:]          # one entry per kernel, fields in alphabetical order
    mine = [e for e in entries if re.search(r"\.name:\s+\S*" + name, e)]
    assert len(mine) == 1, (name, len(entries))
    return {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", mine[0], re.M)}


def test_the_fold_check_trips_when_llvm_keeps_the_loop(tmp_path, monkeypatch):
    # negative control: the same module built with LLVM's own default threshold restored to a low value keeps a
    # run-time loop over the table -- the check above must see it
    out = tmp_path / "m.co"
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_DUMP", str(out))
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_FLAGS", "-DPT_TEST_FOLD=2 -mllvm -pragma-unroll-threshold=2000")
    sc = pt.Scene(scene_path("cornell_box_with_volume"), 64, 36)
    assert pt.spec_build_check(sc, 4) > 20000
    assert _table_survives(out)


def _leaf_slots(text):
    rows = re.findall(r"^  \{([-0-9, ]+)\},$", text, re.M)
    return [(int(r.split(",")[0]), int(r.split(",")[1]), int(r.split(",")[2])) for r in rows]   # (kind, instance, slot)


def test_the_rects_that_wall_a_scene_in_are_marked():
    # DOp::slot bit 5 (pt_context.cpp rect_walls_scene): the per-scene k_connect proves these unreachable for a shadow ray instead of
    # testing them.  The Cornell boxes: floor, ceiling (one unit above the light), back and side walls -- never the light
    import json
    from scene_gen import room_scene

    # -- and of a box its sides (bits 8-13, box::hit's order): the bottoms (XZ at p0.y, bit 12) of the blocks that stand on the floor
    for scene, walls, boxes in (("cornell_box", 5, 2), ("cornell_box_with_volume", 5, 1), ("cornell_box_small_lights", 6, 2)):
        text = pt.spec_header(pt.Scene(scene_path(scene), 64, 36))
        assert int(re.search(r"#define PT_SPEC_NWALL (\d+)", text).group(1)) == walls + boxes
        assert sum(1 for k, _, s in _leaf_slots(text) if 2 <= k <= 4 and (s & 48) == 32) == walls
        assert [s >> 8 for k, _, s in _leaf_slots(text) if k == 5] == [16] * boxes
    # generated rooms: a partition inside the room and the lights are never marked; a ceiling whose light hangs closer than 2^-10 of
    # the room under it is not either (the kernel's per-ray proof would fail for the far corners), every other wall is
    marked = unmarked_ceilings = spheres = 0
    for seed in range(24):
        js = room_scene(seed)
        text = pt.spec_header(pt.Scene(text=json.dumps(js), width=48, height=36))
        inst = js["instances"]
        H = max(i["transform"]["translate"][1] for i in inst if i["type"] == "ref" and i["primitive"]["id"] == "floor")
        gap = min(H - i["transform"]["translate"][1] for i in inst if i["type"] == "direct" and i["primitive"].get("material", {}).get("id") == "light"
                  and i["primitive"]["type"] == "rect")
        if any(i["type"] == "direct" and i["primitive"]["type"] == "sphere" for i in inst):   # a light that is not sampled by point: nothing marked
            assert not any((s & 48) == 32 or (k == 5 and s >> 8) for k, _, s in _leaf_slots(text)), seed
            spheres += 1
            continue
        for kind, ii, slot in _leaf_slots(text):
            if not 2 <= kind <= 4:
                continue
            e, wall = inst[ii], (slot & 48) == 32
            mat = e["primitive"].get("material", {}).get("id") if e["type"] == "direct" else None
            is_partition = e["type"] == "direct" and mat == "white" and e["primitive"].get("align") == "yz"
            is_ceiling = e["type"] == "ref" and e["primitive"]["id"] == "floor" and e["transform"]["translate"][1] > 0
            if mat == "light" or is_partition:
                assert not wall, (seed, ii)
            elif is_ceiling and gap < 0.25:
                assert not wall, (seed, ii, gap)
                unmarked_ceilings += 1
            elif not is_ceiling or gap >= 1.0:
                assert wall, (seed, ii, gap)
            marked += wall
    assert marked > 80 and unmarked_ceilings > 0 and spheres > 0
