import sys, os, subprocess, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
def load(n): return json.load(open(os.path.join(ROOT, "scenes", n + ".json")))
def variants():
    v = {}
    s = load("cornell_box"); s["world"]["color"] = [0.1, 0.1, 0.1]; v["cb_bg"] = s
    s = load("cornell_box_with_volume"); s["world"]["color"] = [0, 0, 0]; v["vol_blackbg"] = s
    s = load("cornell_box_with_volume"); s["instances"][6]["skip"] = True; v["vol_skipped_fog"] = s   # NV=0 but prims present
    s = load("cornell_box_with_volume"); s["instances"][6]["transform"]["translate"] = [212.5, 5000, 147.5]; v["vol_far_away"] = s  # NV=1, never hit
    return v
if len(sys.argv) > 1:
    import numpy as np
    import pathtrace_amd as pt
    from oracle import pt_oracle as po, scene_params as sp
    name, mb, ls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    s = variants()[name]
    w = h = 64; spp = 4
    r = pt.Renderer(pt.Scene(text=json.dumps(s), width=w, height=h), max_bounces=mb, light_samples=ls)
    fb = r.render(spp)
    ref, oc = po.Scene(sp.load_scene_params(s)).render_stream(po.make_config(w, h, spp, max_bounces=mb, light_samples=ls), seed=0)
    print(name, mb, ls, "equal:", np.array_equal(fb, ref), "ndiff", int((fb != ref).sum()), r.counters()["rays"], oc["rays"])
else:
    for name in variants():
        p = subprocess.run([sys.executable, __file__, name, "1", "0"], capture_output=True, text=True)
        err = [l for l in p.stderr.splitlines() if "HSA_STATUS" in l or "Error" in l]
        print(name, "rc", p.returncode, p.stdout.strip()[-200:], err[:1], flush=True)
