import sys, json, copy, numpy as np, os, tempfile
sys.path.insert(0, '.')
import pathtrace_amd as pt
from oracle import pt_oracle as po
base = json.load(open('scenes/textured_room.json'))
def variant(name, f):
    s = copy.deepcopy(base); f(s); return name, s
def set_mat(s, mid, data, typ=None):
    for m in s['materials']:
        if m['id'] == mid:
            m['data'] = data
            if typ: m['type'] = typ
V = [
 variant('full', lambda s: None),
 variant('const_bg', lambda s: s.__setitem__('world', {'color': [0.3, 0.4, 0.5]})),
 variant('const_lamp', lambda s: set_mat(s, 'lamp', {'color': [3, 3, 3]})),
 variant('const_marble', lambda s: set_mat(s, 'marble', {'color': [0.5, 0.5, 0.5]})),
 variant('const_tiles', lambda s: set_mat(s, 'tiles', {'color': [0.5, 0.5, 0.5]})),
 variant('const_nested', lambda s: set_mat(s, 'nested', {'color': [0.5, 0.5, 0.5]})),
 variant('two_sided_light', lambda s: set_mat(s, 'light', {'color': [9, 9, 9]})),
]
def allconst(s):
    s['world'] = {'color': [0.3, 0.4, 0.5]}
    for mid in ('lamp',): set_mat(s, mid, {'color': [3, 3, 3]})
    for mid in ('marble', 'tiles', 'nested'): set_mat(s, mid, {'color': [0.5, 0.5, 0.5]})
V.append(variant('all_const', allconst))
def allconst2(s):
    allconst(s); set_mat(s, 'light', {'color': [9, 9, 9]})
V.append(variant('all_const_two_sided', allconst2))
d = tempfile.mkdtemp()
for name, s in V:
    p = os.path.join(d, name + '.json'); json.dump(s, open(p, 'w'))
    for mb in (1, 10):
        sc = pt.Scene(p, 48, 48); r = pt.Renderer(sc, max_bounces=mb); g = r.render(2); gc = r.counters(); r.close()
        osc = po.Scene.from_json(p); o, oc = osc.render_stream(po.make_config(48, 48, 2, max_bounces=mb), seed=0, threads=4)
        bad = (g.view(np.uint32) != o.view(np.uint32)) & (g != o)
        print(name, 'mb', mb, 'mismatch', int(bad.sum()), 'of', bad.size, 'rays', gc['rays'], oc['rays'], 'mean', g.mean(), o.mean(), flush=True)
