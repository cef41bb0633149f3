"""Diagnostic: the fast walks of a library (RTX_LIB) over final-scene, saved; with a second argument, compared with a saved set.
A medium bounded by one sphere makes both boundary queries in one step; the step-by-step form must give the same hits bit for bit."""
import importlib, importlib.util, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
render = importlib.import_module("ray-tracer_amd.render"); scenes = importlib.import_module("ray-tracer_amd.scenes")
from oracle import binding as ob
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
out, ref = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
cfg, bvh = scenes.make("final-scene", seed=1, width=200, height=200, samples_per_pixel=8)
gs = render.Scene(bvh)
rays = np.concatenate([ob.OracleCamera(cfg).center_rays(0.5), m._secondary_rays(ob, cfg, bvh, 100_000, 3)])
res = {}
for mode in (render.TRAVERSAL_FAST, render.TRAVERSAL_RENDER):
    h = gs.trace(rays, traversal=mode, seed=77)
    res[f"id{mode}"] = h["prim_id"]; res[f"t{mode}"] = h["t"]
_, acc = render.Camera(cfg).render(gs, seed=5, want_accum=True)
res["acc"] = acc
np.savez(out, **res)
if ref:
    r = np.load(ref)
    for k in res:
        same = np.array_equal(res[k].view(np.uint32), r[k].view(np.uint32))
        print(k, "bit-identical" if same else f"DIFFERS in {int(np.count_nonzero(res[k].view(np.uint32) != r[k].view(np.uint32)))} of {res[k].size}")
