import importlib, numpy as np, sys
sys.path.insert(0, '.')
render = importlib.import_module("ray-tracer_amd.render")
scenes = importlib.import_module("ray-tracer_amd.scenes")
from oracle import binding as ob
cfg, bvh = scenes.make("bouncing-spheres", seed=1, width=480, height=270)
rays = ob.OracleCamera(cfg).center_rays(0.5)
want = ob.OracleMesh(bvh).trace(rays)
got = render.Scene(bvh).trace(rays, traversal=1)
hit = want["prim_id"] != 0xFFFFFFFF
d = np.abs(got["uv"][hit] - want["uv"][hit])
print("max abs diff uv", d.max(0))
i = np.argmax(d.max(1))
print(got[hit][i], want[hit][i])
n = want["n"][hit][i]
print("normal", n, "acos(-ny)", np.arccos(np.float32(-n[1])), "atan2", np.arctan2(np.float32(-n[2]), np.float32(n[0])))
print("count > 1e-6:", (d > 1e-6).sum(), "of", d.size, " >3e-7:", (d>3e-7).sum())
