"""C1 (the `default` scene, 400x225, 16 spp, depth 50) through rtx_render with host buffers: per-call latency."""
import importlib, sys, time
sys.path.insert(0, '.')
render = importlib.import_module("ray-tracer_amd.render")
scenes = importlib.import_module("ray-tracer_amd.scenes")
cfg, bvh = scenes.make("default", seed=0, width=400, height=225, samples_per_pixel=16)
t0 = time.perf_counter(); scene = render.Scene(bvh); t1 = time.perf_counter()
cam = render.Camera(cfg)
ts = []
for i in range(10):
    t = time.perf_counter(); rgb = cam.render(scene, seed=i); ts.append((time.perf_counter() - t) * 1e3)
rgb, st = cam.render(scene, seed=99, want_stats=True)
print("scene create %.1f ms; rtx_render calls (ms):" % ((t1 - t0) * 1e3), [round(x, 2) for x in ts], "| with stats: kernel ms", round(st.kernel_ms, 3), "total_ms", round(st.total_ms, 3))
print("Msamples/s per call (median): %.0f" % (400 * 225 * 16 / (sorted(ts)[len(ts) // 2] * 1e-3) / 1e6))
