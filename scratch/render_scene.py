import importlib, sys, time
sys.path.insert(0, '.')
from PIL import Image
render = importlib.import_module("ray-tracer_amd.render")
scenes = importlib.import_module("ray-tracer_amd.scenes")
name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg, bvh = scenes.make(name, seed=1, width=w, height=h, samples_per_pixel=spp)
scene, cam = render.Scene(bvh), render.Camera(cfg)
t = time.time()
rgb, st = cam.render(scene, seed=1, want_stats=True)
print(name, "kernel ms", st.kernel_ms, "Msamples/s", st.samples / st.kernel_ms / 1e3, "rays/sample", st.rays / st.samples)
Image.fromarray(rgb).save(f"gpurun_out/{name}.png")
