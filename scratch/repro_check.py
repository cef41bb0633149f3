"""Are two runs of the same frame bit-identical?  usage: [RTX_ITEM_SLOTS=1] python scratch/repro_check.py [scene] [spp]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
render = importlib.import_module("ray-tracer_amd.render")
scenes = importlib.import_module("ray-tracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "bouncing-spheres"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 640
cfg, bvh = scenes.make(name, seed=1, width=256, height=256, samples_per_pixel=spp)
scene, cam = render.Scene(bvh), render.Camera(cfg)
n = cfg.width * cfg.height * 3
st = torch.cuda.current_stream().cuda_stream
runs = []
for _ in range(4):
    buf = torch.zeros(n, device="cuda")
    cam.render_accumulate(scene, buf.data_ptr(), 0, spp, seed=11, stream=st)
    runs.append(buf)
torch.cuda.synchronize()
print(name, spp, "slots", os.environ.get("RTX_ITEM_SLOTS", "default"), "equal:", [bool(torch.equal(runs[0], r)) for r in runs[1:]],
      "max rel diff", float(((runs[0] - runs[1]).abs() / runs[0].abs().clamp_min(1.0)).max()))
