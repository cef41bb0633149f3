"""Command line of the renderer: `python -m ray-tracer_amd [OPTIONS] <SCENE>`.

Mirrors the reference binary (main.rs:543-585): the same positional scene names (clap's kebab-case ValueEnum,
main.rs:17-28), the same options -s/--samples, -w/--width, --height, -d/--depth, -v/--vfov overriding the scene's camera
(main.rs:567-574), and the result written to `image.png` in the working directory.  The render itself is the GPU
library's; there is no CPU path.  Options the reference does not have are grouped under "backend".
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import time

_pkg = __package__ or "ray-tracer_amd"

REFERENCE_SCENES = ["bouncing-spheres", "checkered-spheres", "earth", "perlin-spheres", "quads", "simple-light",
                    "cornell-box", "cornell-smoke", "final-scene"]


def parser(scene_names):
    ap = argparse.ArgumentParser(prog="ray-tracer", description="Path tracer on MI355X (scenes and options of the reference CLI)")
    ap.add_argument("-s", "--samples", type=int, default=None, help="samples per pixel")
    ap.add_argument("-w", "--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("-d", "--depth", type=int, default=None, help="maximum number of ray segments per sample")
    ap.add_argument("-v", "--vfov", type=float, default=None, help="vertical field of view in degrees")
    ap.add_argument("scene", choices=scene_names, metavar="SCENE", help=" | ".join(scene_names))
    be = ap.add_argument_group("backend")
    be.add_argument("--seed", type=int, default=None, help="seed of the scene's random placement and of the sample streams "
                    "(default: from the OS, as the reference's thread_rng)")
    be.add_argument("--traversal", choices=["fast", "strict"], default="fast",
                    help="strict walks the reference's own BVH cull for cull; fast uses the library's SAH tree")
    be.add_argument("-o", "--output", default="image.png", help="output file (the reference always writes image.png)")
    be.add_argument("--quiet", action="store_true")
    return ap


def main(argv=None):
    scenes = importlib.import_module(_pkg + ".scenes")
    names = REFERENCE_SCENES + sorted(n for n in scenes.WORLDS if n not in REFERENCE_SCENES)
    args = parser(names).parse_args(argv)
    for flag in ("samples", "width", "height", "depth"):
        v = getattr(args, flag)
        if v is not None and v < 0:
            print(f"error: invalid value '{v}' for '--{flag}': must not be negative", file=sys.stderr)
            return 2
    seed = args.seed if args.seed is not None else int.from_bytes(os.urandom(4), "little")
    overrides = {}
    if args.samples is not None:
        overrides["samples_per_pixel"] = args.samples
    if args.width is not None:
        overrides["width"] = args.width
    if args.height is not None:
        overrides["height"] = args.height
    if args.depth is not None:
        overrides["max_depth"] = args.depth
    if args.vfov is not None:
        overrides["vfov"] = args.vfov
    cfg, world = scenes.make(args.scene, seed=seed, **overrides)

    render = importlib.import_module(_pkg + ".render")
    png = importlib.import_module(_pkg + ".png")
    t0 = time.time()
    scene, cam = render.Scene(world), render.Camera(cfg)
    mode = render.TRAVERSAL_STRICT if args.traversal == "strict" else render.TRAVERSAL_FAST
    rgb, st = cam.render(scene, seed=seed, traversal=mode, want_stats=True)
    png.write_rgb8(args.output, rgb)
    if not args.quiet:
        print(f"{args.scene}: {cfg.width}x{cfg.height}, {cfg.samples_per_pixel} spp, depth {cfg.max_depth}, seed {seed}: "
              f"kernel {st.kernel_ms:.1f} ms ({st.samples / max(st.kernel_ms, 1e-9) / 1e3:.0f} Msamples/s, "
              f"{st.rays / max(st.samples, 1):.2f} rays/sample), total {time.time() - t0:.2f} s -> {args.output}", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
