"""`python -m slr_amd.host scene.txt [--spectral] [--samples N] [--out DIR]` — the HostProgram of the reference
(HostProgram/main.cpp:20-62: read a scene script, build, render) on the HIP path.

Reads a scene in the reference's scene language (slr_amd/scene_language.py), renders it with the path tracer and leaves
behind what PathTracingRenderer::render leaves behind (PathTracingRenderer.cpp:83-94): "%03u.bmp" after 1, 2, 4, ...
samples with scale brightness / samples, and one stdout line "%u samples: %s, %g[s]" per image, at most 16 images.
A scene that asks for another renderer ("BPT", ...) is rendered with the path tracer and a note on stderr.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

from . import abi, binding, scene_language


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m slr_amd.host")
    ap.add_argument("scene")
    ap.add_argument("--spectral", action="store_true", help="16 wavelength samples per path (the reference's default build)")
    ap.add_argument("--samples", type=int, default=0, help="override the script's sample count")
    ap.add_argument("--out", default=".")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    try:
        scene, settings, renderer = scene_language.load_scene(args.scene)
    except scene_language.SceneLanguageError as e:
        print("failed to read the scene: %s" % e, file=sys.stderr)
        return -1                                                            # main.cpp:39-42
    if renderer["method"] != "PT":
        print("note: the scene asks for %r; only the unidirectional path tracer is built, rendering with it" % renderer["method"], file=sys.stderr)
    spp = args.samples or int(renderer["samples"])
    st = abi.RenderSettings(int(settings["width"]), int(settings["height"]), float(settings["timeStart"]), float(settings["timeEnd"]),
                            float(settings["brightness"]), int(settings["rngSeed"]))
    ctx = binding.Context(device=args.device, mode=abi.MODE_SPECTRAL if args.spectral else abi.MODE_RGB)
    ctx.upload_scene(scene)
    ctx.render_begin(st)
    cam = scene.camera
    sensitivity = cam.sensitivity if cam.sensitivity > 0 else float(np.float32(1.0 / (np.pi * float(np.float32(cam.lens_radius)) ** 2))) if cam.lens_radius > 0 else 1.0
    w, h = st.image_width, st.image_height
    bmp = np.zeros((3 * w + w % 4) * h, np.uint8)
    lib = ctx.lib
    start = time.time()
    done, export, img = 0, 1, 0
    while done < spp and img < 16:
        upto = min(export, spp)
        ctx.render(done, upto - done)
        done = upto
        if done == export:
            fb = ctx.read_framebuffer()
            name = "%03u.bmp" % img
            scale = float(np.float32(np.float32(st.brightness) / np.float32(done)) * np.float32(sensitivity))
            binding._check(lib, lib.slrhip_tonemap_bgr8(fb.ctypes.data, w, h, ctx.components, C.c_float(scale), bmp.ctypes.data, bmp.size), "slrhip_tonemap_bgr8")
            binding._check(lib, lib.slrhip_save_bmp(os.path.join(args.out, name).encode(), bmp.ctypes.data, w, h), "slrhip_save_bmp")
            print("%u samples: %s, %g[s]" % (export, name, time.time() - start), flush=True)
            img += 1
            export += export
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
