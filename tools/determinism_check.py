"""Same scene, same seeds, rendered by the batch and by the wave-specialised schedule (twice each): framebuffer hashes
and ray counts must all agree (run with no SLRHIP_TRACE set; the script sets it per subprocess)."""
import hashlib, os, subprocess, sys
CODE = r'''
import sys, hashlib
sys.path.insert(0, ".")
import numpy as np
from slr_amd import Context, abi, scenes
W, H, SPP = 1280, 720, int(sys.argv[1])
c = Context()
fb = c.render_image(scenes.cornell_box_spheres(W / H, 48, 24, "matte"), abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED), SPP)
k = c.counters()
print(hashlib.sha1(fb.tobytes()).hexdigest()[:16], k.extension_rays, k.shadow_rays, k.samples)
'''
spp = sys.argv[1] if len(sys.argv) > 1 else "128"
for mode in ("batch", "ws", "batch", "ws"):
    env = dict(os.environ, SLRHIP_TRACE=mode)
    out = subprocess.run([sys.executable, "-c", CODE, spp], env=env, capture_output=True, text=True)
    print(mode, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:], flush=True)
