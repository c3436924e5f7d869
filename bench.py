#!/usr/bin/env python
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one complete render of BASELINE.json's configs[1]: the Cornell-box-spheres scene
(Lambert + specular only), RGB, 1280x720, 1024 spp, through the C ABI (slrhip_render_begin ->
slrhip_render -> slrhip_resolve_framebuffer), with the scene already resident in HBM.  With N > 1
the 8x8 image tiles are dealt round-robin to the ranks (one process per GPU, no exchange while
rendering) and the step ends with ONE RCCL reduce of the float framebuffer to rank 0 over xGMI;
the total work is fixed, so the scaling is "strong".  value = Msamples/s of the whole job.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L2_PEAK_GBS = 34500.0          # aggregate L2 bandwidth (MI355X_MICROARCH.md, "L2 (per XCD)")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s peak, ~6.3 TB/s achievable)

# Algorithmic bytes ("Kernels and their rooflines").  Per WORKLOAD: the node record the traversal kernel reads is the
# 128-byte QNode, or the 64-byte quantized QNodeQ once the tree has >= 65 536 nodes (slrhip_upload_scene); the shade kernel's
# slot record depends on the colour mode.
TRI_BYTES = 48
QUANT_NODE_THRESHOLD = 65536
CLOSEST_RAY_BYTES = 4 + 16 + 16 + 16       # state flag, ray origin, ray direction, hit record written
SHADOW_RAY_BYTES = 4 + 16 + 16 + 4         # queue entry, origin, direction+distMax, visibility written
# k_shade per live slot.  RGB: state read 136 (flags 4, rng 16, alpha 16, sp pair 32, nee 16, hit 16, ray 32, visible 4)
# + ShadeTri 96 + written ~140 + shadow entry ~20 + material 80 (LDS) + per finished path (x 0.44 per visit) the sample's result
# 16 + header 32 + the restarted sample's state 68 (round 3: the 64-byte accumulator read-modify-write per sample is gone).
# Spectral: read 156 (flags, rng, alpha 64 + pdf 4, hit, ray, visible, hdr) + ShadeTri 96 + written 120 (flags, rng, alpha 68,
# ray 32) + pending light sample 64 x 0.55 + radiance-sum read-modify-write 256 x 0.3 (only when a contribution arrives)
SHADE_SLOT_BYTES = {"rgb": 567, "spectral": 484}


def node_bytes(num_nodes):
    return 64 if num_nodes >= QUANT_NODE_THRESHOLD else 128


class quiet_stdout:
    """The reference prints progress lines with printf (PathTracingRenderer.cpp:89, SBVH.h:405); keep the
    process's stdout for the one JSON line by pointing fd 1 at stderr while the CPU legs run."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=0, help="0 = the workload's BASELINE value")
    ap.add_argument("--right-sphere", default="matte", choices=["matte", "glass"])
    ap.add_argument("--workload", default="cornell", choices=["cornell", "boxes_spectral", "ibl", "grid10m"],
                    help="cornell = BASELINE configs[1] (the headline metric); the others are configs[2..4], measured the same way")
    ap.add_argument("--grid-n", type=int, default=2236, help="grid10m: cells per side (2 n^2 triangles)")
    ap.add_argument("--instanced", action="store_true",
                    help="grid10m as BASELINE configs[4] words it, an INSTANCED mesh: one 8 192-triangle patch placed 1 250 times (two-level traversal)")
    ap.add_argument("--stripes", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--tree", default="auto", choices=["auto", "host", "device"],
                    help="accelerator build: host binned SAH, device LBVH (SLRHIP_FLAG_BVH_DEVICE_BUILD), auto = device from 2^20 triangles on")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="per-kernel HBM traffic from separate rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py)")
    return ap.parse_args()


def traffic_entry(args, W, H, spp):
    """PMC traffic per launch for EXACTLY this workload, image size and pass count (the share of nearly empty iterations at
    the end of a render — and with it the average bytes per launch — changes with the pass count); {} when no pass covers it,
    and the roofline figure is then a modelled one under its own key, never `frac`."""
    table = json.load(open(args.traffic_json))
    return table.get("%s%s_%dx%d_%dspp" % (args.workload, "_instanced" if args.instanced else "", W, H, spp), {})


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks as CHILD processes (torch.distributed.run, one per GPU,
    rendezvous on 127.0.0.1) before this process has touched the GPU, pass their output through and exit with their status."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus > 1:
        self_launch(args)
    import torch
    import torch.distributed as dist

    from slr_amd import Context, abi, distributed, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal of the N-rank flow on a box with ONE GPU (SLRHIP_BENCH_SHARE_GPU=1): every rank renders its shard on device 0 and
    # the frames meet over gloo.  It checks the launch, sharding, reduce, barrier and reporting code; its timings mean nothing.
    rehearsal = os.environ.get("SLRHIP_BENCH_SHARE_GPU") == "1" and world > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed.init("gloo" if rehearsal else "nccl")

    W, H = args.width, args.height
    mode, ref_name = abi.MODE_RGB, "ref_rgb"
    if args.workload == "cornell":
        spp = args.spp or 1024
        scene = scenes.cornell_box_spheres(W / H, 48, 24, args.right_sphere)
        what = ("BASELINE configs[1]: Cornell_Box_Spheres walls/light/camera (TestScenes/Cornell_Box_Spheres.txt:8-107,132-138) + two "
                "tessellated spheres (aluminium mirror, %s)" % ("Lambert" if args.right_sphere == "matte" else "BK7 glass"))
    elif args.workload == "boxes_spectral":
        spp = args.spp or 1024
        mode, ref_name = abi.MODE_SPECTRAL, "ref_spectral"
        scene = scenes.cornell_box_boxes(W / H)
        what = "BASELINE configs[2]: Cornell_Box_Boxes-shaped, GGX titanium box + matte box, 16-wavelength spectral mode"
    elif args.workload == "ibl":
        spp = args.spp or 2048
        scene = scenes.ibl_test_scene(W / H, (2048, 1024), 48, 24)
        what = "BASELINE configs[3]: IBL_Test-shaped, 24-patch floor + mirror sphere under a synthetic 2048x1024 binary16 sky (scale 4)"
    else:
        spp = args.spp or 4096
        if args.instanced:
            scene = scenes.instanced_grid(25, 50, 64, W / H)
            what = ("BASELINE configs[4], instanced: one 8 192-triangle heightfield patch placed 1 250 times (non-uniform scales, half turns), "
                    "matte, one area light, thin lens r=0.025")
        else:
            scene = scenes.displaced_grid(args.grid_n, W / H)
            what = "BASELINE configs[4]: one displaced grid (hash-noise heightfield, seed 20240611), matte, one area light, thin lens r=0.025"
    settings = abi.RenderSettings(W, H, 0.0, 0.0, 1.0, abi.DEFAULT_SEED)
    build_flag = abi.FLAG_BVH_DEVICE_BUILD if args.tree == "device" else 0
    if args.tree == "host":
        os.environ["SLRHIP_BVH"] = "host"
    flags = (0 if args.no_kernel_timing else abi.FLAG_TIME_KERNELS) | build_flag
    ctx = Context(device=local_rank, mode=mode, stripes=args.stripes, flags=flags)
    ctx.upload_scene(scene)
    comps = 16 if mode == abi.MODE_SPECTRAL else 3
    fb = torch.zeros((H, W, comps), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    shard_ms = []

    def step():
        # shard render -> resolve -> one RCCL sum-reduce to rank 0 (disjoint tile supports: sum == gather)
        t = time.perf_counter()
        ctx.render_begin(settings, shard=distributed.shard_for(rank, world))
        ctx.render(0, spp, stream)             # blocks the host until this rank's passes are done
        shard_ms.append((time.perf_counter() - t) * 1e3)
        ctx.resolve_into(fb.data_ptr(), fb.numel(), stream)
        distributed.reduce_framebuffer(fb, world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    del shard_ms[:]
    prof0 = ctx.profile()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof1 = ctx.profile()
    counters = ctx.counters()
    rank_ms = [sum(shard_ms) / max(len(shard_ms), 1)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own shard-render time (host clock around slrhip_render), gathered so that rank 0 can report the spread
        mine = torch.tensor(rank_ms, dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(x.item()) for x in every]

    total_samples = float(W) * H * spp * args.steps
    value = total_samples / elapsed / 1e6

    out = {
        "metric": "Msamples/sec @1024spp 1280x720 (unidirectional path tracing, Cornell_Box_Spheres-shaped scene)" if args.workload == "cornell"
                  else "Msamples/sec @%dspp %dx%d (unidirectional path tracing, workload %s)" % (spp, W, H, args.workload + (" instanced" if args.instanced else "")),
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s, %d triangles, %s mode, %dx%d, %d spp, seed %d" % (
                       what, len(scene.triangles), "spectral" if mode == abi.MODE_SPECTRAL else "RGB", W, H, spp, abi.DEFAULT_SEED),
                   "sharding": "8x8 tiles round-robin over %d rank(s), one RCCL reduce of the framebuffer per step" % world,
                   "stripes": int(args.stripes), "tree": "device LBVH" if (args.tree == "device" or (args.tree == "auto" and len(scene.triangles) >= 1 << 20)) else "host binned SAH"},
        # evidence that the collective saw N ranks: the process group's backend and size as torch.distributed reports
        # them, and each rank's own render time for its shard (ms per step, host clock around slrhip_render)
        "backend": (dist.get_backend() if world > 1 else None), "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
        "shard_render_ms": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3), "per_rank": [round(x, 3) for x in rank_ms]},
    }
    if rehearsal:
        out["config"]["rehearsal"] = "all %d ranks share GPU 0, frames reduced over gloo: timings are not a measurement" % world

    if rank == 0:
        # ---- per-kernel timing over the timed region (HIP events on the render stream) ------------
        roof = None
        kernels = {}
        if not args.no_kernel_timing:
            for k, name in enumerate(abi.KERNEL_NAMES):
                n = prof1.launches[k] - prof0.launches[k]
                ms = prof1.milliseconds[k] - prof0.milliseconds[k]
                kernels[name] = {"launches": int(n), "ms_total": round(ms, 3), "avg_us": round(ms / n * 1e3, 3) if n else None}
            if kernels.get("tail", {}).get("launches") == 0:
                kernels.pop("tail")       # listed only when it ran (SLRHIP_TAIL_SLOTS=0 turns it off)
            # traversal statistics from an instrumented, untimed pass on this rank's shard
            cctx = Context(device=local_rank, mode=mode, stripes=args.stripes, flags=abi.FLAG_COUNT_TRAVERSAL | build_flag)
            cctx.upload_scene(scene)
            cctx.render_begin(settings, shard=(rank, world))
            cctx.render(0, min(spp, 16))
            cp = cctx.profile()
            cc = cctx.counters()
            cctx.close()
            nodes_c, tris_c = cp.nodes[0] / max(cp.rays[0], 1), cp.triangles[0] / max(cp.rays[0], 1)
            nodes_s, tris_s = cp.nodes[1] / max(cp.rays[1], 1), cp.triangles[1] / max(cp.rays[1], 1)
            ext_per_sample = cc.extension_rays / max(cc.samples, 1)
            shd_per_sample = cc.shadow_rays / max(cc.samples, 1)
            NODE_BYTES = node_bytes(int(counters.bvh_nodes))
            slot_bytes = SHADE_SLOT_BYTES["spectral" if comps == 16 else "rgb"]
            per_ray = {"closest": nodes_c * NODE_BYTES + tris_c * TRI_BYTES + CLOSEST_RAY_BYTES,
                       "shadow": nodes_s * NODE_BYTES + tris_s * TRI_BYTES + SHADOW_RAY_BYTES}
            shard_samples = float(counters.samples) * args.steps     # sample totals restart at every render_begin
            rays = {"closest": shard_samples * ext_per_sample, "shadow": shard_samples * shd_per_sample}
            # algorithmic bytes of the timed region per kernel class: k_trace_ws traces both ray kinds in one launch;
            # k_shade visits every live slot once per iteration = once per extension ray (+ idle tail ignored)
            alg_bytes = {"trace": rays["closest"] * per_ray["closest"] + rays["shadow"] * per_ray["shadow"],
                         "shade": rays["closest"] * slot_bytes}
            # the tail kernel (the last paths of a frame, one launch) is listed with the others but is not a roofline subject
            dom = max((n for n in kernels if n != "tail"), key=lambda n: kernels[n]["ms_total"])
            selection = "largest total time over the timed region"
            # The two kernels of an iteration take the same share of the time to within a per cent or two, so which one is ahead
            # changes from run to run and with it a factor of 2.5 in an HBM fraction (the shade kernel is HBM-bound, the traversal
            # kernel is bound by the L1 request path).  A near-tie (within 5 %) goes to the kernel that moves more HBM bytes per
            # launch by PMC: this is an HBM roofline.  `per_kernel` below always carries both.
            try:
                ent0 = traffic_entry(args, W, H, spp) if world == 1 else {}
            except (OSError, ValueError):
                ent0 = {}
            for other in kernels:
                if other in ("tail", dom) or other not in ent0 or dom not in ent0:
                    continue
                if kernels[other]["ms_total"] >= 0.95 * kernels[dom]["ms_total"] and \
                        ent0[other]["traffic_bytes_per_launch"] > ent0[dom]["traffic_bytes_per_launch"]:
                    selection = ("%s and %s within 5 %% of each other in total time (%.1f vs %.1f ms): the one with more HBM bytes per launch"
                                 % (dom, other, kernels[dom]["ms_total"], kernels[other]["ms_total"]))
                    dom = other
            launches = max(kernels[dom]["launches"], 1)
            bytes_per_launch = alg_bytes[dom] / launches
            avg_s = kernels[dom]["ms_total"] / launches * 1e-3
            algorithmic = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
            # `achieved` / `frac` are HBM figures: bytes that reached HBM per launch (PMC, below) over the launch time.  The
            # algorithmic bytes of SURVEY 8d count every node and triangle a ray touches; on a tree that fits L2 / the Infinity
            # Cache most of them never reach HBM, so that rate is reported next to it as a cache-side figure, never as `frac`.
            roof = {"bound": "hbm", "kernel": dom, "kernel_selection": selection, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None, "traffic": None, "basis": None,
                    "algorithmic_bytes_per_launch": round(bytes_per_launch), "algorithmic_rate": round(algorithmic, 2),
                    "frac_of_l2_peak": round(algorithmic / L2_PEAK_GBS, 5),
                    "node_bytes": NODE_BYTES, "shade_slot_bytes": slot_bytes, "avg_launch_us": kernels[dom]["avg_us"],
                    "per_ray": {"nodes_closest": round(nodes_c, 3), "tris_closest": round(tris_c, 3), "nodes_shadow": round(nodes_s, 3),
                                "tris_shadow": round(tris_s, 3), "extension_rays_per_sample": round(ext_per_sample, 4),
                                "shadow_rays_per_sample": round(shd_per_sample, 4)}}
            # HBM bytes per launch of that kernel from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, KiB; collected in their
            # own rocprofv3 runs of THIS command line and committed under profiles/) -- null when no pass covers this exact workload
            ent = {}
            try:
                ent = traffic_entry(args, W, H, spp) if world == 1 else {}
            except (OSError, ValueError):
                pass
            if dom in ent:
                tr = ent[dom]
                roof["traffic"] = round(tr["traffic_bytes_per_launch"])
                roof["traffic_source"] = os.path.relpath(args.traffic_json, ROOT)
                roof["achieved"] = round(tr["traffic_bytes_per_launch"] / avg_s / 1e9, 1)
                roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 5)
                roof["basis"] = ("pmc: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch of this kernel from %s (separate rocprofv3 runs of "
                                 "this command) / its average launch time in THIS run" % roof["traffic_source"])
            else:
                # no PMC pass covers this workload / pass count / world size: `frac` stays null.  A MODELLED figure under its own
                # key: the per-ray and per-slot STATE records (which are streamed from HBM) plus the node and triangle bytes only
                # when the tree exceeds the 256 MB Infinity Cache
                tree_bytes = int(counters.bvh_nodes) * NODE_BYTES + len(scene.triangles) * TRI_BYTES
                cached = tree_bytes < 256e6
                if dom == "shade":
                    model = bytes_per_launch
                else:
                    state = (rays["closest"] * CLOSEST_RAY_BYTES + rays["shadow"] * SHADOW_RAY_BYTES) / launches
                    model = state if cached else bytes_per_launch
                roof["modelled_achieved"] = round(model / avg_s / 1e9, 1)
                roof["modelled_frac"] = round(roof["modelled_achieved"] / HBM_PEAK_GBS, 5)
                roof["basis"] = "no PMC entry for this exact workload: frac is null; modelled_frac = state records%s / launch time" % (
                    "" if cached or dom == "shade" else " + node/triangle bytes")
            # every kernel class of an iteration with a PMC entry, not only the dominant one (the two kernels of an iteration take
            # the same share of the time to within a per cent or two, so which one is "dominant" changes from run to run)
            if ent:
                roof["per_kernel"] = {}
                for k in kernels:
                    if k in ent and kernels[k]["launches"]:
                        a = ent[k]["traffic_bytes_per_launch"] / (kernels[k]["ms_total"] / kernels[k]["launches"] * 1e-3) / 1e9
                        roof["per_kernel"][k] = {"traffic": round(ent[k]["traffic_bytes_per_launch"]), "avg_launch_us": kernels[k]["avg_us"],
                                                 "achieved": round(a, 1), "frac": round(a / HBM_PEAK_GBS, 5)}
            # whole-sample algorithmic bytes (SURVEY 8d formula) for reference
            sample_bytes = (ext_per_sample * per_ray["closest"] + shd_per_sample * per_ray["shadow"] + ext_per_sample * slot_bytes)
            roof["algorithmic_bytes_per_sample"] = round(sample_bytes, 1)
            # whole-iteration HBM figure: PMC traffic of both kernels of an iteration over their summed launch times
            it_kernels = [k for k in kernels if k != "tail"]
            if ent and all(k in ent for k in it_kernels):
                tot_b = sum(ent[k]["traffic_bytes_per_launch"] for k in it_kernels)
                tot_s = sum(kernels[k]["ms_total"] / max(kernels[k]["launches"], 1) for k in it_kernels) * 1e-3
                roof["iteration_traffic_bytes"] = round(tot_b)
                roof["iteration_frac"] = round(tot_b / tot_s / 1e9 / HBM_PEAK_GBS, 5)
        out["roofline"] = roof
        out["kernels"] = kernels
        out["counters"] = {"samples": int(counters.samples), "extension_rays": int(counters.extension_rays),
                           "shadow_rays": int(counters.shadow_rays), "iterations": int(counters.iterations),
                           "bvh_nodes": int(counters.bvh_nodes), "bvh_depth": int(counters.bvh_depth),
                           "build_seconds": round(counters.build_seconds, 4)}

        # ---- CPU baseline and matched-seed parity (N = 1 only; the checker, never the thing measured) ---
        out["cpu_baseline"] = None
        if world == 1 and args.cpu_seconds > 0:
            guard = quiet_stdout()
            guard.__enter__()
            from oracle import binding as ob
            # the GPU box gives a 1-GPU job a 16-CPU share; never start more workers than that
            cores = max(1, min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16))
            ref = ob.load(ref_name, mode) if len(scene.triangles) <= 1000000 else None   # the reference's SBVH build of 10 M triangles takes minutes
            if ref is not None:
                rs = ref.scene(scene)
                _, sec = ob.render_native(rs, settings, 1, cores)
                n = int(max(1, min(64, args.cpu_seconds / max(sec, 1e-3))))
                _, sec = ob.render_native(rs, settings, n, cores)
                out["cpu_baseline"] = {"value": round(W * H * n / sec / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "reference",
                                       "sample": "libSLR PathTracingRenderer::render unmodified (%s build, SBVH), %dx%d x %d spp of the "
                                                 "same scene, %d worker threads" % ("spectral" if comps == 16 else "RGB", W, H, n, cores)}
            orc = ob.load("oracle", mode).scene(scene)
            t = time.perf_counter()
            want, _ = orc.render(settings, 1, threads=cores)
            sec1 = time.perf_counter() - t
            n = int(max(1, min(16, 0.5 * args.cpu_seconds / max(sec1, 1e-3))))
            t = time.perf_counter()
            want, octr = orc.render(settings, n, threads=cores)
            sec = time.perf_counter() - t
            port = {"value": round(W * H * n / sec / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                    "sample": "oracle restatement, %dx%d x %d spp of the same scene, %d threads" % (W, H, n, cores)}
            if out["cpu_baseline"] is None:
                out["cpu_baseline"] = port
            else:
                out["cpu_baseline_port"] = port
            if not args.no_parity:
                # the TIMED configuration (automatic slot count): the sensor adds in pass order, so the slot count changes no bit
                pctx = Context(device=local_rank, mode=mode, stripes=args.stripes, flags=build_flag)
                got = pctx.render_image(scene, settings, n)
                pc = pctx.counters()
                pctx.close()
                sens = 1.0 / (np.pi * 0.025 ** 2)
                d = (got.astype(np.float64) - want) / n * sens
                nz = np.abs(want) > 1e-9
                rel = np.abs(got.astype(np.float64) - want)[nz] / np.abs(want[nz])
                out["parity"] = {"spp": n, "configuration": "as timed (stripes = %d: automatic slot count, wave work queues)" % args.stripes if args.stripes == 0
                                                            else "as timed (stripes = %d)" % args.stripes,
                                 "rmse_vs_cpu_matched_seeds": float(np.sqrt(np.mean(d * d))),
                                 "max_rel_err": float(rel.max()) if rel.size else 0.0,
                                 "fraction_within_2e-6": float(np.isclose(got, want, rtol=2e-6, atol=1e-9).mean()),
                                 "bit_exact_fraction": float(((got.view(np.uint32) == want.view(np.uint32)) | ((got == 0) & (want == 0))).mean()),
                                 "ray_counts_equal": bool(int(pc.extension_rays) == int(octr.extension_rays) and int(pc.shadow_rays) == int(octr.shadow_rays)),
                                 "mean_radiance": float(want.mean() / n * sens)}
                # and with ONE slot per pixel: the same frame to the last bit (kept as a cross-check of that claim)
                pctx = Context(device=local_rank, mode=mode, stripes=1, flags=build_flag)
                got1 = pctx.render_image(scene, settings, n)
                pctx.close()
                exact = (got1.view(np.uint32) == want.view(np.uint32)) | ((got1 == 0) & (want == 0))
                out["parity"]["stripes1_bit_exact_fraction"] = float(exact.mean())
                out["parity"]["stripes1_equals_timed_configuration"] = bool(np.array_equal(got1.view(np.uint32), got.view(np.uint32)))
                if comps == 16:
                    # spectral RMSE on the 16 bins (above) and after DiscretizedSpectrum::getRGB (SpectrumTypes.h:702-721)
                    from slr_amd import spectra
                    t = spectra.tables()["cmf16"].astype(np.float64)
                    m = np.array([[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]])
                    to_rgb = lambda fb: ((fb.astype(np.float64) / n * sens) @ t[:48].reshape(3, 16).T / t[48]) @ m.T
                    drgb = to_rgb(got) - to_rgb(want)
                    out["parity"]["rmse_after_getRGB"] = float(np.sqrt(np.mean(drgb * drgb)))
            # the restatement on ONE host core, a small sample: 320x180 x 4 spp of the same scene
            t1 = time.perf_counter()
            orc.render(abi.RenderSettings(W // 4, H // 4, 0.0, 0.0, 1.0, abi.DEFAULT_SEED), 4, threads=1)
            sec1 = time.perf_counter() - t1
            out["cpu_baseline_port_1thread"] = {"value": round((W // 4) * (H // 4) * 4 / sec1 / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
                                                "sample": "oracle restatement, %dx%d x 4 spp of the same scene, 1 thread" % (W // 4, H // 4)}
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
            guard.__exit__()
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
