#!/usr/bin/env python3
"""Throughput benchmark of the MI355X render path.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched through torch.distributed.run)

A step is one complete frame of the workload (default C4: 3840x2160, depth 5,
4x4 anti-aliasing, 1 002 528-triangle heightfield under the reference's KD
builder at depth 22, one point + one spot light).  The frame is sharded as
interleaved 32x32 screen tiles over the N ranks (one process per GPU); each
rank renders its tiles into a packed device buffer through the C-ABI
(rt_render) and one gather per frame brings the RGB8 tiles to rank 0.  Total
work is fixed as N grows (strong scaling).

Rank 0 prints one JSON line: Mrays/s (primary + reflection + refraction +
shadow rays = Scene::intersect queries, counted on the device), the roofline
of the traversal kernels, and the CPU oracle timed on a bounded sample of the
same workload on this host.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
TILE = 32


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_scene(name, scale):
    """Returns (HostScene, width, height, render kwargs, description)."""
    from raytracer_amd import abi, host, scenes
    cfg = dict(scenes.CONFIGS[name])
    width = max(64, int(cfg["width"] * scale) // 8 * 8)
    height = scenes.frame_height(width, cfg["aspect"])
    t0 = time.time()
    if name == "C4" and scale < 1.0:
        text = scenes.heightfield_scene(n=max(16, int(708 * scale)))
    elif name == "C5" and scale < 1.0:
        text = scenes.many_lights_scene(n=max(16, int(500 * scale)))
    elif name == "C2" and scale < 1.0:
        text = scenes.mesh_scene(level=4)
    else:
        text = cfg["scene"]()
    hs = host.HostScene(text=text, kd_max_depth=cfg["kd_depth"])
    if cfg["cubemap"]:
        hs.set_cubemap(scenes.cube_map_faces(512, seed=3))
    kw = dict(depth=cfg["depth"], use_cubemap=cfg["cubemap"])
    if cfg["aa"]:
        kw.update(aa_mode=abi.RT_AA_FULL, aa_samples=cfg["aa"], jitter=host.jitter_table(1234567, cfg["aa"]))
    d = hs.desc
    desc = "%s %dx%d depth %d aa %s, %d triangles + %d objects, %d lights, kd depth %d leaf 10 (%d nodes)" % (
        name, width, height, cfg["depth"], ("%dx%d" % (cfg["aa"], cfg["aa"])) if cfg["aa"] else "off",
        d.n_faces, d.n_objects - d.n_meshes, d.n_lights, cfg["kd_depth"], d.n_nodes)
    log("[bench] scene %s built in %.1f s" % (desc, time.time() - t0))
    return hs, width, height, kw, desc


def cpu_baseline(hs, width, height, kw, budget_s):
    """The oracle (CPU restatement of the reference path, std::thread fan-out like
    CommandLineUI::run) on a bounded, evenly spread pixel sample of the same frame."""
    from raytracer_amd import host
    from oracle import loader
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    p, keep = host.render_params(width, height, **kw)
    npx = width * height
    stride = 7919 * 13 + 1  # spreads the sample over the whole frame

    def pixels(count):
        idx = (np.arange(count, dtype=np.int64) * stride) % npx
        return np.stack([idx % width, idx // width], axis=1).astype(np.int32)

    probe = 32 * cores
    t0 = time.time()
    _, st = loader.render_pixels(hs.desc_ptr, p, pixels(probe), threads=cores)
    dt = max(time.time() - t0, 1e-3)
    count = int(min(npx, max(probe, probe * budget_s / dt)))
    t0 = time.time()
    _, st = loader.render_pixels(hs.desc_ptr, p, pixels(count), threads=cores)
    dt = time.time() - t0
    rays = st.rays_total
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d of %d pixels (every %d-th, %d rays) of the same frame in %.1f s, "
                      "reference KD traversal, %d std::threads" % (count, npx, stride, rays, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C4", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink frame and mesh (smoke runs)")
    ap.add_argument("--traversal", default="fast", choices=["fast", "reference"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget; 0 = skip")
    ap.add_argument("--no-stats-pass", action="store_true")
    ap.add_argument("--out", default=None, help="write the frame as BMP (rank 0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, device, host, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RT_BENCH_FORCE_DIST=1: take the RCCL path (process group, gather, reductions)
    # even with one rank, so that a single-GPU box can exercise it
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if world != args.gpus and rank == 0:
        log("[bench] note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))

    hs, width, height, kw, desc = build_scene(args.config, args.scale)
    trav = abi.RT_TRAVERSE_FAST if args.traversal == "fast" else abi.RT_TRAVERSE_REFERENCE
    t_create = time.perf_counter()
    ds = device.DeviceScene(hs.desc_ptr)
    if rank == 0:
        log("[bench] rt_scene_create (validation, upload, BVH build): %.2f s" % (time.perf_counter() - t_create))
    pad_px = parallel.max_shard_pixels(width, height, TILE, TILE, world)
    # Two packed buffers, used in turn: the gather of frame k may still read its
    # buffer on the backend's stream while the library renders frame k+1 on its
    # own streams.  An event per buffer says when the gather that read it is done.
    packed = [torch.zeros((pad_px, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    packed_free = [None, None]
    gathered = [torch.empty_like(packed[0]) for _ in range(world)] if (rank == 0 and use_dist) else None
    frame = torch.zeros((height, width, 3), dtype=torch.uint8, device="cuda") if rank == 0 else None
    p, keep = host.render_params(width, height, traversal=trav, tile_w=TILE, tile_h=TILE,
                                 shard_index=rank, shard_count=world, out_is_device=True, **kw)
    torch.cuda.synchronize()   # the fills above run on torch's stream, the renders on the library's
    frames_done = [0]

    def step():
        k = frames_done[0] % 2
        frames_done[0] += 1
        if packed_free[k] is not None:
            packed_free[k].synchronize()
        st = ds.render_device(p, packed[k].data_ptr())
        bufs = parallel.gather_packed(packed[k], world, rank, force=use_dist, out=gathered)
        if rank == 0:
            for r, b in enumerate(bufs):
                device.unpack_tiles(b.data_ptr(), 3, width, height, TILE, TILE, r, world,
                                    frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if packed_free[k] is None:
            packed_free[k] = torch.cuda.Event()
        packed_free[k].record()
        return st

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = 0
    sums = np.zeros(6)          # ms_trace, trace launches, ms_lights, lights launches, ms_device, batches
    for _ in range(args.steps):
        st = step()
        rays += st.rays_total
        sums += [st.ms_trace, st.trace_launches, st.ms_lights, st.lights_launches, st.ms_device, st.batches]
    fence()
    elapsed = time.perf_counter() - t0
    tvals = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    cvals = torch.tensor([float(rays)] + list(sums), dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tvals, op=dist.ReduceOp.MAX)
        dist.all_reduce(cvals, op=dist.ReduceOp.SUM)
    elapsed = float(tvals[0])
    total_rays, trace_ms, trace_n, lights_ms, lights_n, total_dev_ms, total_batches = [float(x) for x in cvals.tolist()]

    # Untimed pass with device counters: tree nodes visited (V) and primitives
    # tested (T), split between the closest-hit kernel (k_trace: primary,
    # reflection and refraction rays) and the shadow-query kernel
    # (k_shadow).  Algorithmic bytes of a ray: node_bytes V + 128 T + 160 (node
    # record: 64 B BVH node on the fast path, 16 B KD node on the reference path;
    # 128 B triangle record; an 80 B ray record written and read once through the
    # level queues).  The roofline entry is k_trace, the kernel with the largest
    # share of the frame; its launch duration comes from HIP events recorded
    # around every launch on the library's stream during the timed steps.
    roof = None
    if not args.no_stats_pass:
        ps, keep2 = host.render_params(width, height, traversal=trav, tile_w=TILE, tile_h=TILE,
                                       shard_index=rank, shard_count=world, out_is_device=True,
                                       collect_stats=True, **kw)
        torch.cuda.synchronize()
        sst = ds.render_device(ps, packed[0].data_ptr())
        svals = torch.tensor([float(sst.nodes_visited), float(sst.prims_tested), float(sst.rays_total),
                              float(sst.shadow_nodes_visited), float(sst.shadow_prims_tested),
                              float(sst.rays_shadow)], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(svals, op=dist.ReduceOp.SUM)
        V, T, R, Vs, Ts, Rs = [float(x) for x in svals.tolist()]
        node_bytes = 64.0 if args.traversal == "fast" else 16.0

        def kernel_line(v, t, r, ms, n):
            frame_bytes = node_bytes * v + 128.0 * t + 160.0 * r
            per_launch = frame_bytes * args.steps / max(n, 1.0)
            ms_launch = ms / max(n, 1.0)
            ach = per_launch / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
            return {"bytes_per_ray": round(frame_bytes / max(r, 1.0), 1), "achieved": round(ach, 3),
                    "bytes_per_launch": round(per_launch), "ms_per_launch": round(ms_launch, 4),
                    "launches_per_step": round(n / args.steps, 1), "rays_per_step": round(r),
                    "nodes_per_ray": round(v / max(r, 1.0), 2), "prims_per_ray": round(t / max(r, 1.0), 2),
                    "Grays_per_s": round(r * args.steps / max(ms, 1e-9) / 1e6, 3)}

        kt = kernel_line(V - Vs, T - Ts, R - Rs, trace_ms, trace_n)
        kl = kernel_line(Vs, Ts, Rs, lights_ms, lights_n)
        traffic = None
        profiled = {}
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tpath) and args.scale == 1.0 and world == 1 and args.traversal == "fast":
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("k_trace_hbm_bytes_per_launch")
                busy, lanes = tj.get("k_trace_valu_busy"), tj.get("k_trace_lanes_active")
                profiled = {"valu_busy_profiled": busy, "lanes_active_profiled": lanes, "profile": tj.get("source"),
                            # the VALU roof of the same kernel, from the committed counter pass: share of
                            # cycles in which a SIMD issues a vector instruction, and of those the
                            # share of lanes that do work
                            "valu": {"bound": "valu issue", "issue_busy_frac": busy, "lanes_active_of_64": lanes,
                                     "frac": round(busy * lanes / 64.0, 4) if busy and lanes else None}}
            except Exception:
                traffic = None
        roof = {"bound": "hbm",
                "kernel": "k_trace (%s)" % ("BVH closest-hit traversal" if args.traversal == "fast"
                                            else "reference KD traversal"),
                "achieved": kt["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(kt["achieved"] / HBM_PEAK_GBS, 6), "traffic": traffic,
                "note": "algorithmic bytes are served by L1/L2 (the tree is shared by neighbouring rays): "
                        "measured HBM traffic is a small fraction of them, the kernel is bound by VALU "
                        "issue (see valu_busy_profiled) and vector-memory latency, not by HBM",
                "k_trace": kt, "k_shadow": kl}
        roof.update(profiled)

    if rank == 0:
        if args.out:
            host.write_bmp(args.out, frame.cpu().numpy())
        cpu = None
        if args.cpu_seconds > 0 and world == 1:   # the CPU leg belongs to the single-GPU line only
            cpu = cpu_baseline(hs, width, height, kw, args.cpu_seconds)
        value = total_rays / elapsed / 1e6
        out = {
            "metric": "Mrays/s (primary+secondary+shadow)",
            "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "traversal": args.traversal, "tile": TILE,
                       "rays_per_step": round(total_rays / args.steps),
                       "sharding": "interleaved %dx%d tiles over %d rank(s), one gather per frame" % (TILE, TILE, world),
                       "trace_ms_per_step": round(trace_ms / args.steps / world, 3),
                       "lights_ms_per_step": round(lights_ms / args.steps / world, 3),
                       "device_ms_per_step": round(total_dev_ms / args.steps / world, 3),
                       "batches_per_step": round(total_batches / args.steps / world, 2)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    ds.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
