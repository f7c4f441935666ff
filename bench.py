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
# FP64 vector peak with contraction off: the vendor's 78.6 TFLOP/s counts a fused
# multiply-add as two operations; the path may not fuse (bit parity with the
# reference's x86-64 arithmetic), so an instruction slot carries one.
FP64_VALU_PEAK_TFLOPS = 39.3
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
    desc = "%s %dx%d depth %d aa %s, %d triangles + %d objects, %d lights; caller KD-tree: depth %d leaf 10 (%d nodes)" % (
        name, width, height, cfg["depth"], ("%dx%d" % (cfg["aa"], cfg["aa"])) if cfg["aa"] else "off",
        d.n_faces, d.n_objects - d.n_meshes, d.n_lights, cfg["kd_depth"], d.n_nodes)
    log("[bench] scene %s built in %.1f s" % (desc, time.time() - t0))
    return hs, width, height, kw, desc


def cpu_baseline(hs, width, height, kw, budget_s):
    """The oracle (CPU restatement of the reference path, std::thread fan-out like
    CommandLineUI::run) on a bounded, evenly spread pixel sample of the same frame."""
    from raytracer_amd import host
    from oracle import loader
    # every host core this process may run on (OMP_NUM_THREADS, where the box sets one, is its share of the machine)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or avail
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
            "ms_per_step_extrapolated": round(1000.0 * dt * npx / count, 1),
            "cores_available": avail,
            "sample": "%d of %d pixels (every %d-th, %d rays) of the same frame in %.1f s, "
                      "reference KD traversal, %d std::threads (%s)" % (
                          count, npx, stride, rays, dt, cores,
                          "OMP_NUM_THREADS" if os.environ.get("OMP_NUM_THREADS") else "all cores the process may use")}


def csrc_hash():
    """Hash of the device sources: a committed counter pass belongs to the kernels it was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_code_hashes(so_path=None):
    """{mangled kernel name: hash of its gfx950 machine code} of the built library.

    A counter pass belongs to the machine code it was taken on, kernel by kernel:
    a change to one kernel (or to a comment) leaves the bytes of the others as
    they were.  Read straight from the file: the .hip_fatbin section of the
    shared object holds an offload bundle, its gfx950 entry is an ELF whose
    function symbols are the kernels."""
    import hashlib
    import struct

    def sections(b, base=0):
        shoff, = struct.unpack_from("<Q", b, base + 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, base + 0x3A)
        secs = []
        for i in range(shnum):
            name, typ, _flags, addr, off, size, link = struct.unpack_from("<IIQQQQI", b, base + shoff + i * shentsize)
            secs.append({"name_off": name, "type": typ, "addr": addr, "off": off, "size": size, "link": link})
        strtab = secs[shstrndx]
        for s in secs:
            o = base + strtab["off"] + s["name_off"]
            s["name"] = b[o:b.index(b"\0", o)].decode()
        return secs

    b = open(so_path or os.path.join(ROOT, "raytracer_amd", "librt_amd.so"), "rb").read()
    if b[:4] != b"\x7fELF" or b[4] != 2:
        raise ValueError("not a 64-bit ELF")
    fat = next((s for s in sections(b) if s["name"] == ".hip_fatbin"), None)
    if fat is None:
        raise ValueError("no .hip_fatbin section")
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    fb = fat["off"]
    if b[fb:fb + len(magic)] != magic:
        raise ValueError("offload bundle is not in the plain (uncompressed) form")
    n, = struct.unpack_from("<Q", b, fb + len(magic))
    p = fb + len(magic) + 8
    co = None
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", b, p)
        triple = b[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple:
            co = fb + off
    if co is None or b[co:co + 4] != b"\x7fELF":
        raise ValueError("no gfx950 code object in the bundle")
    secs = sections(b, co)
    symtab = next((s for s in secs if s["type"] == 2), None) or next((s for s in secs if s["type"] == 11), None)
    if symtab is None:
        raise ValueError("code object without a symbol table")
    strs = secs[symtab["link"]]
    code, desc = {}, {}
    for i in range(symtab["size"] // 24):
        name, info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", b, co + symtab["off"] + 24 * i)
        if (info & 0xf) not in (1, 2) or size == 0 or shndx == 0 or shndx >= len(secs):
            continue
        o = co + strs["off"] + name
        sym = b[o:b.index(b"\0", o)].decode()
        sec = secs[shndx]
        start = co + sec["off"] + (value - sec["addr"])
        if (info & 0xf) == 2:
            code[sym] = b[start:start + size]
        elif sym.endswith(".kd") and size == 64:
            # the kernel's descriptor (LDS and scratch size, register counts, set-up bits) without bytes
            # 16..23, the distance from the descriptor to the code, which moves when another kernel grows
            desc[sym[:-3]] = b[start:start + 16] + b[start + 24:start + 64]
    return {k: hashlib.sha256(v + desc.get(k, b"")).hexdigest()[:16] for k, v in code.items()}


# the closest-hit kernel of the benchmark scenes: k_trace<FAST = true, STATS = false, TRI = true>
K_TRACE_SYMBOL = "7k_traceILb1ELb0ELb1EE"


def kd_reference_rate(ds, hs, width, height, kw, count=8000):
    """The caller's KD-tree walked with Scene::intersectKdTree's own rules
    (traversal = reference) on a stratified subset of the same frame: Mrays/s of
    that mode, outside the timed region (the timed frames use the library's BVH)."""
    from raytracer_amd import abi, host
    gy = max(1, int(round((count * height / float(width)) ** 0.5)))
    gx = max(1, count // gy)
    xs = (np.arange(gx) * width // gx + width // (2 * gx)).astype(np.int32)
    ys = (np.arange(gy) * height // gy + height // (2 * gy)).astype(np.int32)
    xy = np.stack(np.meshgrid(xs, ys, indexing="xy"), axis=-1).reshape(-1, 2).astype(np.int32)
    p, keep = host.render_params(width, height, traversal=abi.RT_TRAVERSE_REFERENCE, **kw)
    ds.render_pixels(p, xy[:256])                      # pool and kernels warm
    t0 = time.perf_counter()
    _, st = ds.render_pixels(p, xy)
    dt = time.perf_counter() - t0
    return {"Mrays_per_s": round(st.rays_total / dt / 1e6, 2), "rays": int(st.rays_total), "pixels": int(len(xy)),
            "ms": round(dt * 1e3, 1),
            "subset": "%d x %d grid of pixels over the frame, same depth and anti-aliasing" % (gx, gy)}


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
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames a rank renders at a time, each on a scene handle and a host thread of its own "
                         "(0: one per rank up to three -- the thinner a rank's share of the frame, the more of its "
                         "time is the fixed cost of its recursion levels, which another frame's kernels can fill)")
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

    # the roofline needs the duration of every traversal launch of the timed steps
    os.environ.setdefault("RT_AMD_LAUNCH_EVENTS", "1")
    hs, width, height, kw, desc = build_scene(args.config, args.scale)
    trav = abi.RT_TRAVERSE_FAST if args.traversal == "fast" else abi.RT_TRAVERSE_REFERENCE
    t_create = time.perf_counter()
    ds = device.DeviceScene(hs.desc_ptr)
    if rank == 0:
        log("[bench] rt_scene_create (validation, upload, BVH build): %.2f s" % (time.perf_counter() - t_create))
    # Frames in flight on this rank: a scene handle serves one caller at a time, so
    # every frame in flight has a handle (and a host thread) of its own.
    flights = args.frames_in_flight if args.frames_in_flight > 0 else max(1, min(3, world))
    handles = [ds] + [device.DeviceScene(hs.desc_ptr) for _ in range(flights - 1)]
    pad_px = parallel.max_shard_pixels(width, height, TILE, TILE, world)
    # Two packed buffers per handle, used in turn: the gather of a frame may still
    # read its buffer on the backend's stream while the library renders the
    # handle's next frame on its own streams.  An event per buffer says when the
    # gather that read it is done.
    packed = [[torch.zeros((pad_px, 3), dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in handles]
    packed_free = [[None, None] for _ in handles]
    gathered = [torch.empty_like(packed[0][0]) for _ in range(world)] if (rank == 0 and use_dist) else None
    frame = torch.zeros((height, width, 3), dtype=torch.uint8, device="cuda") if rank == 0 else None
    p, keep = host.render_params(width, height, traversal=trav, tile_w=TILE, tile_h=TILE,
                                 shard_index=rank, shard_count=world, out_is_device=True, **kw)
    torch.cuda.synchronize()   # the fills above run on torch's stream, the renders on the library's
    frames_done = [0]
    # per-frame phases on this rank: host wall time of rt_render (it returns when the
    # shard is rendered), then torch events around the gather and the unpack
    phase_events = []

    def render_one(n):
        # frame n of the run: handle n mod F, that handle's buffers in turn
        h, k = n % flights, (n // flights) % 2
        if packed_free[h][k] is not None:
            packed_free[h][k].synchronize()
        t_r = time.perf_counter()
        st = handles[h].render_device(p, packed[h][k].data_ptr())
        return st, h, k, (time.perf_counter() - t_r) * 1e3

    def deliver(h, k, render_ms, timed):
        # gather and unpack of a rendered frame: on this thread, in frame order
        # (every rank issues its collectives in the same order)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
        if ev:
            ev[0].record()
        bufs = parallel.gather_packed(packed[h][k], world, rank, force=use_dist, out=gathered)
        if ev:
            ev[1].record()
        if rank == 0:
            for r, b in enumerate(bufs):
                device.unpack_tiles(b.data_ptr(), 3, width, height, TILE, TILE, r, world,
                                    frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if ev:
            ev[2].record()
            phase_events.append((render_ms, ev))
        if packed_free[h][k] is None:
            packed_free[h][k] = torch.cuda.Event()
        packed_free[h][k].record()

    workers = None
    if flights > 1:
        from concurrent.futures import ThreadPoolExecutor
        workers = ThreadPoolExecutor(max_workers=flights)

    def run(nframes, timed=False):
        """Renders nframes frames, at most `flights` of them at a time; returns their stats."""
        import collections
        stats, pending = [], collections.deque()
        start = frames_done[0]
        frames_done[0] += nframes
        nxt = 0
        while nxt < nframes or pending:
            while workers is not None and nxt < nframes and len(pending) < flights:
                pending.append(workers.submit(render_one, start + nxt))
                nxt += 1
            if workers is None:
                res = render_one(start + nxt)
                nxt += 1
            else:
                res = pending.popleft().result()
            st, h, k, render_ms = res
            deliver(h, k, render_ms, timed)
            stats.append(st)
        return stats

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for h in range(1, flights):   # part of an extra handle's set-up: its first frame sizes its pool and its grids
        handles[h].render_device(p, packed[h][0].data_ptr())
    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    rays = 0
    sums = np.zeros(6)          # ms_trace, trace launches, ms_lights, lights launches, ms_device, batches
    for st in run(args.steps, timed=True):
        rays += st.rays_total
        sums += [st.ms_trace, st.trace_launches, st.ms_lights, st.lights_launches, st.ms_device, st.batches]
    fence()
    elapsed = time.perf_counter() - t0
    if workers is not None:
        workers.shutdown()
    tvals = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    cvals = torch.tensor([float(rays)] + list(sums), dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tvals, op=dist.ReduceOp.MAX)
        dist.all_reduce(cvals, op=dist.ReduceOp.SUM)
    elapsed = float(tvals[0])
    total_rays, trace_ms, trace_n, lights_ms, lights_n, total_dev_ms, total_batches = [float(x) for x in cvals.tolist()]
    # phases per frame on each rank -> mean and max over ranks (tells compute
    # imbalance from gather cost in a multi-GPU line)
    ph = np.array([[r, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])] for r, e in phase_events]).mean(axis=0)
    psum = torch.tensor(ph, dtype=torch.float64, device="cuda")
    pmax = psum.clone()
    if use_dist:
        dist.all_reduce(psum, op=dist.ReduceOp.SUM)
        dist.all_reduce(pmax, op=dist.ReduceOp.MAX)
    phases = {}
    for i, name in enumerate(("render_ms", "gather_ms", "unpack_ms")):
        phases[name] = {"mean_over_ranks": round(float(psum[i]) / world, 3), "max_over_ranks": round(float(pmax[i]), 3)}

    # Untimed pass with device counters: tree nodes visited (V), triangles tested
    # (T) and non-mesh objects tested (O), split between the closest-hit kernel
    # (k_trace: primary, reflection and refraction rays) and the shadow-query
    # kernel (k_shadow).  The roofline entry is k_trace, the kernel with the
    # largest share of the frame, against the FP64 vector roof: algorithmic
    # operations of a ray = 8 V + 90 T + 40 O (a plane step of the tree: 8; the
    # reference's triangle test: 90; the object-space transform of a primitive
    # test: 40), per launch, over the launch duration that HIP events recorded
    # around every launch on the library's own stream during the timed steps.
    # HBM is reported from counters only (a committed rocprofv3 pass of this
    # command, with the commit it was taken at): the tree is shared by
    # neighbouring rays and served by the caches, so algorithmic bytes say nothing
    # about HBM.
    roof = None
    if not args.no_stats_pass:
        ps, keep2 = host.render_params(width, height, traversal=trav, tile_w=TILE, tile_h=TILE,
                                       shard_index=rank, shard_count=world, out_is_device=True,
                                       collect_stats=True, **kw)
        torch.cuda.synchronize()
        sst = ds.render_device(ps, packed[0][0].data_ptr())
        svals = torch.tensor([float(sst.nodes_visited), float(sst.prims_tested), float(sst.objects_tested),
                              float(sst.rays_total), float(sst.shadow_nodes_visited), float(sst.shadow_prims_tested),
                              float(sst.shadow_objects_tested), float(sst.rays_shadow)], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(svals, op=dist.ReduceOp.SUM)
        V, T, O, R, Vs, Ts, Os, Rs = [float(x) for x in svals.tolist()]

        def kernel_line(v, t, o, r, ms, n):
            flops_frame = 8.0 * v + 90.0 * t + 40.0 * o
            per_launch = flops_frame * args.steps / max(n, 1.0)
            ms_launch = ms / max(n, 1.0)
            ach = per_launch / (ms_launch * 1e-3) / 1e12 if ms_launch > 0 else 0.0
            f64_only = (90.0 * t + 40.0 * o) * args.steps / max(n, 1.0) / (ms_launch * 1e-3) / 1e12 if ms_launch > 0 else 0.0
            return {"flop_per_ray": round(flops_frame / max(r, 1.0), 1), "achieved_TFLOPs": round(ach, 4),
                    # the box steps (8 V) run in binary32 on the fast path; triangle and object tests in binary64
                    "binary64_flop_per_ray": round((90.0 * t + 40.0 * o) / max(r, 1.0), 1),
                    "box_step_flop_per_ray": round(8.0 * v / max(r, 1.0), 1),
                    "frac_of_fp64_valu_peak_binary64_terms_only": round(f64_only / FP64_VALU_PEAK_TFLOPS, 4),
                    "frac_of_fp64_valu_peak": round(ach / FP64_VALU_PEAK_TFLOPS, 4),
                    "flop_per_launch": round(per_launch), "ms_per_launch": round(ms_launch, 4),
                    "launches_per_step": round(n / args.steps, 1), "rays_per_step": round(r),
                    "nodes_per_ray": round(v / max(r, 1.0), 2), "tris_per_ray": round(t / max(r, 1.0), 2),
                    "nodes_per_ray_exact": v / max(r, 1.0), "tris_per_ray_exact": t / max(r, 1.0),
                    "objects_per_ray": round(o / max(r, 1.0), 3),
                    "Grays_per_s": round(r * args.steps / max(ms, 1e-9) / 1e6, 3)}

        kt = kernel_line(V - Vs, T - Ts, O - Os, R - Rs, trace_ms, trace_n)
        kl = kernel_line(Vs, Ts, Os, Rs, lights_ms, lights_n)
        # What the same rays cost without the normal cones (subtrees that a ray would only meet from behind are
        # not entered where back faces are culled): the tests a plain closest-hit / occlusion traversal of the
        # same BVH runs, counted on a second scene handle created with RT_AMD_CONES=0.
        # Orientation only -- `frac` above prices the tests that WERE run.
        avoided = None
        if world == 1 and args.traversal == "fast" and os.environ.get("RT_AMD_CONES") is None:
            try:
                os.environ["RT_AMD_CONES"] = "0"
                ds0 = device.DeviceScene(hs.desc_ptr)
                del os.environ["RT_AMD_CONES"]
                s0 = ds0.render_device(ps, packed[0][0].data_ptr())
                ds0.close()
                v0, t0_, o0 = float(s0.nodes_visited - s0.shadow_nodes_visited), float(s0.prims_tested - s0.shadow_prims_tested), \
                    float(s0.objects_tested - s0.shadow_objects_tested)
                r0 = float(s0.rays_total - s0.rays_shadow)
                f0 = (8.0 * v0 + 90.0 * t0_ + 40.0 * o0) / max(r0, 1.0)
                per_launch0 = f0 * (R - Rs) * args.steps / max(trace_n, 1.0)
                ach0 = per_launch0 / (kt["ms_per_launch"] * 1e-3) / 1e12 if kt["ms_per_launch"] > 0 else 0.0
                avoided = {"k_trace_nodes_per_ray_without_culls": round(v0 / max(r0, 1.0), 2),
                           "k_trace_tris_per_ray_without_culls": round(t0_ / max(r0, 1.0), 2),
                           "k_trace_flop_per_ray_without_culls": round(f0, 1),
                           "k_shadow_nodes_per_ray_without_culls": round(float(s0.shadow_nodes_visited) / max(float(s0.rays_shadow), 1.0), 2),
                           "k_trace_frac_of_fp64_valu_peak_at_that_work": round(ach0 / FP64_VALU_PEAK_TFLOPS, 4)}
            except Exception as e:
                os.environ.pop("RT_AMD_CONES", None)
                avoided = {"error": str(e)}
        traffic = None
        hbm = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tpath) and args.scale == 1.0 and world == 1 and args.traversal == "fast":
            try:
                tj = json.load(open(tpath))
                # same sources, or at least the same machine code of the kernel the bytes belong to
                if tj.get("csrc_sha") != csrc_hash():
                    then = {k: v for k, v in (tj.get("kernel_code_sha") or {}).items() if K_TRACE_SYMBOL in k}
                    now = {k: v for k, v in kernel_code_hashes().items() if K_TRACE_SYMBOL in k}
                    if not then or then != now:
                        raise ValueError("counter pass taken on other kernels")
                traffic = tj.get("k_trace_hbm_bytes_per_launch")
                gbs = traffic / (kt["ms_per_launch"] * 1e-3) / 1e9 if traffic and kt["ms_per_launch"] > 0 else None
                hbm = {"bytes_per_launch_from_counters": traffic,
                       "achieved_GBs": round(gbs, 1) if gbs else None, "peak_GBs": HBM_PEAK_GBS,
                       "frac": round(gbs / HBM_PEAK_GBS, 4) if gbs else None,
                       "source": tj.get("source"), "measured_at_commit": tj.get("commit"),
                       "same_kernel_by": "sources" if tj.get("csrc_sha") == csrc_hash() else "machine code of k_trace",
                       "counter_pass": {k: tj.get(k) for k in ("k_trace_valu_busy", "k_trace_lanes_active",
                                                               "k_trace_waves_per_simd")}}
            except Exception as e:
                # no bytes rather than the bytes of other kernels
                traffic, hbm = None, {"stale": str(e), "file": "profiles/traffic_%s.json" % args.config}
        roof = {"bound": "fp64_valu",
                "kernel": "k_trace (%s)" % ("closest hit on the library's 4-wide BVH" if args.traversal == "fast"
                                            else "reference KD traversal"),
                "achieved": kt["achieved_TFLOPs"], "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": kt["frac_of_fp64_valu_peak"], "traffic": traffic,
                "note": "FP64 VALU, contraction off; algorithmic operations 8 V + 90 T + 40 O per ray (the reference's "
                        "binary64 cost of what was tested) from the in-run counter pass, launch durations from HIP "
                        "events of the timed steps.  On the fast path only 90 T + 40 O run in binary64 -- the box "
                        "steps 8 V are binary32 packed multiply-adds -- see frac_of_fp64_valu_peak_binary64_terms_only. "
                        "traffic = HBM bytes per launch from the committed counter pass named under hbm",
                "hbm": hbm, "k_trace": kt, "k_shadow": kl, "tests_avoided": avoided}
        if flights > 1:
            roof["launches_share_the_chip"] = ("%d frames in flight on this rank: a launch's duration includes the time "
                                               "other frames' kernels held the chip's wave slots" % flights)

    if rank == 0:
        if args.out:
            host.write_bmp(args.out, frame.cpu().numpy())
        kd_ref = None
        if world == 1 and args.traversal == "fast" and not args.no_stats_pass:
            try:
                kd_ref = kd_reference_rate(ds, hs, width, height, kw)
            except Exception as e:   # a number for orientation, never a reason to lose the line
                kd_ref = {"error": str(e)}
        cpu = None
        if args.cpu_seconds > 0 and world == 1:   # the CPU leg belongs to the single-GPU line only
            try:
                cpu = cpu_baseline(hs, width, height, kw, args.cpu_seconds)
            except Exception as e:   # the GPU figures are measured by now: never lose the line to the CPU leg
                cpu = {"value": 0.0, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}
        value = total_rays / elapsed / 1e6
        if cpu and cpu["value"] > 0:
            cpu["gpu_over_cpu"] = round(value / cpu["value"], 1)   # a reported ratio, not a quality measure
        out = {
            "metric": "Mrays/s (primary+secondary+shadow)",
            "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc,
                       "traversal": ("fast: 4-wide BVH built by the library at scene creation; the caller's KD-tree is "
                                     "uploaded but not traversed" if args.traversal == "fast"
                                     else "reference: the caller's KD-tree, Scene::intersectKdTree's rules"),
                       "kd_reference": kd_ref,
                       "tile": TILE, "rccl_ranks": world if use_dist else 0, "frames_in_flight": flights,
                       "phases_per_step": phases,
                       "rays_per_step": round(total_rays / args.steps),
                       "sharding": "interleaved %dx%d tiles over %d rank(s), one gather per frame" % (TILE, TILE, world),
                       "trace_ms_per_step": round(trace_ms / args.steps / world, 3),
                       "lights_ms_per_step": round(lights_ms / args.steps / world, 3),
                       "device_ms_per_step": round(total_dev_ms / args.steps / world, 3),
                       "batches_per_step": round(total_batches / args.steps / world, 2)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    for h in handles:
        h.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
