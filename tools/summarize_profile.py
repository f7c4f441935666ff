#!/usr/bin/env python3
"""Condenses gpurun_out/profile_<tag>/ (written by tools/profile_round.sh) into
profiles/<tag>_*: the kernel-stats table as it came from rocprofv3, a
per-kernel counter summary, and profiles/traffic_C4.json, which bench.py reads
for roofline.traffic.

HBM bytes per launch follow the recipe for this chip: FETCH_SIZE and WRITE_SIZE
are collected in separate passes, both are in KiB, and FETCH_SIZE counts a
128-byte request as 64 bytes on gfx950, so read bytes = 2 * FETCH_SIZE * 1024
(an upper bound for the narrow gathers of a traversal kernel, exact for wide
streaming reads); written bytes = WRITE_SIZE * 1024.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[a-z, ]+>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:48]


def counters(path):
    """kernel -> counter -> (sum over launches, launches)"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "profile_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
    for name in ("frame_split_one_stream.md", "bench_one_stream.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, tag + "_" + name))

    # rocprofv3's own stats average every launch of the process, warm-up frame and
    # counter pass included; the timed steps are the last launches of the
    # <true, false> instantiations before the <true, true> (counter pass) ones.
    bench = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tl = ["# %s: launch durations from the rocprofv3 kernel trace of `python3 bench.py` against bench.py's HIP events" % tag, "",
          "| kernel | launches (all) | avg ms (all, = rocprofv3 --stats) | timed launches | avg ms (timed launches) | bench.py HIP-event avg ms |",
          "|---|---|---|---|---|---|"]
    for key, jk in (("k_trace<true, false, true>", "k_trace"), ("k_shadow<true, false, true>", "k_shadow")):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if short(r["Kernel_Name"]) == key]
        n = int(round(bench["roofline"][jk]["launches_per_step"] * bench["steps"]))
        timed = d[-n:]
        tl.append("| %s | %d | %.4f | %d | %.4f | %.4f |" % (key, len(d), sum(d) / len(d), n, sum(timed) / n,
                                                          bench["roofline"][jk]["ms_per_launch"]))
    open(os.path.join(dst, tag + "_launch_durations.md"), "w").write("\n".join(tl) + "\n")
    print("\n".join(tl))

    fetch, nf = counters(os.path.join(src, "pmc_fetch", "c_counter_collection.csv"))
    write, nw = counters(os.path.join(src, "pmc_write", "c_counter_collection.csv"))
    sq, ns = counters(os.path.join(src, "pmc_sq", "c_counter_collection.csv"))
    lines = ["# %s: counters per kernel of `python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-stats-pass`" % tag,
             "# (two frames of C4; one rocprofv3 --pmc pass per counter group)", "",
             "| kernel | launches | HBM read MB/launch (2 x FETCH_SIZE) | HBM write MB/launch | waves/SIMD | VALU busy | lanes active | wave time: waiting / issue stall / issuing |",
             "|---|---|---|---|---|---|---|---|"]
    traffic = {}
    for k in sorted(sq, key=lambda k: -sq[k].get("SQ_WAVE_CYCLES", 0)):
        if not k.startswith("k_"):
            continue
        v = sq[k]
        wc = max(v["SQ_WAVE_CYCLES"], 1.0)
        simd_quads = v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / 4.0      # SIMD x quad-cycles the kernel was resident
        rd = 2.0 * fetch[k].get("FETCH_SIZE", 0.0) * 1024.0 / max(nf.get(k, 1), 1)
        wr = write[k].get("WRITE_SIZE", 0.0) * 1024.0 / max(nw.get(k, 1), 1)
        traffic[k] = {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "launches": ns[k],
                      "valu_busy": round(v["SQ_ACTIVE_INST_VALU"] / simd_quads, 3),
                      "lanes_active": round(v["SQ_THREAD_CYCLES_VALU"] / max(v["SQ_ACTIVE_INST_VALU"], 1.0), 1),
                      "waves_per_simd": round(wc / simd_quads, 2)}
        lines.append("| %s | %d | %.1f | %.1f | %.2f | %.0f %% | %.1f / 64 | %.0f %% / %.0f %% / %.0f %% |" % (
            k, ns[k], rd / 1e6, wr / 1e6, wc / simd_quads, 100.0 * v["SQ_ACTIVE_INST_VALU"] / simd_quads,
            v["SQ_THREAD_CYCLES_VALU"] / max(v["SQ_ACTIVE_INST_VALU"], 1.0),
            100.0 * v["SQ_WAIT_ANY"] / wc, 100.0 * v["SQ_WAIT_INST_ANY"] / wc, 100.0 * v["SQ_ACTIVE_INST_ANY"] / wc))
    open(os.path.join(dst, tag + "_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
    kt = next((traffic[k] for k in traffic if k.startswith("k_trace")), None)
    import subprocess
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "raytracer_amd/csrc", "bench.py"],
                               capture_output=True, text=True).stdout.strip()
        if dirty:
            commit += "+uncommitted"
    except Exception:
        commit = None
    sys.path.insert(0, ROOT)
    import bench
    out = {"source": "profiles/%s_pmc_summary.md" % tag, "commit": commit, "csrc_sha": bench.csrc_hash(), "kernels": traffic}
    try:   # per kernel, so that a later change to another kernel does not orphan these counters
        out["kernel_code_sha"] = bench.kernel_code_hashes()
    except Exception as e:
        out["kernel_code_sha"] = None
        print("no machine-code hashes: %s" % e, file=sys.stderr)
    if kt:
        out["k_trace_waves_per_simd"] = kt["waves_per_simd"]
        out["k_trace_hbm_bytes_per_launch"] = kt["read_bytes_per_launch"] + kt["write_bytes_per_launch"]
        out["k_trace_valu_busy"] = kt["valu_busy"]
        out["k_trace_lanes_active"] = kt["lanes_active"]
    json.dump(out, open(os.path.join(dst, "traffic_C4.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
