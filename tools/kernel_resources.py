#!/usr/bin/env python3
"""Registers, LDS, scratch and code size of every kernel of the built library, from the metadata of its gfx950
code object (no GPU needed): what decides how many wavefronts of a kernel a SIMD holds.
usage: kernel_resources.py [librt_amd.so] > profiles/<tag>_kernel_resources.md"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
sys.path.insert(0, ROOT)


def short(name):
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    rest = name[m.end():]
    base, rest = rest[:n], rest[n:]
    t = re.match(r"I((?:Lb[01]E)+)E", rest)
    if t:
        base += "<" + ", ".join("true" if b == "1" else "false" for b in re.findall(r"Lb([01])E", t.group(1))) + ">"
    return base


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "raytracer_amd", "librt_amd.so")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
        subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True)
        syms = subprocess.check_output([LLVM + "llvm-readelf", "-sW", co], text=True)
    size = {}
    for ln in syms.splitlines():
        p = ln.split()
        if len(p) >= 8 and p[3] == "FUNC":
            size[p[7]] = int(p[2])
    import bench
    sha = bench.kernel_code_hashes(so)
    rows = []
    for block in notes.split("  - .agpr_count:")[1:]:
        def field(k):
            m = re.search(r"\.%s:\s+(\S+)" % k, block)
            return m.group(1) if m else "?"
        name = field("name")
        vg, sg = int(field("vgpr_count")), int(field("sgpr_count"))
        # a SIMD has 512 vector registers per lane, handed out in blocks of 8; at most 8 wavefronts
        waves = min(8, 512 // max(8, (vg + 7) // 8 * 8))
        rows.append((short(name), vg, sg, int(field("private_segment_fixed_size")), int(field("group_segment_fixed_size")),
                     int(field("vgpr_spill_count")), waves, size.get(name, 0), sha.get(name, "?")))
    print("# kernels of %s: resources from the code object's metadata" % os.path.relpath(so, ROOT))
    print("# (wavefronts per SIMD by vector registers alone; dynamic LDS -- the traversal stacks -- comes on top of the static figure)")
    print()
    print("| kernel | vector registers | scalar registers | scratch B / lane | static LDS B | spilled registers | wavefronts / SIMD by registers | code bytes | machine-code hash |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in sorted(rows):
        print("| %s | %d | %d | %d | %d | %d | %d | %d | %s |" % r)


if __name__ == "__main__":
    main()
