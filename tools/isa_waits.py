"""Lists, per kernel of an assembly file (hipcc -S --cuda-device-only), the
vector-memory loads and the waits between them in program order: a run of
`load wait load wait` is a chain of memory latencies that may be avoidable.
usage: tools/isa_waits.py file.s kernel-name-substring"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l))
out, n = [], 0
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"):
        break
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            out.append("\n" + t.split(":")[0])
        continue
    n += 1
    op = t.split()[0]
    if op.startswith(("global_load", "scratch_load", "buffer_load", "flat_load")):
        out.append("%d:L%s" % (n, op.split("_")[-1].replace("dword", "") or "1"))
    elif op.startswith(("global_store", "scratch_store")):
        out.append("%d:S" % n)
    elif op.startswith("global_atomic"):
        out.append("%d:A" % n)
    elif op == "s_waitcnt" and "vmcnt" in t:
        out.append("%d:w%s" % (n, re.search(r"vmcnt\((\d+)\)", t).group(1)))
print(" ".join(out))
