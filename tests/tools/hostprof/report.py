"""the table of a sampler.out (tests/tools/hostprof/sampler.c)   (python report.py <roast_prof binary> <sampler.out> [rows 40])"""
import collections, subprocess, sys
exe, path = sys.argv[1], sys.argv[2]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
samples = [l.split() for l in open(path)]
addrs = sorted({f[1:] for s in samples for f in s if f[0] == "b"})
out = subprocess.run(["addr2line", "-f", "-e", exe] + ["0x" + a for a in addrs], capture_output=True, text=True).stdout.split("\n")
name = {a: out[2 * i] for i, a in enumerate(addrs)}
line = {a: out[2 * i + 1].split("/")[-1].split(" ")[0] for i, a in enumerate(addrs)}
lines = collections.Counter()
own, lib_top, incl, pair = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
for s in samples:
    if not s:
        continue
    top = s[0]
    first = next((f for f in s if f[0] == "b"), None)
    fn = name[first[1:]] if first else "(no program frame)"
    own[fn] += 1
    if first:
        lines[(fn, line[first[1:]], top[1:] if top[0] == "l" else "")] += 1
    if top[0] == "l":
        lib_top[top[1:]] += 1
        pair[(fn, top[1:])] += 1
    for g in {name[f[1:]] for f in s if f[0] == "b"}:
        incl[g] += 1
n = len(samples)
print(f"{n} samples")
print("  self+libs  inclusive  function")
for fn, c in own.most_common(rows):
    libs = ", ".join(f"{l} {k}" for (g, l), k in sorted(pair.items(), key=lambda x: -x[1]) if g == fn and k >= max(3, c // 20))
    print(f"  {c:8d}  {incl[fn]:9d}  {fn}   [{libs}]")
print("  library functions on top:")
for l, c in lib_top.most_common(20):
    print(f"  {c:8d}  {l}")
print("  hottest lines (first program frame; the library function on top, if any):")
for (fn, ln, lib), c in lines.most_common(rows):
    print(f"  {c:8d}  {fn}  {ln}  {lib}")
