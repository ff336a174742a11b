import sys, subprocess, collections, re, glob, os
paths = {}
for root in [os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), d) for d in ("oracle/_ref", "tests/hipemu", "openhevc_amd")] + ["/opt/rocm/lib"]:
    for f in glob.glob(root + "/**/*.so", recursive=True): paths.setdefault(os.path.basename(f), f)
cnt = collections.Counter(); per_lib = collections.defaultdict(set)
rows = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])]
for r in rows:
    m = re.match(r"(.+?)\+0x([0-9a-f]+)\((.*)\)", r[0])
    if m: per_lib[m.group(1)].add(m.group(2))
names = {}
for lib, offs in per_lib.items():
    if lib not in paths: continue
    offs = sorted(offs)
    out = subprocess.run(["addr2line", "-f", "-C", "-e", paths[lib]] + ["0x" + o for o in offs], capture_output=True, text=True).stdout.split("\n")
    for i, o in enumerate(offs): names[(lib, o)] = out[2 * i]
for r in rows:
    m = re.match(r"(.+?)\+0x([0-9a-f]+)\((.*)\)", r[0])
    if not m: cnt["?"] += 1; continue
    lib, off, sym = m.groups()
    cnt[(lib, names.get((lib, off), sym))] += 1
tot = sum(cnt.values())
for k, v in cnt.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 45): print(f"{100*v/tot:5.1f}% {v:6d} {k}")
