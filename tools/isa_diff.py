#!/usr/bin/env python3
"""Per-kernel comparison of the gfx950 machine code inside two builds of libohevc_hip.so - a check that needs no GPU.

Use: device-code commits made after a round's GPU budget is spent cannot be re-run on the device; this shows which kernels'
instructions differ from the last build that was (kernels whose text is byte-identical behave identically).

    git worktree add /tmp/wt <commit> && make -C /tmp/wt/openhevc_amd/csrc
    tools/isa_diff.py /tmp/wt/openhevc_amd/libohevc_hip.so openhevc_amd/libohevc_hip.so > profiles/<name>.txt

Every offload bundle is extracted (llvm-objdump --offloading), disassembled, and split at function symbols; branch targets and
other absolute addresses are normalised so that a kernel that merely moved compares equal.  `--map old=new` renames symbol
substrings before matching (a template parameter added to a kernel changes its mangled name, not its code)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(so):
    out = {}
    with tempfile.TemporaryDirectory() as d:
        lib = os.path.join(d, "lib.so")
        with open(so, "rb") as f, open(lib, "wb") as g:
            g.write(f.read())
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], cwd=d, check=True, stdout=subprocess.DEVNULL)
        for name in sorted(os.listdir(d)):
            if "amdgcn" not in name:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", "-C", os.path.join(d, name)],
                                 check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    out[cur] = []
                    continue
                if cur is None or not line.strip():
                    continue
                line = re.sub(r"//.*$", "", line).strip()
                line = re.sub(r"<[^>]*\+0x[0-9a-f]+>", "<L>", line)          # branch targets: symbol + offset
                line = re.sub(r"\b(s_c?branch\w*|s_call\w*)\s+\S+", r"\1 T", line)
                if line:
                    out[cur].append(line)
    return out


def digest(lines):
    return hashlib.sha1("\n".join(lines).encode()).hexdigest()[:12]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    maps = [a[6:].split("=", 1) for a in sys.argv[1:] if a.startswith("--map=")]
    old, new = kernels_of(args[0]), kernels_of(args[1])
    for a, b in maps:
        old = {k.replace(a, b): v for k, v in old.items()}
    same = changed = 0
    rows = []
    for k in sorted(set(old) | set(new)):
        if k not in old:
            rows.append(("only in new", k, len(new[k])))
        elif k not in new:
            rows.append(("only in old", k, len(old[k])))
        elif digest(old[k]) == digest(new[k]):
            same += 1
        else:
            changed += 1
            rows.append((f"differs ({len(old[k])} -> {len(new[k])} instructions)", k, 0))
    print(f"old: {args[0]}: {len(old)} device functions; new: {args[1]}: {len(new)}")
    print(f"identical instruction streams: {same}; differing: {changed}")
    for what, k, n in rows:
        print(f"  {what}: {k}" + (f" ({n} instructions)" if n else ""))


if __name__ == "__main__":
    main()
