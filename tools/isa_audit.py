#!/usr/bin/env python3
"""Static audit of the shipped gfx950 kernels for the pattern that cost this round the most: memory loads that wait for each other although
nothing makes them dependent (loads inside the branches of a rare case, behind bounds checks, behind each other's address arithmetic).

For every kernel of openhevc_amd/csrc/*.hip (compiled here with hipcc -S, no GPU needed):
  vgpr / sgpr / scratch / lds        the resource line of the kernel descriptor
  loads                              global / flat / scalar load instructions
  load rounds                        how many times the straight-line listing goes  load ... wait(vmcnt) ... load  (a lower bound of the
                                     dependent memory round trips of one wavefront; branches make the real number path dependent)
  scalar rounds                      the same for s_load ... s_waitcnt lgkmcnt
  spills                             v_writelane / v_readlane pairs (scalar registers parked in vector lanes), scratch_ instructions

    python tools/isa_audit.py [> profiles/<name>.txt]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openhevc_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out)) if len(out) == len(names) else {n: n for n in names}
    except OSError:
        return {n: n for n in names}


def audit(path):
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        r = subprocess.run([HIPCC, "-S", "--cuda-device-only", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                            "-I" + CSRC, path, "-o", asm], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-2000:])
        text = open(asm).read()
    desc = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", body).group(1)) if re.search(r"\.amdhsa_" + k + r" (\d+)", body) else 0
        desc[m.group(1)] = dict(vgpr=g("next_free_vgpr"), sgpr=g("next_free_sgpr"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    rows = []
    for name, dsc in desc.items():
        a = text.find("\n" + name + ":")
        b = text.find(".amdhsa_kernel " + name, a)
        body = text[a:b] if a >= 0 and b > a else ""
        lines = [l.strip() for l in body.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        ins = [l.split()[0] for l in lines]
        vloads = sum(1 for i in ins if i.startswith(("global_load", "flat_load", "buffer_load")))
        sloads = sum(1 for i in ins if i.startswith("s_load"))
        stores = sum(1 for i in ins if i.startswith(("global_store", "flat_store", "buffer_store")))
        vrounds = srounds = 0
        pend_v = pend_s = False
        for l in lines:
            op = l.split()[0]
            if op.startswith(("global_load", "flat_load", "buffer_load")):
                pend_v = True
            elif op.startswith("s_load"):
                pend_s = True
            elif op == "s_waitcnt":
                if "vmcnt" in l and pend_v:
                    vrounds += 1; pend_v = False
                if "lgkmcnt" in l and pend_s:
                    srounds += 1; pend_s = False
        rows.append(dict(name=name, n=len(ins), vloads=vloads, sloads=sloads, stores=stores, vrounds=vrounds, srounds=srounds,
                         lane_spills=sum(1 for i in ins if i in ("v_writelane_b32", "v_readlane_b32")),
                         scratch_ins=sum(1 for i in ins if i.startswith("scratch_")), **dsc))
    return rows


def main():
    files = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    allrows = []
    for f in files:
        try:
            rows = audit(os.path.join(CSRC, f))
        except RuntimeError as e:
            print(f"{f}: did not compile: {e}", file=sys.stderr)
            continue
        for r in rows:
            r["file"] = f
        allrows += rows
    dm = demangle([r["name"] for r in allrows])
    print("shipped gfx950 kernels (hipcc -S of openhevc_amd/csrc/*.hip at HEAD); rounds = load ... wait ... load sequences in the listing")
    print(f"{'kernel':74s} {'instr':>6s} {'vgpr':>4s} {'sgpr':>4s} {'lds':>6s} {'scr':>4s} {'vld':>4s} {'vrnd':>4s} {'sld':>4s} {'srnd':>4s} {'st':>3s} {'lanesp':>6s}")
    for r in sorted(allrows, key=lambda r: (r["file"], r["name"])):
        short = re.sub(r"\(.*", "", dm[r["name"]]).replace("void ohevc::", "").replace("ohevc::", "")
        print(f"{short[:74]:74s} {r['n']:6d} {r['vgpr']:4d} {r['sgpr']:4d} {r['lds']:6d} {r['scratch']:4d} {r['vloads']:4d} {r['vrounds']:4d} {r['sloads']:4d} {r['srounds']:4d} "
              f"{r['stores']:3d} {r['lane_spills']:6d}")


if __name__ == "__main__":
    main()
