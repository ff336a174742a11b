#!/usr/bin/env python3
"""Per-configuration HBM traffic of a tools/bench_kernels.py run from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) and one
--kernel-trace pass: consecutive dispatches of one kernel with one grid size are one configuration (bench_kernels launches each 14 times).

    pmc_per_kernel.py <trace_results.db> <fetch_results.db> <write_results.db>

Prints one JSON line per configuration, in launch order: kernel, grid, dispatches, mean duration (us), HBM read / written bytes per launch
(read side = FETCH_SIZE * 1024 * 2: the gfx950 half-count correction of MI355X_MICROARCH.md; write side = WRITE_SIZE * 1024)."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]{0,60}>)?)", name)
    return (m.group(1) if m else name)[:80]


def groups(rows):
    out, cur = [], None
    for name, grid, val in rows:
        key = (short(name), grid)
        if cur is None or cur[0] != key:
            cur = [key, []]
            out.append(cur)
        cur[1].append(val)
    return out


def main():
    trace, fetch, write = sys.argv[1:4]
    t = sqlite3.connect(trace).execute("select name, grid_x, duration from kernels order by start").fetchall()
    gt = [g for g in groups(t) if "ohevc" in g[0][0]]
    def pmc(db, counter):
        rows = sqlite3.connect(db).execute("select kernel_name, grid_size, value from counters_collection where counter_name = ? order by dispatch_id", (counter,)).fetchall()
        return [g for g in groups(rows) if "ohevc" in g[0][0]]
    gf, gw = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
    for i, g in enumerate(gt):
        row = {"kernel": g[0][0], "grid": g[0][1], "dispatches": len(g[1]), "mean_us": round(sum(g[1]) / len(g[1]) / 1e3, 2)}
        if i < len(gf) and gf[i][0][0] == g[0][0]:
            row["hbm_read_bytes"] = round(sum(gf[i][1]) / len(gf[i][1]) * 1024 * 2)
        if i < len(gw) and gw[i][0][0] == g[0][0]:
            row["hbm_written_bytes"] = round(sum(gw[i][1]) / len(gw[i][1]) * 1024)
        print(json.dumps(row))


if __name__ == "__main__":
    main()
