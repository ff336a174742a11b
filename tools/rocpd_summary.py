#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite output) into small text files that can be committed under profiles/.

  rocpd_summary.py stats  <results.db>            -> per-kernel calls / total / average (us), like --stats
  rocpd_summary.py pmc    <results.db> [substr]   -> per-kernel mean of every collected counter
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]{0,60}>)?)", name)
    return (m.group(1) if m else name)[:100]


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for n, c, t, a, p in rows:
        print(f"{short(n):100s} {c:6d} {t:12.1f} {a:10.2f} {p:6.2f}")


def pmc(db, substr=""):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    print(f"{'kernel':100s} {'counter':>14s} {'mean_value':>16s} {'n':>4s} {'avg_ns':>10s}")
    for k, cn, v, n, d in rows:
        if substr in k:
            print(f"{short(k):100s} {cn:>14s} {v:16.3f} {n:4d} {d:10.0f}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
