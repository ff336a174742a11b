#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch of the dominant kernel.

Units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are
reported in KiB-like units of 1024 B... (hbm_bytes = counter * 1024) and on gfx950 FETCH_SIZE reads exactly 1/2 of the
bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is uncalibrated and taken as is.
Writes <out>/pmc_traffic.json; copy it to profiles/pmc_traffic.json for bench.py to report as roofline.traffic.
"""
import csv
import glob
import json
import os
import sys


def per_dispatch(dirpath, counter, kernel_substr):
    vals = []
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
    return vals


def main():
    out = sys.argv[1]
    kern = sys.argv[2] if len(sys.argv) > 2 else "tu_idct_add_kernel"
    fetch = per_dispatch(os.path.join(out, "prof_pmc_fetch"), "FETCH_SIZE", kern)
    write = per_dispatch(os.path.join(out, "prof_pmc_write"), "WRITE_SIZE", kern)
    res = {"kernel": kern, "fetch_size_raw_mean": sum(fetch) / len(fetch) if fetch else None,
           "write_size_raw_mean": sum(write) / len(write) if write else None, "dispatches": [len(fetch), len(write)]}
    if fetch and write:
        rd = res["fetch_size_raw_mean"] * 1024 * 2      # gfx950: FETCH_SIZE = 1/2 of wide coalesced read bytes
        wr = res["write_size_raw_mean"] * 1024
        res.update({"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; WRITE_SIZE uncalibrated"})
    json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
