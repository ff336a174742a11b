#!/usr/bin/env python3
"""HBM bytes per launch of the dominant kernel from two rocprofv3 PMC passes (rocpd sqlite output).

    pmc_traffic.py <fetch_results.db> <write_results.db> [kernel substring] > profiles/pmc_traffic.json

Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB
(their expressions end in /1024); on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane)
coalesced streaming read, so the read side is doubled.  WRITE_SIZE was calibrated in the same run on a kernel with
a known byte count (torch's 1 GiB random fill reads 1048750 KiB) and is used as is.
"""
import json
import sqlite3
import sys


def mean_counter(db, counter, substr):
    con = sqlite3.connect(db)
    r = con.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like ?",
                    (counter, f"%{substr}%")).fetchone()
    return r


def main():
    fetch_db, write_db = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else "tu_idct_add_kernel<5, unsigned char"
    f, nf = mean_counter(fetch_db, "FETCH_SIZE", kern)
    w, nw = mean_counter(write_db, "WRITE_SIZE", kern)
    res = {"kernel": kern, "fetch_size_kib_mean": f, "write_size_kib_mean": w, "dispatches": [nf, nw]}
    if f and w:
        rd, wr = f * 1024 * 2, w * 1024
        res.update({"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                    "note": "read side = FETCH_SIZE*1024*2 (gfx950 half-count correction), write side = WRITE_SIZE*1024"})
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
