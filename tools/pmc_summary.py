"""Summarises rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, --output-format csv) into HBM bytes per launch per kernel.

Corrections for gfx950: both counters are in KiB; FETCH_SIZE reads half the bytes of wide (16 B per lane) coalesced
streams, so it is doubled.  Usage: pmc_summary.py <fetch.csv> <write.csv> <workload> <out.json>
"""
import collections
import csv
import hashlib
import json
import os
import sys


def source_id():
    """Same hash as bench.py's source_id(): the sources the device library is built from."""
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer-rs_amd", "csrc")
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def load(path, counter):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        name = "k_" + k.split("k_")[1].split("<")[0].split("(")[0] if "k_" in k else k[:40]
        tot[name] += float(r["Counter_Value"])
        n[name] += 1
    return tot, n


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
out = {"workload": sys.argv[3], "source_id": source_id(), "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs; KiB * 1024; FETCH_SIZE doubled (gfx950)",
       "bytes_per_launch": {}, "detail": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f = 2.0 * 1024.0 * fetch.get(k, 0.0) / max(1, nf.get(k, 1))
    w = 1024.0 * write.get(k, 0.0) / max(1, nw.get(k, 1))
    out["bytes_per_launch"][k] = round(f + w)
    out["detail"][k] = {"launches": nf.get(k, 0), "fetch_bytes_per_launch_corrected": round(f), "write_bytes_per_launch": round(w)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
