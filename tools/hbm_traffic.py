"""Turns two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) into HBM bytes per
launch, as MI355X_MICROARCH.md prescribes: KiB -> bytes, FETCH_SIZE doubled (gfx950 counts 64 B
per 128-B read request), WRITE_SIZE as is.
    python tools/hbm_traffic.py <fetch.csv> <write.csv> <kernel substring>[,<substring>...] <launches>"""
import csv
import json
import sys


def total(path, names, counter):
    s, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in names):
            s += float(r["Counter_Value"])
            n += 1
    return s, n


fetch_csv, write_csv, names, launches = sys.argv[1], sys.argv[2], sys.argv[3].split(","), int(sys.argv[4])
f, nf = total(fetch_csv, names, "FETCH_SIZE")
w, nw = total(write_csv, names, "WRITE_SIZE")
out = {
    "FETCH_SIZE_KiB_raw": f / launches, "WRITE_SIZE_KiB_raw": w / launches,
    "launches_sampled": launches, "dispatches_summed": [nf, nw],
    "fetch_bytes_corrected": 2.0 * 1024.0 * f / launches, "write_bytes": 1024.0 * w / launches,
}
out["hbm_bytes_per_launch"] = out["fetch_bytes_corrected"] + out["write_bytes"]
print(json.dumps(out, indent=1))
