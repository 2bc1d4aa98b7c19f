#!/usr/bin/env python
"""Parse the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into per-kernel HBM bytes per launch.
Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
half the bytes of a wide coalesced streaming read, so fetch is doubled (upper bound for narrower patterns, which
are uncalibrated); WRITE_SIZE is exact for streaming stores and float atomics."""
import collections
import csv
import glob
import json
import sys


def load(dirname, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(dirname + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    return acc


def main():
    out = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "vgg"
    shapes = {"vgg": "config V: N=20646 K=12000 P=2000 R=128 C=512 37x62 (bench.py frames)",
              "fpn": "config F: N=268569 K=4000 P=1000 R=512 C=256, 4 RoIAlign levels of 800x1344 (bench.py --config fpn frames)",
              "fpn_bf16": "config F under bf16 autocast (bench.py --config fpn --amp bf16 frames): adds the fused RPN conv head kernel"}
    fetch = load(out + "/fetch", "FETCH_SIZE")
    write = load(out + "/write", "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if not any(t in k for t in ("kernel",)) or "at::" in k or "Cijk" in k:
            continue
        f = fetch.get(k, [])
        w = write.get(k, [])
        fk = sum(f) / len(f) if f else 0.0
        wk = sum(w) / len(w) if w else 0.0
        res[k] = {"launches": max(len(f), len(w)), "fetch_bytes_raw": round(fk * 1024), "fetch_bytes_x2": round(2 * fk * 1024),
                  "write_bytes": round(wk * 1024), "traffic_bytes": round((2 * fk + wk) * 1024)}
    print(json.dumps({"note": "per launch; traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), gfx950 fetch correction applied",
                      "command": "bench.py %s --steps 4 --warmup 1 (all launches of the process, incl. the 3 initialisation steps)" % ("--config fpn --amp bf16" if cfg == "fpn_bf16" else "--config " + cfg),
                      "shape": shapes.get(cfg, cfg), "kernels": res}, indent=1))


if __name__ == "__main__":
    main()
