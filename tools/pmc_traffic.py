#!/usr/bin/env python
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950) into a
per-kernel HBM traffic summary, with the corrections MI355X_MICROARCH.md section 'HBM' prescribes:
  * both counters are in KiB (x1024);
  * FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) on gfx950, so the
    read side is doubled for the kernels that stream the cost volume as 16-byte quads (stm_k_agg_*, calibrated on
    stm_k_agg_h: corrected read = 534.7 MB vs 534.9 MB algorithmic); WRITE_SIZE is exact for 16 B/lane stores.
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [commit]"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("stm::", "")
    if "stm_k_pq_hc" in n:
        return "pq_h"                             # round 3: the cost-computing streaming pass
    if "stm_k_pq_hsr" in n:
        return "pq_hw"                            # round 4: the last pass + WTA with the window range in registers
    if "stm_k_pq_hs" in n or "stm_k_pq_h<" in n:
        args = n[n.index("<") + 1:n.index(">")].split(", ")
        return "pq_hw" if len(args) >= 2 and args[1] == "true" else "pq_h"   # <NW, WTA, ...>
    if "vwin_table" in n:
        return "pq_vtab"
    for k in ("pq_v12", "pq_cost"):
        if k in n:
            return k
    if "agg_h" in n:
        args = n[n.index("<") + 1:n.index(">")].split(", ") if "<" in n else []
        if len(args) >= 2 and args[1] == "true":
            return "agg_hw"                       # <QUAD, WTA = true, ...>
        if len(args) >= 5 and args[4] == "true":
            return "agg_h_cost"                   # <..., COST = true>: costs computed on the fly, no volume read
        return "agg_h"
    for k in ("agg_v", "cost_init", "cross_arms", "irv_vote", "bilateral", "gaussian_max", "view_synth", "synth_mux", "demux"):
        if k in n:
            return k
    if "mux" in n:
        return "mux"
    return n


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    fused_cost = not any("pq_cost" in k for k in set(fetch) | set(write))  # the first pass then reads dword planes, not a volume
    for k in sorted(set(fetch) | set(write)):
        if "stm_k_" not in k:
            continue
        f_raw = fetch.get(k, 0.0) * 1024.0
        w = write.get(k, 0.0) * 1024.0
        # the x2 applies to kernels that stream a quad volume with 16 B/lane loads; the on-the-fly cost pass reads only
        # dword planes (pixels, census, arms)
        # matrix-pipe kernels: pq_h / pq_hw / pq_v12 stream the volume as 16-byte elements; pq_cost reads dword planes
        corr = 2.0 if (("agg_" in k and short(k) != "agg_h_cost") or "cost_init" in k or short(k) in ("pq_hw", "pq_v12") or (short(k) == "pq_h" and not fused_cost)) else 1.0
        name = short(k)
        if name in out:  # two kernels behind one short name (e.g. the per-stage path's passes): keep both apart
            name = k.split("(")[0].replace("void ", "").replace("stm::", "")
        out[name] = {"kernel": k.split("(")[0], "fetch_raw_bytes": f_raw, "fetch_correction": corr,
                     "write_bytes": w, "traffic_bytes": f_raw * corr + w}
    if len(sys.argv) > 4:
        out["commit"] = {"commit": sys.argv[4], "note": "git commit of the build these counters were taken on"}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        if k == "commit":
            continue
        print("%-14s read %8.1f MB (raw %8.1f) write %8.1f MB total %8.1f MB" % (k, v["fetch_raw_bytes"] * v["fetch_correction"] / 1e6,
                                                                            v["fetch_raw_bytes"] / 1e6, v["write_bytes"] / 1e6, v["traffic_bytes"] / 1e6))


if __name__ == "__main__":
    main()
