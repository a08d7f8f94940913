#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<dir>/**) into small tracked files
under profiles/: the --stats kernel summary and the per-kernel PMC means
(FETCH_SIZE / WRITE_SIZE, KiB per dispatch).  Also refreshes
profiles/latest_pmc.json which bench.py quotes as `roofline.traffic`.

gfx950 correction (MI355X_MICROARCH.md, section HBM): FETCH_SIZE reports 1/2 of
the bytes of a wide coalesced streaming read, WRITE_SIZE is exact; so
traffic = 2*FETCH_SIZE + WRITE_SIZE.  The doubling was re-calibrated here on the
512 MiB nsol_scale_f32 copy in the same trace (FETCH 256 MiB, WRITE 512 MiB).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def pmc_means(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"),
                  recursive=True)
    acc = collections.defaultdict(list)
    for path in f:
        for r in csv.DictReader(open(path)):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(
                float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True, help="e.g. r01")
    ap.add_argument("--stats", help="rocprofv3 --stats output dir")
    ap.add_argument("--fetch", help="--pmc FETCH_SIZE output dir")
    ap.add_argument("--write", help="--pmc WRITE_SIZE output dir")
    ap.add_argument("--kernel", default="k_pd_fusedk",
                    help="substring of the timed kernel (the one-iteration k_pd_fused of "
                         "bench.py's replay must not be picked: it has more dispatches)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--skip-first", type=int, default=1,
                    help="dispatches of --kernel to drop (p = 0 first launch)")
    ap.add_argument("--config", default="",
                    help="waves:ntx:zchunk the PMC passes were pinned to")
    ap.add_argument("--last", type=int, default=0,
                    help="use only the last N dispatches of --kernel (steady "
                         "state after the online tuner has settled)")
    args = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if args.stats:
        for f in glob.glob(os.path.join(args.stats, "**", "*_kernel_stats.csv"),
                           recursive=True):
            shutil.copy(f, os.path.join(out, "%s_kernel_stats.csv" % args.tag))
    summary = {"kernel": args.kernel, "size": args.size, "tag": args.tag}
    if args.config:
        summary["kernel_config"] = [int(t) for t in args.config.split(":")]
    if args.stats:
        # steady state of the dominant instantiation: the first-call autotuner
        # launches other footprint shapes of the same kernel, so the average of
        # ALL dispatches (what --stats reports) is slightly pessimistic
        durs = collections.defaultdict(list)
        for f in glob.glob(os.path.join(args.stats, "**", "*_kernel_trace.csv"),
                           recursive=True):
            for r in csv.DictReader(open(f)):
                if args.kernel in r["Kernel_Name"]:
                    durs[r["Kernel_Name"]].append(
                        (int(r["Start_Timestamp"]),
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        if durs:
            name = max(durs, key=lambda k: len(durs[k]))
            d = [x[1] for x in sorted(durs[name])]
            tail = d[-args.last:] if args.last > 0 else d[len(d) // 2:]
            summary["dominant_instantiation"] = name[:80]
            summary["dispatches"] = len(d)
            summary["avg_ns_all"] = sum(d) / len(d)
            summary["avg_ns_steady"] = sum(tail) / len(tail)
            summary["steady_dispatches"] = len(tail)
    rows = []
    for name, d in (("FETCH_SIZE", args.fetch), ("WRITE_SIZE", args.write)):
        if not d:
            continue
        acc = pmc_means(d)
        match = [k for k in acc if args.kernel in k[0] and k[1] == name]
        dominant = max(match, key=lambda k: len(acc[k]))[0] if match else None
        for (kern, ctr), vals in sorted(acc.items()):
            use = vals
            if kern == dominant and args.last > 0:
                use = vals[-args.last:]
            elif kern == dominant and len(vals) > args.skip_first:
                use = vals[args.skip_first:]
            mean = sum(use) / len(use)
            rows.append((kern[:100], ctr, len(use), mean))
            if kern == dominant and ctr == name:
                summary[name.lower() + "_kib"] = mean
                summary.setdefault("dominant_instantiation", kern[:80])
    with open(os.path.join(out, "%s_pmc_summary.csv" % args.tag), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_KiB_per_dispatch"])
        w.writerows(rows)
    if "fetch_size_kib" in summary and "write_size_kib" in summary:
        summary["traffic_bytes_per_launch"] = 1024.0 * (
            2.0 * summary["fetch_size_kib"] + summary["write_size_kib"])
        summary["note"] = ("traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950: "
                           "FETCH_SIZE counts half of a wide coalesced read)")
        with open(os.path.join(out, "latest_pmc.json"), "w") as f:
            json.dump(summary, f, indent=1)
        # bench.py looks the traffic up by the configuration its tuner chose
        table_path = os.path.join(out, "pmc_by_config.json")
        try:
            table = json.load(open(table_path))
        except Exception:
            table = {}
        # keyed by the template name of the dominant instantiation
        # ("k_pd_fusedk"), not by the --kernel substring it was found with:
        # bench.py looks the entry up by the kernel it timed
        base = re.search(r"(k_[A-Za-z0-9_]+)",
                         summary.get("dominant_instantiation", "") or
                         (dominant or ""))
        key = "%s:%d:%s" % (base.group(1) if base else args.kernel, args.size,
                            ":".join(str(v) for v in
                                     summary.get("kernel_config", [])))
        table[key] = {"traffic_bytes_per_launch":
                      summary["traffic_bytes_per_launch"], "tag": args.tag}
        with open(table_path, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
