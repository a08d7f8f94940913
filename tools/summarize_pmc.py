#!/usr/bin/env python3
"""Mean counter values per kernel from rocprofv3 --pmc output directories:
    python3 tools/summarize_pmc.py <kernel substring> <dir> [<dir> ...]
Prints one JSON line per (kernel, counter)."""
import collections, csv, glob, json, os, sys

pat = sys.argv[1]


def short(name):
    """k_lsmr_u<float, 4, 4, false> from 'void (anonymous namespace)::k_lsmr_u<...>(...)'."""
    name = name.replace("(anonymous namespace)::", "").replace("nsol_blur3::", "")
    if name.startswith("void "):
        name = name[5:]
    depth, out = 0, []
    for ch in name:                     # cut at the argument list, keep <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[:80]


acc = collections.defaultdict(list)
for d in sys.argv[2:]:
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"),
                          recursive=True):
        for r in csv.DictReader(open(path)):
            if pat in r["Kernel_Name"]:
                acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(
                    float(r["Counter_Value"]))
    for path in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if pat in r["Name"]:
                print(json.dumps({"kernel": short(r["Name"]),
                                  "calls": int(r["Calls"]),
                                  "avg_ns": float(r["AverageNs"]),
                                  "min_ns": float(r["MinNs"])}))
for (k, c), v in sorted(acc.items()):
    print(json.dumps({"kernel": k, "counter": c, "dispatches": len(v),
                      "mean": sum(v) / len(v)}))
