#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc output directory: per kernel of interest, every counter's value per
dispatch (in dispatch order) and the dispatch duration.  Usage: pmc_summary.py <dir> [kernel-substring ...]"""
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    wanted = tuple(sys.argv[2:]) or ("k_search", "k_order", "k_locate")
    out = {}
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(w in name for w in wanted):
                continue
            k = name.split("(")[0].replace("void ", "")
            e = out.setdefault(k, {}).setdefault(int(r["Dispatch_Id"]), {})
            e["dispatch_id"] = int(r["Dispatch_Id"])
            e[r["Counter_Name"]] = float(r["Counter_Value"])
            e["duration_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            e["vgpr"] = int(r["VGPR_Count"])
            e["lds"] = int(r["LDS_Block_Size"])
    print(json.dumps({k: [v[i] for i in sorted(v)] for k, v in out.items()}))


if __name__ == "__main__":
    main()
