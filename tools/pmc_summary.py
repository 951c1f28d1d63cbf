#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc output directory: per kernel of interest, every counter's value per
dispatch and the dispatch duration.  Usage: pmc_summary.py <dir> [kernel-prefix ...]"""
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    prefixes = tuple(sys.argv[2:]) or ("k_search", "k_order", "k_locate")
    out = {}
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith(prefixes):
                continue
            k = r["Kernel_Name"].split("(")[0]
            e = out.setdefault(k, {}).setdefault(r["Dispatch_Id"], {})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
            e["duration_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            e["vgpr"] = int(r["VGPR_Count"])
            e["lds"] = int(r["LDS_Block_Size"])
    print(json.dumps({k: list(v.values()) for k, v in out.items()}))


if __name__ == "__main__":
    main()
