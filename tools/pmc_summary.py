#!/usr/bin/env python3
"""Mean per counter over the demux kernel's dispatches of one rocprofv3 --pmc output directory; the
(large) CSV is deleted afterwards.  usage: python3 tools/pmc_summary.py <dir> <label>"""
import collections
import csv
import glob
import os
import sys

d, label = sys.argv[1], sys.argv[2]
fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
if not fs:
    print(label, "no counter csv")
    sys.exit(0)
agg = collections.defaultdict(list)
meta = None
for r in csv.DictReader(open(fs[0])):
    if "demux_" in r["Kernel_Name"] and "fixup" not in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
print(label, "grid/wg/lds/vgpr/sgpr", meta)
for k, v in sorted(agg.items()):
    print("      %-36s %.6g   (n=%d)" % (k, sum(v) / len(v), len(v)))
for f in glob.glob(os.path.join(d, "**", "*"), recursive=True):
    if os.path.isfile(f):
        os.remove(f)
