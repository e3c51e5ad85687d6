#!/usr/bin/env python3
"""profiles/traffic_<cfg>.json is written on the GPU box (no .git there): this stamps each file whose "commit" is
still empty with the newest commit whose quade_kernels.hip has the recorded hash (the sources the PMC passes ran on).
bench.py prints it as roofline.traffic_age_commit."""
import glob
import hashlib
import json
import os
import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = "quade_amd/csrc/quade_kernels.hip"
revs = subprocess.check_output(["git", "log", "--format=%h", "-n", "60", "--", K], cwd=root, text=True).split()
sha_of = {}
for r in revs:
    blob = subprocess.check_output(["git", "show", "%s:%s" % (r, K)], cwd=root)
    sha_of.setdefault(hashlib.sha256(blob).hexdigest()[:16], r)
with open(os.path.join(root, K), "rb") as fh:
    work = hashlib.sha256(fh.read()).hexdigest()[:16]
for f in sorted(glob.glob(os.path.join(root, "profiles", "traffic_*.json"))):
    j = json.load(open(f))
    if j.get("commit"):
        continue
    k = j.get("kernel_source_sha16")
    j["commit"] = sha_of.get(k) or ("uncommitted working tree" if k == work else None)
    json.dump(j, open(f, "w"))
    print(os.path.basename(f), "->", j["commit"])
