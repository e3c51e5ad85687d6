# -*- coding: utf-8 -*-
"""
One process per GPU for the command line:  python -m quade_amd.launch -n N -c Conf.txt

Starts N copies of `python -m quade_amd.quade -c Conf.txt` with QUADE_RANK / QUADE_WORLD /
QUADE_LOCAL_RANK / QUADE_RUN_TOKEN in their environment (rank r drives GPU r), waits for them, and
tears the others down as soon as one fails, so that no rank is left waiting in the count all-reduce.
No PyTorch: the ranks find each other through the output directory (quade_amd/dist.py).
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import time
import uuid


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m quade_amd.launch")
    ap.add_argument("-n", "--nproc", type=int, required=True, help="number of ranks = GPUs")
    ap.add_argument("-c", dest="conf_file", required=True, help="configuration file (as for Quade.py -c)")
    ap.add_argument("--module", default="quade_amd.quade", help="module run by every rank (default: the command line driver)")
    args = ap.parse_args(argv)
    token = uuid.uuid4().hex[:16]
    procs = []
    for r in range(args.nproc):
        env = dict(os.environ, QUADE_RANK=str(r), QUADE_WORLD=str(args.nproc), QUADE_LOCAL_RANK=str(r), QUADE_LOCAL_WORLD=str(args.nproc),
                   QUADE_RUN_TOKEN=token)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-m", args.module, "-c", args.conf_file], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:  # a failed rank: the others would wait for it in the collective
                    q.terminate()
    return rc


if __name__ == "__main__":
    sys.exit(main())
