#!/usr/bin/env python3
"""Diagnostic (GPU box): libquade_hip.so and torch share one HIP runtime whatever the import order.
usage: python tools/check_load_order.py lib_first|torch_first"""
import os
import subprocess
import sys


def main(mode):
    sys.path.insert(0, os.getcwd())
    if mode == "lib_first":
        from quade_amd.hip_backend import Engine
        e = Engine(0)
        print("engine ok", e.device_info()["name"])
        import torch
        try:
            torch.cuda.init()
            print("torch after lib: ok", torch.cuda.device_count())
            x = torch.ones(4, device="cuda")
            print(x.sum().item())
        except Exception as ex:
            print("torch after lib: FAIL", ex)
    else:
        import torch
        torch.cuda.init()
        print("torch ok")
        from quade_amd.hip_backend import Engine
        Engine(0)
        print("engine after torch ok")
    print(subprocess.run("grep -E 'hip64|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid(),
                         shell=True, capture_output=True, text=True).stdout)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "torch_first")
