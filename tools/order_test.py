import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "lib_first":
    from quade_amd.hip_backend import Engine
    e = Engine(0)
    print("engine ok", e.device_info()["name"])
    import torch
    try:
        torch.cuda.init(); print("torch after lib: ok", torch.cuda.device_count())
        x = torch.ones(4, device="cuda"); print(x.sum().item())
    except Exception as ex:
        print("torch after lib: FAIL", ex)
else:
    import torch
    torch.cuda.init(); print("torch ok")
    from quade_amd.hip_backend import Engine
    e = Engine(0); print("engine after torch ok")
import subprocess
print(subprocess.run("grep -E 'hip64|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
