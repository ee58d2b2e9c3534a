import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
import numpy as np
if mode == "torch_first":
    import torch
    print("avail", torch.cuda.is_available()); x = torch.zeros(1, device="cuda"); print("torch ok")
from palace_amd import capi
with capi.Ctx(0) as ctx:
    b = ctx.upload(np.zeros(10, dtype=np.uint8)); print("ctx ok")
import torch
try:
    g = torch.Generator(device=torch.device("cuda", 0)); y = torch.zeros(4, device="cuda"); print("torch after ctx ok")
except Exception as e:
    print("torch after ctx FAILED:", str(e)[:100])
os.system("grep -c amdhip /proc/%d/maps; grep amdhip /proc/%d/maps | awk '{print $6}' | sort -u" % (os.getpid(), os.getpid()))
