import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
import aria_slam_amd as A
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.zeros(1).cuda().item())
    e = A.OrbHipExtractor(max_features=500); print("ext ok"); m = A.HipMatcher(); print("mat ok")
else:
    e = A.OrbHipExtractor(max_features=500); print("ext ok"); m = A.HipMatcher(); print("mat ok")
    import torch
    print("torch avail", torch.cuda.is_available(), torch.zeros(1).cuda().item())
os.system("ldd %s | grep -i hip" % A.library_path())
import ctypes
print([l.strip().split()[-1] for l in open("/proc/self/maps") if "amdhip" in l][:3])
