import sys, os
sys.path.insert(0, '.')
mode = sys.argv[1]
if mode == "lib_first":
    from mpsfm_amd import capi
    print("devcount(lib)", capi.device_count())
    from mpsfm_amd.synthetic import make_scene
    p,_ = make_scene(4, 50, True)
    print(capi.ba_solve(p)["final_cost"])
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    import ctypes
    print([l for l in open("/proc/self/maps").read().split() if "amdhip" in l][:4])
else:
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    from mpsfm_amd import capi
    print("devcount(lib)", capi.device_count())
    print(set(l for l in open("/proc/self/maps").read().split() if "amdhip" in l))
