import sys
sys.path.insert(0, '.')
import torch
from mpsfm_amd.dist import _DevView
a = torch.arange(8, dtype=torch.float64, device="cuda")
v = torch.as_tensor(_DevView(a.data_ptr(), 8), device="cuda")
print("same ptr", v.data_ptr() == a.data_ptr())
v.mul_(2); torch.cuda.synchronize(); print("aliased", a.tolist())
s = torch.cuda.Stream()
ext = torch.cuda.ExternalStream(s.cuda_stream)
with torch.cuda.stream(ext):
    v2 = torch.as_tensor(_DevView(a.data_ptr(), 8), device="cuda")
    print("same ptr under ext stream", v2.data_ptr() == a.data_ptr(), torch.cuda.current_stream().cuda_stream == s.cuda_stream)
