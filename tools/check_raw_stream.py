import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimp_amd import ops
a = torch.cuda.current_stream().cuda_stream
b = ops._stream()
print("same handle", a == b, a, b)
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    print("side", torch.cuda.current_stream().cuda_stream == ops._stream())
t0=time.perf_counter()
for _ in range(10000): ops._stream()
print("us per call", (time.perf_counter()-t0)/10000*1e6)
t0=time.perf_counter()
for _ in range(10000): torch.cuda.current_stream().cuda_stream
print("us per call (torch)", (time.perf_counter()-t0)/10000*1e6)
