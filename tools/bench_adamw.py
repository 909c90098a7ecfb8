"""fused clip + AdamW over the flat buffers at the step's size (1.34 G trainable parameters): ms per call and TB/s
(30 bytes per parameter: fp32 master / m / v read + written, bf16 gradient read + zeroed, bf16 parameter written)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
n = 1339915296 // 8 * 8
master = torch.randn(n, device="cuda") * 0.02
m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
p16 = master.to(torch.bfloat16); g16 = (torch.randn(n, device="cuda") * 1e-3).to(torch.bfloat16)
ss = torch.ones(1025, device="cuda")
def run(i): ops.adamw_flat(master, m, v, p16, g16, n, 1e-4, 0.9, 0.999, 1e-8, 0.1, i + 1, ss, 1.0, 1.0, zero_grad=True)
run(0); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5): run(i + 1)
e1.record(); e1.synchronize()
t = e0.elapsed_time(e1) / 5
print(f"adamw var={os.environ.get('UNIMP_ADAMW_VAR','0')} grid={os.environ.get('UNIMP_ADAMW_GRID','8192')}: {t:.3f} ms  {30*n/t/1e9:.2f} TB/s")
