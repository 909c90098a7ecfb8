"""RoPE pass over the fused QKV buffer at the step's shape (GPT-NeoX 3B: 32 heads x 80, rotary_pct 1.0), streaming from HBM:
--rotate N distinct buffers so the Infinity Cache cannot serve the re-reads.  Algorithmic bytes = q and k read + written."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unimp_amd import ops
bf = torch.bfloat16
ROT = int(sys.argv[sys.argv.index("--rotate") + 1]) if "--rotate" in sys.argv else 3
B, L, H, hd = 64, 512, 32, 80
bufs = [torch.randn(B * L, 3 * H * hd, device="cuda").to(bf) for _ in range(ROT)]
pos = torch.arange(L, device="cuda", dtype=torch.float32)
inv = 1.0 / (10000 ** (torch.arange(0, hd, 2, device="cuda").float() / hd))
ang = pos[:, None] * inv[None]
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
def run(i): ops.rope_(bufs[i], L, H, 3 * hd, hd, (0, hd), cos, sin)
for i in range(ROT): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30
e0.record()
for i in range(n): run(i % ROT)
e1.record(); e1.synchronize()
t = e0.elapsed_time(e1) / n
nb = B * L * 2 * H * hd * 2 * 2
print(f"rope [{B*L} x {3*H*hd}] q,k: {t*1e3:7.1f} us  {nb/t/1e9:.2f} TB/s")
