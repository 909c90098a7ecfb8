"""Which part of the third-generation dK/dV kernel costs what: the backward of the LM shape (b = 64) with parts of attention3.hip
switched off (UNIMP_A3_DBG is read once per process: one child process per setting).  Times are dQ + dK/dV; generation 4 = the
first-generation dK/dV kernel beside the same dQ kernel."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, torch
sys.path.insert(0, %r)
from unimp_amd import ops, _lib
B, H, S, D = 64, 32, 512, 80
torch.manual_seed(0)
qkv = torch.randn(B, S, H, 3 * D, device="cuda").to(torch.bfloat16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
do = torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16)
gen = int(sys.argv[1])
rope = (40, 13.287712379549449) if len(sys.argv) > 2 and sys.argv[2] == "rope" else None       # the LM's fused inverse rotation of dq / dk (adjacent-pair form)
_lib.lib().unimp_attn_set_generation(gen)
o, lse = ops.attn_fwd(q, k, v, D ** -0.5, ops.MASK_CAUSAL)
def run():
    ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, D ** -0.5, ops.MASK_CAUSAL, rope=rope)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); e1.synchronize()
print("%%.1f" %% (e0.elapsed_time(e1) / 20 * 1000))
''' % ROOT
names = {0: "everything", 1: "no tile arithmetic", 2: "no result stores", 4: "no K / V loads", 8: "no tile fetches", 3: "no arithmetic, no stores", 15: "nothing but the loop skeleton"}
for gen, dbg, rp in [(4, 0, ""), (2, 0, ""), (2, 2, ""), (4, 0, "rope"), (2, 0, "rope")]:
    env = dict(os.environ, UNIMP_A3_DBG=str(dbg))
    out = subprocess.run([sys.executable, "-c", CHILD, str(gen), rp], env=env, capture_output=True, text=True)
    us = out.stdout.strip().split("\n")[-1] if out.returncode == 0 else "failed: " + out.stderr[-300:]
    print(f"generation {gen}  UNIMP_A3_DBG={dbg:2d} ({names[dbg] if gen == 2 else 'first-generation dK/dV'}{', dq / dk rotated back in the epilogues' if rp else ''}): dQ + dK/dV {us} us", flush=True)
