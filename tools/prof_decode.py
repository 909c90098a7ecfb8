"""Kernel trace of the cached decode STEPS only (F1; eval_img_gen.py:102-111 greedy, eval_rec.py:100-110 K = 10): vision + prefill run before
the first marker, `steps` graph replays of the decode step sit between markers 101 and 102 (tools/trace_window.py cuts there).
usage: rocprofv3 --kernel-trace ... -- python3 tools/prof_decode.py [beams] [steps]   (also prints the host-timed ms per token-step)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch           # noqa: E402
import bench           # noqa: E402
from unimp_amd import ops        # noqa: E402
from unimp_amd.decode import DecodeSession       # noqa: E402
from unimp_amd.synthetic import make_batch       # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")
if os.environ.get("PROF_DECODE_MODEL", "4b") == "9b":        # the reference's 9b pair (MPT-7B dims, gated blocks every 4 layers; mmrec.py:515-524)
    model, layout = bench.build_cfg2(dev, gate=0.5, lang="anas-awadalla/mpt-7b", every=4)
else:
    model, layout = bench.build_cfg2(dev, gate=0.5)
model.eval()
bt = make_batch(layout, 1, 8, 512, seed=7, device="cuda", vision_dtype=torch.bfloat16)
n = int(bt["attention_mask"][0].sum())
ids, vx = bt["lang_x"][:1, :n - 2], bt["vision_x"][:1]
with torch.no_grad():
    model.lang_encoder._use_cached_vision_x = True
    model._encode_vision_x(vision_x=vx.unsqueeze(2) if vx.ndim == 5 else vx)
    s = DecodeSession(model, steps + 8, reorder=K > 1, graph=True, beams=K)
    logits = s.prefill(ids, None)
    R = logits.shape[0]
    tok = logits.float().argmax(-1)
    src = torch.arange(R, device=dev)
    for _ in range(4):                       # eager step, capture, two replays
        tok = s.step(tok, src if K > 1 else None).float().argmax(-1)
    torch.cuda.synchronize()
    ops.marker(101)
    t0 = time.perf_counter()
    for _ in range(steps):
        tok = s.step(tok, src if K > 1 else None).float().argmax(-1)
    ops.marker(102)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"decode K={K}: {steps} token-steps, {dt / steps * 1e3:.3f} ms per token-step (graph replay + argmax; prompt {ids.shape[1]} tokens)")
