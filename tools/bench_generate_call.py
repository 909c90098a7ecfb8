import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from unimp_amd.synthetic import make_batch
dev = torch.device("cuda")
model, layout = bench.build_cfg2(dev, gate=0.5)
model.eval()
bt = make_batch(layout, 1, 8, 512, seed=7, device="cuda", vision_dtype=torch.bfloat16)
n = int(bt["attention_mask"][0].sum())
ids, vx = bt["lang_x"][:1, :n - 2], bt["vision_x"][:1]
for K, new in ((10, 50), (5, 256), (1, 600)):
    kw = dict(num_beams=K, num_return_sequences=K, early_stopping=False, max_new_tokens=new, eos_token_id=-1, pad_token_id=layout.eos)
    model.generate(vx, ids, **{**kw, "max_new_tokens": 3})
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = model.generate(vx, ids, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"K={K} new={new}: s per user {min(ts):.4f} (runs {[round(t, 4) for t in ts]})")
