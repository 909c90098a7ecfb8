import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from unimp_amd.synthetic import make_batch
from unimp_amd.decode import DecodeSession
dev = torch.device("cuda")
model, layout = bench.build_cfg2(dev, gate=0.5)
model.eval()
bt = make_batch(layout, 1, 8, 512, seed=7, device="cuda", vision_dtype=torch.bfloat16)
n = int(bt["attention_mask"][0].sum())
ids, vx = bt["lang_x"][:1, :n - 2], bt["vision_x"][:1]
le = model.lang_encoder
def T():
    torch.cuda.synchronize(); return time.perf_counter()
K = 10
kw = dict(num_beams=K, num_return_sequences=K, early_stopping=False, max_new_tokens=3, eos_token_id=-1, pad_token_id=layout.eos)
model.generate(vx, ids, **kw)
for rep in range(2):
    with torch.no_grad():
        t0 = T()
        le._use_cached_vision_x = True
        model._encode_vision_x(vision_x=vx)
        t1 = T()
        sess = DecodeSession(model, 58, reorder=True, graph=True, beams=K)
        lg = sess.prefill(ids, None)
        t2 = T()
        tok = lg.float().argmax(-1); src = torch.arange(K, device=dev)
        tok = sess.step(tok, src).float().argmax(-1)
        t3 = T()
        tok = sess.step(tok, src).float().argmax(-1)
        t4 = T()
        tok = sess.step(tok, src).float().argmax(-1)
        t5 = T()
        model.clear_conditioned_layers(); le._use_cached_vision_x = False
    print(f"vision {1e3*(t1-t0):.1f} ms, prefill {1e3*(t2-t1):.1f}, step 1 (capture or eager) {1e3*(t3-t2):.1f}, step 2 {1e3*(t4-t3):.1f}, step 3 {1e3*(t5-t4):.1f}")
