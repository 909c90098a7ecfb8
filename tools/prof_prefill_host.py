import sys, os, time, cProfile, pstats
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
K = 10
def run():
    with torch.no_grad():
        le._use_cached_vision_x = True
        model._encode_vision_x(vision_x=vx)
        sess = DecodeSession(model, 58, reorder=True, graph=True, beams=K)
        lg = sess.prefill(ids, None)
        torch.cuda.synchronize()
        model.clear_conditioned_layers(); le._use_cached_vision_x = False
run(); run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
# kernel count
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
ev = prof.key_averages()
tot = sum(e.count for e in ev if e.device_type == torch.autograd.DeviceType.CUDA) if hasattr(torch.autograd, "DeviceType") else -1
print("device events", tot)
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12)[:3000])
