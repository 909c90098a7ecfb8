import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from unimp_amd import ops, _lib
import bench
from unimp_amd.synthetic import make_batch
L = _lib.lib()
g = torch.Generator().manual_seed(1)
rows, D, vocab = 32768, 2560, 74053
model, layout = bench.build_cfg2(torch.device("cuda"), gate=0.5)
bt = make_batch(layout, 64, 8, 512, seed=7, device="cuda", vision_dtype=torch.bfloat16)
ids = bt["lang_x"].reshape(-1)
u, c = torch.unique(ids, return_counts=True)
print("rows", ids.numel(), "distinct", u.numel(), "largest runs", sorted(c.tolist())[-5:])
dout = torch.randn(rows, D, device="cuda").bfloat16()
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
acc = torch.zeros((vocab, D), dtype=torch.float32, device="cuda")
def old():
    L.unimp_embedding_bwd(ids.data_ptr(), dout.data_ptr(), dout.stride(0), acc.data_ptr(), D, rows, D, vocab, torch.cuda.current_stream().cuda_stream)
print("atomic kernel alone ms", t(old))
print("ops.embedding_bwd (zeros + sort + kernel + cast) ms", t(lambda: ops.embedding_bwd(ids, dout, vocab)))
sid, perm = torch.sort(ids, stable=True)
print("sort alone ms", t(lambda: torch.sort(ids, stable=True)))
scr = torch.empty(L.unimp_embedding_bwd_sorted_scratch(rows, D), dtype=torch.float32, device="cuda")
def new():
    L.unimp_embedding_bwd_sorted(sid.data_ptr(), perm.data_ptr(), dout.data_ptr(), dout.stride(0), acc.data_ptr(), D, scr.data_ptr(), rows, D, vocab, torch.cuda.current_stream().cuda_stream)
print("sorted kernel alone ms", t(new))
