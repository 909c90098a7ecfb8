"""repeated generate() calls at cfg2 size: reserved memory must stay flat (one HIP graph + private pool per call, freed on return)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from unimp_amd.synthetic import make_batch
model, layout = bench.build_cfg2(torch.device("cuda"), gate=0.5)
kw = dict(num_beams=10, num_return_sequences=10, early_stopping=True, max_new_tokens=20, eos_token_id=layout.eos, pad_token_id=layout.eos)
t0 = time.perf_counter()
for i in range(24):
    bt = make_batch(layout, 1, 8, 512, seed=100 + i, device="cuda", vision_dtype=torch.bfloat16)
    n = int(bt["attention_mask"][0].sum())
    out = model.generate(bt["vision_x"][:1], bt["lang_x"][:1, :n - 2], **kw)
    if i % 4 == 3:
        torch.cuda.synchronize()
        print(f"user {i + 1:3d}: prompt {n - 2} tokens, reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB, allocated {torch.cuda.memory_allocated() / 2**30:6.2f} GiB, "
              f"{(time.perf_counter() - t0) / (i + 1):.3f} s/user")

# several users per call (right-padded prompts, per-row positions)
for ub in (2, 4):
    users = []
    for i in range(ub * 3):
        bt = make_batch(layout, 1, 8, 512, seed=300 + i, device="cuda", vision_dtype=torch.bfloat16)
        n = int(bt["attention_mask"][0].sum()) - 2
        users.append((bt["vision_x"][0], bt["lang_x"][0, :n]))
    def call(group):
        L = max(u[1].numel() for u in group)
        ids = torch.full((len(group), L), layout.pad, dtype=torch.long, device="cuda")
        for r, u in enumerate(group):
            ids[r, :u[1].numel()] = u[1]
        mask = (torch.arange(L, device="cuda")[None, :] < torch.tensor([u[1].numel() for u in group], device="cuda")[:, None]).long()
        return model.generate(torch.stack([u[0] for u in group]), ids, attention_mask=mask, **kw)
    call(users[:ub]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for g in range(1, 3):
        call(users[g * ub:(g + 1) * ub])
    torch.cuda.synchronize()
    print(f"{ub} users per generate() call: {(time.perf_counter() - t0) / (2 * ub):.3f} s/user")
