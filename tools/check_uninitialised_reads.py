"""Does anything a train step computes depend on memory it did not write?  A TINY-model step runs three times in one process from the same weights
and batch; before the second and third run every free block of the caching allocator is filled with 0xFF (NaN in bf16 and fp32) resp. 0x00, so that
whatever torch.empty() hands out next holds that pattern.  Loss and updated parameters must have the same bits in all three.  Not a pytest file."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from unimp_amd.train import Trainer


def poison(byte):
    torch.cuda.synchronize()
    sizes = [b["size"] for seg in torch.cuda.memory_snapshot() for b in seg["blocks"] if b["state"] == "inactive"]
    held = []
    for sz in sorted(sizes, reverse=True):
        try:
            t = torch.empty(sz, dtype=torch.uint8, device="cuda"); t.fill_(byte); held.append(t)
        except RuntimeError:
            pass
    n = sum(t.numel() for t in held)
    del held
    torch.cuda.synchronize()
    return n


for cfg_name in ("TINY",) + tuple(n for n in ("TINY_MPT", "TINY_LLAMA") if hasattr(P, n)):
    cfg = getattr(P, cfg_name)
    om, layout = P.build_oracle(cfg)
    batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=11).items()}
    outs = []
    for tag, byte in (("as is", None), ("free memory = 0xFF", 0xFF), ("free memory = 0x00", 0x00)):
        tr = Trainer(P.build_hip(cfg, om, layout), layout.special(), lr=1e-3, lr_scheduler="constant")
        n = poison(byte) if byte is not None else 0
        l, _ = tr.step(batch)
        l2, _ = tr.step(batch)
        torch.cuda.synchronize()
        outs.append((l.item(), l2.item(), tr.opt.flat_p.clone()))
        print(f"{cfg_name} {tag:22s} poisoned {n >> 20:5d} MiB  loss {l.item()!r} second step {l2.item()!r}")
        del tr
    ok = all(o[0] == outs[0][0] and o[1] == outs[0][1] and torch.equal(o[2], outs[0][2]) for o in outs[1:])
    print(f"{cfg_name}: same bits in all three runs: {ok}" + ("" if ok else f"   differing parameters: {[int((o[2] != outs[0][2]).sum()) for o in outs[1:]]}"))


# ---- the cached decode: the tokens and last logits of generate() (greedy and 4 beams) under the same poisoning
if os.environ.get("CHECK_DECODE", "1") != "0":
    import bench
    from unimp_amd.synthetic import make_batch
    from unimp_amd.decode import DecodeSession
    dev = torch.device("cuda")
    model, layout = bench.build_cfg2(dev, gate=0.5)
    model.eval()
    bt = make_batch(layout, 1, 8, 512, seed=7, device="cuda", vision_dtype=torch.bfloat16)
    n = int(bt["attention_mask"][0].sum())
    ids, vx = bt["lang_x"][:1, :n - 2], bt["vision_x"][:1]
    for K in (1, 10):
        outs = []
        for tag, byte in (("as is", None), ("free memory = 0xFF", 0xFF), ("free memory = 0x00", 0x00)):
            npo = poison(byte) if byte is not None else 0
            with torch.no_grad():
                model.lang_encoder._use_cached_vision_x = True
                model._encode_vision_x(vision_x=vx)
                s = DecodeSession(model, 24, reorder=K > 1, graph=True, beams=K)
                lg = s.prefill(ids, None)
                tok = lg.float().argmax(-1); src = torch.arange(lg.shape[0], device=dev)
                toks = [tok.clone()]
                for _ in range(12):
                    if byte is not None:
                        poison(byte)
                    lg = s.step(tok, src if K > 1 else None)
                    tok = lg.float().argmax(-1); toks.append(tok.clone())
                model.clear_conditioned_layers(); model.lang_encoder._use_cached_vision_x = False
            outs.append((torch.stack(toks), lg.clone()))
            print(f"decode K={K} {tag:22s} poisoned {npo >> 20:6d} MiB  tokens {outs[-1][0][:, 0].tolist()[:8]}")
        ok = all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs[1:])
        print(f"decode K={K}: same tokens and last-step logits bits in all three runs: {ok}")
