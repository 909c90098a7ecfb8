"""Error budget of the bf16 storage points -- a script, not a collected test (minutes of CPU):

    python tests/error_budget.py [CFG2_SLIM|TINY|...]

The fp32 oracle is run with its storage points (oracle/numerics.py) rounded to bf16 one class at a time, all together (= the
product's storage precision) and all-but-one (what moving THAT class to fp32 would buy, e.g. an fp32 residual stream).  Reported
per configuration: logits rel-L2 against the pure-fp32 oracle, argmax agreement over the valid positions, loss relative error.
Test infrastructure: imports oracle/ (never the product)."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _parity as P                                   # noqa: E402
from oracle import numerics as N, train_step as ots   # noqa: E402


def run(cfgname="CFG2_SLIM"):
    cfg = getattr(P, cfgname)
    om, layout = P.build_oracle(cfg)
    batch = P.make_batch(cfg, layout)
    sp = layout.special()
    labels = torch.from_numpy(ots.label_mask_loop(batch["lang_x"].numpy(), sp["answer_id"], sp["eoc_id"], sp["pad_id"], sp["media_id"]))
    valid = batch["attention_mask"].bool()

    def fwd(tags):
        with torch.no_grad(), N.storage(*tags):
            lg = om(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"]
            return lg, float(ots.weighted_focal_ce(lg, labels, batch["weights"], 2.0, True))
    t0 = time.time()
    ref, ref_loss = fwd(())
    print(f"[{cfgname}] fp32 oracle forward {time.time() - t0:.1f} s; logits {tuple(ref.shape)}", flush=True)
    rows = [("(bf16 autocast of the oracle, for scale)", None)]
    rows += [(f"only {t}", (t,)) for t in N.ALL]
    rows += [("ALL storage points (= the product's storage precision)", N.ALL)]
    rows += [(f"ALL but {t} (that class kept in fp32)", tuple(x for x in N.ALL if x != t)) for t in ("res", "ln", "attn_o", "attn_p", "act", "gemm", "logits")]
    rows += [("ALL but res + ln", tuple(x for x in N.ALL if x not in ("res", "ln"))),
             ("ALL but res + logits", tuple(x for x in N.ALL if x not in ("res", "logits")))]
    print(f"{'storage rounded to bf16':62s} {'logits rel-L2':>13s} {'argmax agree':>12s} {'loss rel err':>12s}")
    for name, tags in rows:
        if tags is None:
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
                lg = om(batch["vision_x"], batch["lang_x"], batch["attention_mask"])["logits"].float()
            loss = float(ots.weighted_focal_ce(lg, labels, batch["weights"], 2.0, True))
        else:
            lg, loss = fwd(tags)
        agree = float((lg.argmax(-1) == ref.argmax(-1))[valid].float().mean())
        print(f"{name:62s} {P.rel_l2(lg, ref):13.3e} {agree:12.4f} {abs(loss - ref_loss) / abs(ref_loss):12.2e}", flush=True)


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "CFG2_SLIM")
