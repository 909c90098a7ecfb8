"""The data-parallel step on the GPU with two ranks: both processes share cuda:0 and exchange through gloo (RCCL needs one
device per rank, which a 1-GPU box cannot offer), so everything but the RCCL transport itself is the production path --
autograd hooks firing on HIP tensors, bucket slices of the flat bf16 gradient buffer, async all-reduce handles, 1/W folded
into the fused AdamW, replicas staying bit-identical."""
import os
import socket
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _parity as P
        from unimp_amd.optim import FlatAdamW
        from unimp_amd.train import Trainer
        cfg = P.TINY
        om, layout = P.build_oracle(cfg)                                   # same seed on both ranks: identical replicas
        batch = {k: v.cuda() for k, v in P.make_batch(cfg, layout, seed=100 + rank).items()}      # rank-dependent data
        # local gradient of this rank's batch, no exchange
        hm0 = P.build_hip(cfg, om, layout)
        t0 = Trainer.__new__(Trainer)
        t0.model, t0.sparse_head, t0.ids, t0.gamma, t0.use_reweight = hm0, False, layout.special(), 2.0, True
        opt0 = FlatAdamW(hm0.named_parameters())
        loss0, _, _, _ = Trainer.forward_loss(t0, batch)
        loss0.backward()
        local = opt0.flat_g.float().clone()
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean = sum(gathered) / world
        # the production step
        hm = P.build_hip(cfg, om, layout)
        tr = Trainer(hm, layout.special(), lr=1e-3, lr_scheduler="constant", bucket_bytes=1 << 16)       # several buckets
        assert tr.dp.world == world and len(tr.dp.buckets) > 3
        loss, _, _, _ = tr.forward_loss(batch)
        loss.backward()
        gscale = tr.dp.finish()
        red = tr.opt.flat_g.float() * gscale
        err = float((red - mean).norm() / mean.norm())
        assert err < 1e-2, err
        tr.opt.step(lr=1e-3, grad_scale=gscale)
        for _ in range(2):
            tr.step(batch)
        masters = [torch.empty_like(tr.opt.master) for _ in range(world)]
        dist.all_gather(masters, tr.opt.master)
        assert all(torch.equal(masters[0], m) for m in masters)            # replicas bit-identical after 3 steps
        assert not torch.equal(gathered[0], gathered[1])                   # ... although their data differed
        q.put((rank, "ok", err))
    except Exception as e:                                                 # noqa: BLE001 -- report to the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
