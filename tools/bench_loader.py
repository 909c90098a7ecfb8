"""F2 loader throughput beside a running train step (VERDICT r3 missing #4).  The reference feeds each rank from 4 DataLoader workers that
open the item JPEGs with PIL and run the CLIP transform on the CPU (mmrec.py:403, rec_dataset.py:90-107); here the host only DECODES
(PIL, threads: the decoder releases the GIL), the decoded bytes cross PCIe through ImagePreprocessor's pinned ring and the GPU does the
bicubic resize + normalisation (csrc/preprocess.hip).  At the headline rate (~100 samples/s x 8 history images) the step consumes ~800
images/s per GPU.  Three measurements on cfg2 at b = 64 (512 images per step):
  (a) the loader alone: JPEG bytes -> decode (W threads) -> submit (pack into pinned memory, async H2D) -> get (resize kernel), images/s;
  (b) the train step alone on pre-staged synthetic batches, ms per step;
  (c) both together: batch i + 1 is decoded and submitted while step i runs; ms per step and images/s.
usage: python tools/bench_loader.py [workers=8] [steps=8] [jpeg side=500]"""
import io
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402
from PIL import Image       # noqa: E402
import bench                # noqa: E402
from unimp_amd.data import ImagePreprocessor       # noqa: E402
from unimp_amd.synthetic import make_batch         # noqa: E402
from unimp_amd.train import Trainer                # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
SIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 500
B, T, L = 64, 8, 512
dev = torch.device("cuda")

# a pool of distinct JPEGs with photographic statistics (smooth structure + texture): ~40-60 KB each at quality 90, like product photos
rng = np.random.default_rng(0)
jpegs = []
for i in range(128):
    y, x = np.mgrid[0:SIDE, 0:SIDE].astype(np.float32) / SIDE
    img = np.stack([np.sin(6.3 * (x * rng.uniform(0.5, 3) + y * rng.uniform(0.5, 3)) + rng.uniform(0, 6)) for _ in range(3)], -1) * 90 + 128
    img += rng.normal(0, 12, img.shape)
    buf = io.BytesIO()
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, format="JPEG", quality=90)
    jpegs.append(buf.getvalue())
print(f"{len(jpegs)} JPEGs of {SIDE} x {SIDE}, {sum(map(len, jpegs)) / len(jpegs) / 1024:.0f} KiB each; {W} decode threads; host cores {os.cpu_count()}", flush=True)


def decode(k):
    return np.asarray(Image.open(io.BytesIO(jpegs[k % len(jpegs)])).convert("RGB"))


pool = ThreadPoolExecutor(W)
pre = ImagePreprocessor(device=dev)
n_img = B * T


def decode_batch(step):
    return list(pool.map(decode, range(step * n_img, (step + 1) * n_img)))


# (a) loader alone
imgs = decode_batch(0)
pre.submit(imgs).get(); torch.cuda.synchronize()
t0 = time.perf_counter()
pend = None
for s in range(STEPS):
    imgs = decode_batch(s)
    nxt = pre.submit(imgs)
    if pend is not None:
        pend.get()
    pend = nxt
pend.get(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"(a) loader alone: {STEPS * n_img / dt:8.0f} images/s  ({dt / STEPS * 1e3:.0f} ms per batch of {n_img})", flush=True)
t0 = time.perf_counter()
for s in range(3):
    decode_batch(s)
dtd = time.perf_counter() - t0
print(f"    decode only : {3 * n_img / dtd:8.0f} images/s with {W} threads", flush=True)

# the model and trainer of the headline configuration
model, layout = bench.build_cfg2(dev)
trainer = Trainer(model, layout.special(), lr=2e-4, weight_decay=0.1, gamma=2.0, use_reweight=True, lr_scheduler="cosine", warmup_steps=10, total_steps=10000)
batches = [make_batch(layout, B, T, L, seed=1234 + 1000 * i, device=dev, vision_dtype=torch.bfloat16) for i in range(4)]
for i in range(4):
    trainer.step(batches[i % 4])
torch.cuda.synchronize()

# (b) train step alone
t0 = time.perf_counter()
for s in range(STEPS):
    trainer.step(batches[s % 4])
torch.cuda.synchronize()
dtb = time.perf_counter() - t0
print(f"(b) train step alone: {dtb / STEPS * 1e3:7.1f} ms per step ({B * STEPS / dtb:.1f} samples/s)", flush=True)

# (c) together: decode + submit batch s + 1 (host threads + copy stream) while step s runs on the compute stream
fut = pool.submit(decode_batch, 0)
pend = pre.submit(fut.result())
fut = pool.submit(decode_batch, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(STEPS):
    vis = pend.get().view(B, T, 1, 3, 224, 224)
    pend = pre.submit(fut.result())                      # batch s + 1: bytes packed into pinned memory, H2D on the copy stream
    fut = pool.submit(decode_batch, s + 2)               # batch s + 2 decodes on the host threads while the step runs
    b = dict(batches[s % 4], vision_x=vis)
    trainer.step(b)
torch.cuda.synchronize()
dtc = time.perf_counter() - t0
print(f"(c) step fed by the loader: {dtc / STEPS * 1e3:7.1f} ms per step ({B * STEPS / dtc:.1f} samples/s, {n_img * STEPS / dtc:.0f} images/s decoded, copied and resized "
      f"beside it); slowdown vs (b): {dtc / dtb - 1:+.1%}", flush=True)
