import sys, numpy as np, torch
sys.path.insert(0, "go-jpeg2000_amd"); sys.path.insert(0, "oracle")
import oracle as orc
from j2kgfx.codec import FramePlan
# replay the fuzz RNG up to the failing frame
exec(open("tools/fuzz_gpu.py").read().split("t0 = time.time()")[0].split("budget =")[0])
import time
rng = np.random.default_rng(1)
target = (3, 40, 33, 6, 16, 1, 8, 3)
while True:
    Cn = int(rng.choice([1, 3])); W = int(rng.choice([16, 24, 40, 64, 100, 128, 200, 256, 264, 512, 520, 776, 1024, 1032])); H = int(rng.choice([2, 3, 5, 16, 33, 64, 75, 128, 200]))
    if W * H * Cn > 600000: continue
    nres = int(rng.integers(1, 7)); cb = int(rng.choice([16, 32, 64])); coder = 1 if rng.random() < 0.75 else 0
    if coder == 0 and W * H * Cn > 120000: coder = 1
    prec = int(rng.choice([8, 10, 12, 16])); kind = int(rng.integers(0, 5)); top = (1 << prec) - 1
    if kind == 0: frame = rng.integers(0, top + 1, (Cn, H, W))
    elif kind == 1:
        yy, xx = np.mgrid[0:H, 0:W]; frame = np.clip(np.stack([(xx * top // max(W, 1) + c * 3) for c in range(Cn)]) + rng.integers(-2, 3, (Cn, H, W)), 0, top)
    elif kind == 2: frame = np.full((Cn, H, W), int(rng.integers(0, top + 1)))
    elif kind == 3:
        frame = np.zeros((Cn, H, W), np.int64); m = rng.random((Cn, H, W)) < 0.02; frame[m] = rng.integers(0, top + 1, int(m.sum()))
    else: frame = (top // 2) + rng.integers(-1, 2, (Cn, H, W))
    if (Cn, W, H, nres, cb, coder, prec, kind) == target: break
frame = frame.astype(np.int32)
np.save("gpurun_out/repro_frame.npy", frame)
plan = FramePlan(W, H, Cn, precision=prec, lossless=True, num_resolutions=nres, cb=(cb, cb), coder=coder)
coeff = plan.forward(torch.from_numpy(frame).to(plan.device))
stream, offs, lens, nb = plan.encode_stream(coeff)
decoded = plan.decode_blocks(stream, offs, lens, nb); plan.ctx.sync()
want = orc.preprocess([frame[c] for c in range(Cn)], W, H, prec, True, nres)
wb, wl, wn = orc.encode_tile_blocks(want, W, H, nres, cb, cb, coder)
blocks = plan.blocks(); doffs = plan.decoded_offsets(); dh = decoded.cpu().numpy()
pos = 0
for j in range(int(plan.info.blocks)):
    w_, h_ = int(blocks[j]["w"]), int(blocks[j]["h"])
    chunk = wb[pos:pos + int(wl[j])]; pos += int(wl[j])
    ref = orc.ht_decode(chunk, w_, h_)
    got = dh[int(doffs[j]):int(doffs[j]) + w_ * h_].reshape(h_, w_)
    if not np.array_equal(got, ref):
        print("job", j, "w h", w_, h_, "len", int(wl[j]), "bytes", bytes(chunk).hex())
        print("ref rows", ref[::4]); print("got rows", got[::4]); print("nonzero other rows got", np.abs(got).sum() - np.abs(got[::4]).sum())
        break
