"""dev: how uneven are the per-block symbol counts inside one wavefront of the lane-parallel MQ kernel (C3 workload)?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "go-jpeg2000_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from j2kgfx.codec import FramePlan
fr = bench.synth_frame(np, 1)
fr = (fr.astype(np.int64) * 4095 // 255).astype(np.int32)
p = FramePlan(3840, 2160, 3, precision=12, lossless=False, quality=75, num_resolutions=6, cb=(64, 64), tile=(512, 512), coder=0)
d = torch.from_numpy(fr).to(p.device)
i = p.info; n = int(i.blocks)
co = p.alloc_coeff(); sl = p.empty(i.bytes_cap, torch.uint8)
le = p.empty(n, torch.int32); nb = p.empty(n, torch.uint8)
p.forward(d, co); p.encode_blocks(co, sl, le, nb); p.ctx.sync()
nbh = nb.cpu().numpy().astype(np.int64); leh = le.cpu().numpy().astype(np.int64)
print("blocks", n, "numbps mean %.2f min %d max %d" % (nbh.mean(), nbh.min(), nbh.max()), "bytes/block mean %.0f" % leh.mean())
print("hist", np.bincount(nbh))
for K in (2, 4, 7, 8, 16, 64):
    m = (n + K - 1) // K
    pad = np.zeros(m * K, np.int64); pad[:n] = nbh
    g = pad.reshape(m, K)
    print("K=%d  sum(max)/sum(mean) = %.3f   max over waves %d" % (K, g.max(1).sum() / g.mean(1).sum(), g.max(1).max()))
