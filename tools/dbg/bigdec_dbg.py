import sys, os
sys.path.insert(0, "go-jpeg2000_amd"); sys.path.insert(0, "oracle")
import numpy as np
import oracle
from j2kgfx import entropy as ent
np.set_printoptions(linewidth=250)
for (w, h) in [(65, 4)]:
    for seed in range(2):
        rng = np.random.default_rng(seed)
        g = rng.integers(0, 256, 4000).astype(np.uint8)
        g[g == 0xFF] = 0x7F
        got = ent.NewT1(w, h).Decode(bytes(g), 2, 0).reshape(h, w)
        want = oracle.t1_decode(g, 2, 0, w, h)
        bad = np.argwhere(got != want)
        print(w, h, "seed", seed, "mismatches", len(bad))
        print("want\n", want[:, :40]); print("got\n", got[:, :40]); print("xor\n", (np.abs(got) ^ np.abs(want))[:, :40])
