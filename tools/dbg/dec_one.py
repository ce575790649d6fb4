import sys, numpy as np
sys.path.insert(0, "go-jpeg2000_amd"); sys.path.insert(0, "oracle")
import oracle as orc
from j2kgfx import entropy as ent
hexs = "c7bc0b8618122b313595349d4a61c0fa4426b17b414129ff01" + "00" * 128 + "0b00018080b00010000008af000001a003080000003a80000001a103080000005a07008030016c00040000ac"
b = bytes.fromhex(hexs)
print(len(b))
got = ent.NewHTDecoder(16, 16).Decode(b, 0, 0).reshape(16, 16)
ref = orc.ht_decode(np.frombuffer(b, np.uint8), 16, 16)
print(np.array_equal(got, ref)); print(got[::4]); print(ref[::4])
