import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import rware_amd
B = 16384
kw = rware_amd.env_kwargs("rware-small-4ag-v1")
env = rware_amd.WarehouseVecEnv(B, **kw)
eng = env.engines[0]
env.reset(seed=0)
acts = torch.randint(0, 5, (32, B, 4), dtype=torch.int32).cuda()
for t in range(20): eng.step_device(acts[t].data_ptr())
tl = eng.debug_timeline(acts[21].data_ptr())
tl = np.asarray(tl); tl = tl.reshape(tl.shape[0], -1)
hw = tl[:, 10]; xcc = tl[:, 11] & 0xF
w = [(hw >> (16 * k)) & 0xFFFF for k in range(4)]
simd = [(x >> 4) & 3 for x in w]; cu = (w[0] >> 8) & 0xF; se = (w[0] >> 13) & 7; sh = (w[0] >> 12) & 1
print("first 24 WGs: blk xcc se sh cu | simd of waves 0..3 | waveslot")
for b in range(24):
    print(b, xcc[b], se[b], sh[b], cu[b], [int(s[b]) for s in simd], [int(x[b] & 0xF) for x in w])
from collections import Counter
key = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
c = Counter(key); print("distinct CUs", len(c), "WG/CU histogram", Counter(c.values()))
# per CU: simds of wave0 of its WGs
d = {}
for b, k in enumerate(key): d.setdefault(k, []).append((b, int(simd[0][b])))
for k in list(d)[:10]: print(k, d[k])
same = sum(1 for k, v in d.items() if len(set(s for _, s in v)) == 1)
print("CUs where all wave0 on the same SIMD:", same, "of", len(d))
print("distribution of #distinct simds for wave0 per CU", Counter(len(set(s for _, s in v)) for v in d.values()))
