#!/usr/bin/env python3
"""Developer tool: throughput of an ad-hoc table shape (plain A/C/G/T samples), both memo forms.
usage: python tools/bench_custom.py S L [mm] [delta]   (run on the GPU box)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fqtk_amd import BarcodeMatcher
S, L = int(sys.argv[1]), int(sys.argv[2])
mm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
delta = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rng = np.random.default_rng(5)
seen = set()
while len(seen) < S:
    seen.add("".join(rng.choice(list("ACGT"), size=L)))
bcs = sorted(seen)
n = 100_000_000
m0 = n // 100
src = rng.integers(0, S, size=m0)
bc = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in bcs])[src]
obs = bc.copy()
flip = rng.random(obs.shape) < 0.015
obs[flip] = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, int(flip.sum()))]
rnd = rng.random(m0) < 0.1
obs[rnd] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (int(rnd.sum()), L))]
stride = (L + 3) // 4 * 4
pad = np.zeros((m0, stride), dtype=np.uint8)
pad[:, :L] = obs
d = torch.from_numpy(np.tile(pad, (100, 1))).cuda()
out = torch.empty(n, dtype=torch.int32, device="cuda")
cnt = torch.zeros(S + 1, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for kind in (2, 1):
    m = BarcodeMatcher(bcs, mm, delta)
    if kind == 1:
        if m.memo_kind != 2:
            continue
        m.memo_kind = 1
    for _ in range(2):
        m.assign_batch_device(d.data_ptr(), stride, n, out.data_ptr(), cnt.data_ptr(), stream=st)
    torch.cuda.synchronize()
    t = time.perf_counter()   # (untimed) 60 ms of launches: the device's clocks settle (profiles/r06_bench_window.txt)
    while time.perf_counter() - t < 0.06:
        for _ in range(10):
            m.assign_batch_device(d.data_ptr(), stride, n, out.data_ptr(), cnt.data_ptr(), stream=st)
        torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        m.assign_batch_device(d.data_ptr(), stride, n, out.data_ptr(), cnt.data_ptr(), stream=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print(f"S={S} L={L} mm={mm} memo_kind={m.memo_kind} entries={m.memo_entries}: {n / dt / 1e9:.1f} G reads/s, "
          f"{n * (stride + 4) / dt / 1e9:.0f} GB/s")
