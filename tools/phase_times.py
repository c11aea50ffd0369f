#!/usr/bin/env python3
"""Developer tool: where a wave's cycles go in the table-form memo kernel (a library built with -DFQTK_DEV_TIMING must be
the loaded libfqtk_match.so: tools/phase_times.sh swaps it in).  usage: phase_times.py <config> [--memo-table] [--reads N]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fqtk_amd import BarcodeMatcher, synth, _lib

ap = argparse.ArgumentParser()
ap.add_argument("config", type=int)
ap.add_argument("--memo-table", action="store_true")
ap.add_argument("--reads", type=int, default=0)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
cfg = synth.CONFIGS[a.config]
n = a.reads or min(cfg.n_reads, 100_000_000)
w = synth.Workload(cfg)
st = torch.cuda.current_stream().cuda_stream
d_obs = torch.empty((n, cfg.stride), dtype=torch.uint8, device="cuda")
for lo in range(0, n, 50_000_000):
    w.fill_device(lo, min(50_000_000, n - lo), d_obs.data_ptr() + lo * cfg.stride, st)
d_out = torch.empty(n, dtype=torch.int32, device="cuda")
d_cnt = torch.zeros(cfg.n_samples + 1, dtype=torch.int64, device="cuda")
m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, device=0)
if a.memo_table:
    m.memo_kind = BarcodeMatcher.MEMO_TABLE
fn = _lib.load().fqtk_dev_phase_cycles
fn.restype = C.c_int
buf = (C.c_ulonglong * 16)()
for _ in range(2):
    m.assign_batch_device(d_obs.data_ptr(), cfg.stride, n, d_out.data_ptr(), d_cnt.data_ptr(), stream=st)
torch.cuda.synchronize()
assert fn(buf) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    m.assign_batch_device(d_obs.data_ptr(), cfg.stride, n, d_out.data_ptr(), d_cnt.data_ptr(), stream=st)
e1.record()
torch.cuda.synchronize()
assert fn(buf) == 0
ms = e0.elapsed_time(e1) / a.steps
names = ["row stream (load wait)", "encode + hashes", "LDS cache / hot table", "direct gather", "cuckoo table", "rare paths + histogram",
         "result stream (store wait)", "after the loop"]
tot = sum(buf[k] for k in range(8))
waves = buf[15] / a.steps
print(f"cfg {a.config} memo_kind={m.memo_kind} {'(table pinned)' if a.memo_table else ''}: {n / ms / 1e6:.1f} G reads/s WITH the marks ({ms:.4f} ms), "
      f"{waves:.0f} waves, {tot / buf[15] / 1e3:.1f} k cycles per wave")
for k in range(8):
    print(f"  {names[k]:32s} {100.0 * buf[k] / tot:5.1f} %   {buf[k] / buf[15]:10.0f} cycles per wave")
