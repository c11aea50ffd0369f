#!/usr/bin/env python3
"""Developer tool (CPU only): what a WAVEFRONT pays in the BGZF compressor's LZ phase.  The phases run lane by lane on the CPU
with -DFQTK_BGZF_TRACE (every lane reports what it did in each of its steps); the lanes' steps are then put side by side as
the device runs them -- 64 lanes in lockstep, step k of every lane at the same time -- and the code regions a wavefront
would execute are counted: a region runs when ONE lane needs it.
    python tools/bgzf_lockstep.py [-DFLAG ...]"""
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CHILD = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, %(tools)r)
import bgzf_ratio as R
lib = C.CDLL(%(so)r); fn = lib.fqtk_host_bgzf_deflate_level; fn.restype = C.c_int64
rng = np.random.default_rng(1)
text = R.fastq_text(400, rng, %(qual)r)
b = text[65280:2 * 65280]
out = (C.c_uint8 * 70000)(); stored = C.c_int(0)
n = fn(b, C.c_uint32(len(b)), out, C.c_size_t(70000), C.byref(stored), 0, 5)
assert n > 0
"""


def main():
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    so = os.path.join(tempfile.mkdtemp(), "libshim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-DFQTK_BGZF_TRACE", *flags, "-o", so,
                           os.path.join(ROOT, "fqtk_amd/csrc/host/host_capi.cpp"), "-lz", "-ldl"])
    for name, qual in (("varied", b"FFFFFFFFFF:,#IIJJ<<AA"), ("binned", b"F" * 40 + b":,#"), ("constant", b"I")):
        err = subprocess.run([sys.executable, "-c", CHILD % {"tools": os.path.join(ROOT, "tools"), "so": so, "qual": qual}],
                             stderr=subprocess.PIPE, check=True).stderr.decode()
        # per (lane, step): passes [(which, [(c, l), ...])], steps
        steps = collections.defaultdict(lambda: {"passes": [], "mlen": 0, "open": 0})
        n_steps = collections.Counter()
        cur = None
        for line in err.split("\n"):
            if not line.startswith("LZ "):
                continue
            f = line.split()
            lane, step, kind = int(f[1]), int(f[2]), f[3]
            e = steps[(lane, step)]
            if kind == "R":
                e["passes"].append((int(f[4]), []))
            elif kind == "C":
                e["passes"][-1][1].append((int(f[4]), int(f[5])))
            elif kind == "S":
                e["mlen"] = int(f[4])
                n_steps[lane] = max(n_steps[lane], step + 1)
            elif kind == "O":   # a step of a match that is still being compared: rounds
                e["open"] = int(f[4])
                n_steps[lane] = max(n_steps[lane], step + 1)
        tot = collections.Counter()
        for wave in range(16):
            lanes = range(wave * 64, wave * 64 + 64)
            k_max = max((n_steps[l] for l in lanes), default=0)
            tot["wave steps"] += k_max
            for k in range(k_max):
                es = [steps[(l, k)] for l in lanes if (l, k) in steps]
                first = [e["passes"][0] for e in es if e["passes"]]
                second = [e["passes"][1] for e in es if len(e["passes"]) > 1]
                tot["lanes active"] += len(es)
                tot["rounds of open matches"] += max((e["open"] for e in es), default=0)
                tot["steps with an open match"] += any(e["open"] for e in es)
                if first:
                    tot["first passes"] += 1
                    tot["lanes in first passes"] += len(first)
                if second:
                    tot["second passes"] += 1
                if any(e["mlen"] for e in es):
                    tot["steps with a match taken"] += 1
                for ps in (first, second):
                    if not ps:
                        continue
                    # candidate sections as compiled: section c runs when one lane's candidate c is real
                    secs = set(c for _, cs in ps for c, _ in cs)
                    tot["candidate sections (one per candidate index)"] += len(secs)
                    tot["candidate sections (a loop over each lane's real candidates)"] += max(len(cs) for _, cs in ps)
                    rounds = lambda l: (l - 8) // 16 + 1 if l >= 8 else 0
                    tot["extension rounds (a loop per candidate index)"] += sum(max((rounds(l) for _, cs in ps for c, l in cs if c == s), default=0) for s in secs)
                    tot["extension rounds (one loop, every lane through its candidates)"] += max(sum(rounds(l) for _, l in cs) for _, cs in ps)
                    tot["extension rounds (one loop, only the lane's longest-so-far)"] += max(max((rounds(l) for _, l in cs), default=0) for _, cs in ps)
        print(name)
        for k, v in tot.items():
            print(f"    {k:70s} {v:7d}   per wave step {v / max(tot['wave steps'], 1):.3f}")


if __name__ == "__main__":
    main()
