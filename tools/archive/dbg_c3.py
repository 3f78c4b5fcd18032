"""Debug (GPU box): the parity leg of a full-size configuration again and again (a failure seen once in the bench's configs leg)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYTORCH_ALLOC_CONF", "expandable_segments:True")
import numpy as np, torch
import mizuroute_amd as m
from mizuroute_amd import uh as uhmod
sys.argv = sys.argv[:1]
import bench
cfg = os.environ.get("CFG", "c3")
lb = bench.Loopback(torch, m, uhmod, cfg, 8)
for k in range(int(os.environ.get("REPS", "4"))):
    for Wa, Ka in ((128, 1), (256, 2)):
        t0 = time.perf_counter()
        try:
            rep, whole, res = lb.parity(Wa, Ka)
            print(k, Wa, Ka, rep["partitioned_equals_whole_bit_for_bit"], "%.1f s" % (time.perf_counter() - t0), flush=True)
        except Exception as e:
            print(k, Wa, Ka, "FAILED", e, flush=True)
