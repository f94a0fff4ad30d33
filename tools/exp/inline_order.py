"""Experiment: why is the c3 record inside the default bench line ~7 % below `bench.py --config c3`?  Runs the c3 timed region
(a) first in a fresh process, (b) after the c2 measurement, (c) after c2 + gc.collect() + empty_cache(), printing each."""
import gc
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "--steps", "20", "--warmup", "5", "--timed-only", "--repeats", "3"]
import torch
import bench

a = bench.parse()
ctx = bench.Ctx()
import translation_transformer_amd as tta


def c3(tag):
    rec = bench.measure_beam(ctx, a, tta, "c3", 64, 2, full=False)
    print(tag, "c3", round(rec["value"], 1), [round(x, 4) for x in rec["repeats"]["seconds_in_run_order"]], flush=True)


order = os.environ.get("EXP_ORDER", "c3,c2,c3,gc,c3")
a.steps, a.batch_size, a.n_drafts = 20, 32, 3
for what in order.split(","):
    if what == "c3":
        c3("after: " + order)
    elif what == "c2":
        line = bench.measure_c2(ctx, a, tta)
        print("c2", round(line["value"], 1), flush=True)
    elif what == "gc":
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
