"""Timing of the row schedule on the bench workload: first call (graph captures for new shapes inside) against
repeated calls, and the host-side share (regrouping, replay).  Usage: python tools/time_rows.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import translation_transformer_amd as tta
from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
inflight = int(os.environ.get("TTX_INFLIGHT", "8"))
sd = bench.get_weights(1500, "cuda:0")
model = tta.NativeTransformer(sd, num_heads=8, pad_token_idx=PAD, device=0)
src_all, _ = SynthReactions(123456, "mit").dataset((steps + 2) * 32)
dev_batches = [torch.from_numpy(b).to("cuda:0") for b in batches(src_all, 32)]
warm, timed = dev_batches[:2], dev_batches[2:]
g = tta.TranslationInferenceGreedySpeculative(model, 200, 10, 3, PAD, BOS, EOS, C_TOK)
g.generate_many(warm * inflight, in_flight=inflight, reorder=True)
def run(mode, gs, fl):
    g = tta.TranslationInferenceGreedySpeculative(model, 200, 10, 3, PAD, BOS, EOS, C_TOK)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.generate_many(timed, in_flight=fl, reorder=mode, group_size=gs, on_error="skip")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"reorder={mode} group={gs} in_flight={fl}: {len(timed) * 32 / dt:8.1f} reactions/s  {dt * 1e3:7.1f} ms  "
          f"calls={g.model_calls_num} device_calls={g.stats_total.get('device_model_calls')}", flush=True)

if os.environ.get("TTX_ROWS_ONLY"):          # e.g. TTX_ROWS_ONLY=256,4 : one configuration, for rocprofv3
    for cfg in os.environ["TTX_ROWS_ONLY"].split(";"):
        gs, fl = (int(x) for x in cfg.split(","))
        for _ in range(3):
            run(True, gs, fl)
    sys.exit(0)
run(False, None, inflight)
for gs in (32, 64, 128, 256, 512):
    for fl in (4, 8, 12):
        if gs * fl > 2 * len(timed) * 32:
            continue
        run(True, gs, fl)      # first call: shapes new to the graph cache
        run(True, gs, fl)
