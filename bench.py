#!/usr/bin/env python3
"""Benchmark of the MI355X-native speculative-decoding inference path.

    python bench.py --gpus N --steps K --warmup W                  (N > 1 without a launcher: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json configs[1], "c2"): reactions/sec of greedy-speculative decoding at bs=32, draft_len=10, n_drafts=3,
max_len=200 on USPTO-MIT-shaped synthetic SMILES.  A "step" is one given batch of 32 reactions through the generator (encoder
+ the whole verify loop); the K given batches go through ``generate_many(reorder=True)`` (slot pools, the reference's loop
replayed per given batch: outputs and counters of per-batch ``generate``).  Inputs are resident in HBM before the timed region;
every rank decodes its own shard of the synthetic test set (weak scaling, no data-path collective); weights are broadcast once
from rank 0 and predictions gathered once at the end over RCCL.  The timed region is repeated (default 5 times, each bracketed
by barrier + synchronize, MAX over ranks) and the MEDIAN is the value ("5 timed repeats, median + min/max";
the reference's scripts repeat every setting, scripts/product_prediction.sh:160-193).  Rank 0 prints ONE JSON line, which at
N = 1 also carries the sub-records "c3" / "c4" (BASELINE configs[2] / configs[3]: beam-search speculative, batch pools), the
strings-in -> CSV-out pipeline, the roofline of the dominant kernel family and the CPU baseline.

Weights: there is no checkpoint offline, so full-size (d=256, 8 heads, FFN 2048; 4+4 and 6+6 layers) models are trained on the
synthetic task by tools/train_synth.py for a fixed step budget (cached under /tmp, or shipped as .weights_cache/).  That is
set-up, outside every timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0            # same guide: HBM3E 8 TB/s spec
BASELINE_REACTIONS_PER_S = 47.97  # BASELINE.md §1, bs=32 D=10 N=3 (reference's own run, unstated GPU)

# BASELINE.json configs[2] / configs[3]: generator settings of the reference's own grid optimum for that batch size
# (results_grid_search/results_product_500_beam_search_speculative_bs_4_report.txt:31 -> 7.42 reactions/s;
#  results_retro_500_beam_search_speculative_bs_8_nbest_10_report.txt:5 -> 6.12 reactions/s; scripts/product_prediction.sh:197-198,
#  scripts/single_step_retrosynthesis.sh:166-174; model depth configs/cfg_standard_*:90-103)
BEAM_CONFIGS = {
    "c3": dict(kind="mit", layers=4, bs=4, n_best=5, N=7, published=7.42, steps=64,
               name="USPTO-MIT-shaped synthetic SMILES, beam-search speculative n_best=5"),
    "c4": dict(kind="50k", layers=6, bs=8, n_best=10, N=2, published=6.12, steps=32,
               name="USPTO-50K-shaped synthetic SMILES (retrosynthesis), beam-search speculative n_best=10"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=("c2", "c3", "c4"), default="c2",
                    help="headline: BASELINE.json configs[1] (greedy speculative bs=32; default — c3 and c4 are then sub-records), "
                         "configs[2] (MIT beam speculative n_best=5 bs=4 N=7) or configs[3] (50K 6+6 beam speculative n_best=10 bs=8 N=2)")
    ap.add_argument("--steps", type=int, default=None, help="timed batches (default 256 for c2, 64 / 32 for c3 / c4)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=None, help="repetitions of the timed region; the median is reported (default 5)")
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--draft-len", type=int, default=10)
    ap.add_argument("--n-drafts", type=int, default=None)
    ap.add_argument("--smart", type=int, default=0, help="c3/c4: smart_drafts_mode of the timed run (the other mode is reported beside it)")
    ap.add_argument("--max-len", type=int, default=200)
    ap.add_argument("--train-steps", type=int, default=int(os.environ.get("TTX_TRAIN_STEPS", "1500")))
    ap.add_argument("--cpu-batches", type=int, default=None, help="batches of the workload timed on the host cores (default 3 / 1)")
    ap.add_argument("--schedule", choices=("rows", "batches"), default=os.environ.get("TTX_SCHEDULE", "rows"),
                    help="rows: slot pools over the rows / sources of all given batches (exact replay per batch); batches: as given")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("TTX_INFLIGHT", "8")),
                    help="batches decoded concurrently per GPU (1 = the reference's one-batch-at-a-time loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true", help="c2 headline without the c3 / c4 / string-pipeline sub-records")
    ap.add_argument("--timed-only", action="store_true",
                    help="only the timed region of the headline (one repeat unless --repeats is given): for rocprofv3 runs")
    a = ap.parse_args()
    if a.timed_only:
        a.no_profile = a.no_cpu_baseline = a.no_sub_records = True
    if a.repeats is None:
        a.repeats = 1 if a.timed_only else 5
    return a


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("TTX_CPU_THREADS", "64"))))


def log(*a):
    if os.environ.get("TTX_BENCH_VERBOSE") == "1":
        print("[bench]", *a, file=sys.stderr, flush=True)


def get_weights(train_steps: int, device: str, kind: str = "mit", layers: int = 4, info: dict | None = None) -> dict:
    """Weights for the synthetic task: a cache written by an earlier run on this box (/tmp) or shipped with the snapshot
    (.weights_cache/, git-ignored), else trained here (set-up, untimed; `info` records which and how long it took)."""
    path = os.environ.get("TTX_WEIGHTS" if kind == "mit" else "TTX_WEIGHTS_50K") or f"/tmp/ttx_synth_{kind}_{train_steps}.pt"
    for cand in (path, str(ROOT / ".weights_cache" / f"synth_{kind}_{train_steps}.pt")):   # caches written by earlier runs
        if os.path.exists(cand) and os.environ.get("TTX_NO_WEIGHTS_CACHE") != "1":             # =1: what a clean clone does
            if info is not None:
                info[kind] = {"weights": "cache", "path": cand}
            return torch.load(cand, weights_only=True, map_location="cpu")
    from tools.train_synth import train
    t0 = time.perf_counter()
    sd = train(kind, steps=train_steps, n_enc=layers, n_dec=layers, device=device, verbose=os.environ.get("TTX_BENCH_VERBOSE") == "1")
    if info is not None:
        info[kind] = {"weights": "trained in this run", "train_steps": train_steps, "train_seconds": round(time.perf_counter() - t0, 1)}
    try:
        torch.save(sd, path)
        extra = os.environ.get("TTX_SAVE_WEIGHTS")
        if extra:
            torch.save(sd, extra)
    except OSError:
        pass
    return sd


def work(cfg: dict, positions: int, src_tokens: int, steps: int, batches: int, kv_read_positions: int, kv_written_tokens: int) -> dict:
    """Algorithmic work of the KV-cached algorithm: dense FLOPs of every GEMM launch (per verified position
    Ld (12 d^2 + 4 d F) + 2 d V, per encoder token Le (8 d^2 + 4 d F) + Ld 4 d^2) and HBM bytes (weights once per step / per
    encoder pass, K/V cache reads and writes); `gemm_bytes` = operand bytes of the GEMM launches alone."""
    d, F, V, Le, Ld = cfg["d"], cfg["F"], cfg["V"], cfg["Le"], cfg["Ld"]
    dec_dense_per_pos = Ld * (12 * d * d + 4 * d * F) + 2 * d * V          # qkv+o+cq+co = 6 d^2 MAC -> 12 d^2 FLOP
    enc_dense_per_tok = Le * (8 * d * d + 4 * d * F)
    cross_kv_per_tok = Ld * 4 * d * d
    gemm_flops = positions * dec_dense_per_pos + src_tokens * (enc_dense_per_tok + cross_kv_per_tok)
    P_e = 4 * d * d + 4 * d + 2 * d * F + F + d + 4 * d
    P_d = 2 * (4 * d * d + 4 * d) + 2 * d * F + F + d + 6 * d
    W_enc = 4 * (Le * P_e + 2 * d + V * d)
    W_dec = 4 * (Ld * P_d + 2 * d + d * V + V)
    bytes_total = batches * W_enc + steps * W_dec + (kv_read_positions + kv_written_tokens) * 2 * d * 4 * Ld
    dec_io = 4 * (Ld * ((d + 3 * d) + 3 * (d + d) + (d + F) + (F + 2 * d)) + (d + V))
    enc_io = 4 * (Le * ((d + 3 * d) + (d + d) + (d + F) + (F + d)) + (d + Ld * 2 * d))
    gemm_bytes = positions * dec_io + src_tokens * enc_io + steps * W_dec + batches * (W_enc + 4 * Ld * 2 * d * d)
    return {"gemm_flops": float(gemm_flops), "bytes": float(bytes_total), "gemm_bytes": float(gemm_bytes), "steps": int(steps), "Ld": Ld}


def pmc_traffic_for(key: dict):
    """HBM-side traffic per GEMM launch for exactly this command, from the committed rocprofv3 --pmc passes
    (profiles/gemm_pmc_traffic.json: one entry per profiled command, written by tools/roofline_from_trace.py).  Returns
    (entry, None) or (None, reason): a figure measured on another workload is never reported."""
    path = ROOT / "profiles" / "gemm_pmc_traffic.json"
    try:
        entries = json.loads(path.read_text())
    except (OSError, ValueError):
        return None, "no PMC summary committed (profiles/gemm_pmc_traffic.json missing)"
    for e in entries:
        if all(e.get("command_key", {}).get(k) == v for k, v in key.items()):
            return dict(e, file=f"profiles/{e.get('source_file', 'gemm_pmc_traffic.json')}"), None
    return None, ("no PMC pass was collected for this command (" + ", ".join(f"{k}={v}" for k, v in key.items()) +
                  "); profiles/gemm_pmc_traffic.json lists the profiled ones")


class Ctx:
    """Process-wide set-up: rank, device, collective backend."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        n_dev = torch.cuda.device_count()
        if local_rank >= n_dev and os.environ.get("TTX_SHARE_GPU") != "1":
            raise SystemExit(f"rank {self.rank}: no GPU {local_rank} on this node ({n_dev} visible)")
        self.local_rank = local_rank % n_dev          # TTX_SHARE_GPU=1: rehearsal of the N>1 path on a 1-GPU box
        torch.cuda.set_device(self.local_rank)
        self.dev = f"cuda:{self.local_rank}"
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist_mod
            self.dist = dist_mod
            self.dist.init_process_group(os.environ.get("TTX_DIST_BACKEND", "nccl"))   # "nccl" = RCCL over xGMI
        self.setup = {}

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if not self.dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.dist.get_backend() == "gloo" else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, repeats: int, local: bool = False):
        """`repeats` runs of fn() -> result, each bracketed by barrier + synchronize on both sides, elapsed = MAX over ranks.
        Returns (sorted list of seconds, result of the median run, all seconds in run order).  `local`: a measurement only
        rank 0 makes (the records beside the headline): synchronize only, no collective — the other ranks are not there."""
        runs = []
        for _ in range(max(1, repeats)):
            if local:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = fn()
                torch.cuda.synchronize()
                runs.append((time.perf_counter() - t0, res))
                continue
            self.barrier()
            t0 = time.perf_counter()
            res = fn()
            self.barrier()
            runs.append((self.max_over_ranks(time.perf_counter() - t0), res))
        order = sorted(range(len(runs)), key=lambda i: runs[i][0])
        med = order[(len(order) - 1) // 2]
        return [runs[i][0] for i in order], runs[med][1], [r[0] for r in runs]


def spread(sorted_s: list, units: float) -> dict:
    """units per second over the repeats: median (the reported value), min, max and the individual runs."""
    vals = sorted(units / s for s in sorted_s)
    return {"n": len(vals), "median": units / median_seconds(sorted_s), "min": vals[0], "max": vals[-1], "values": vals}


def median_seconds(sorted_s: list) -> float:
    return sorted_s[(len(sorted_s) - 1) // 2]


# ------------------------------------------------------------------------------------------------------------------------
def roofline_record(kernel: str, prof: dict, w_pass: dict, w_timed: dict, timed_seconds: float, measured_on: str, pmc, pmc_why, extra=None) -> dict:
    """`prof`: NativeTransformer.kernel_profile() of the roofline pass; `w_pass` / `w_timed`: work() of that pass and of the
    timed region.  achieved = algorithmic GEMM FLOPs of the pass / sum of its per-launch event durations, each net of the
    bracketing overhead calibrated on the same stream; frac_wall = GEMM FLOPs of the timed region / its wall time / peak."""
    launches = max(1, prof["launches"])
    # every session of the pass must have bracketed its launches: a verify step alone makes 6 GEMM launches per decoder layer + 1
    least = int(w_pass.get("steps", 0)) * (6 * int(w_pass.get("Ld", 0)) + 1)
    if prof["launches"] < least:
        raise SystemExit(f"bench.py: the roofline pass bracketed {prof['launches']} GEMM launches but its {w_pass.get('steps')} verify steps "
                         f"alone make {least}: some sessions of the pass did not profile")
    raw_ms = prof["gemm_ms"]
    net_ms = max(1e-9, raw_ms - launches * prof["pair_overhead_ms"])
    ach = w_pass["gemm_flops"] / (net_ms * 1e-3) / 1e12
    ach_raw = w_pass["gemm_flops"] / (max(raw_ms, 1e-9) * 1e-3) / 1e12
    rec = {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
           "frac": ach / PEAK_F32_MATRIX_TFLOPS,
           "frac_wall": w_timed["gemm_flops"] / timed_seconds / 1e12 / PEAK_F32_MATRIX_TFLOPS,
           "traffic": pmc.get("bytes_per_launch") if pmc else None,
           "traffic_note": (f"HBM-side bytes per GEMM launch (2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included) from rocprofv3 "
                            f"--pmc passes of this very command: {pmc.get('file')}") if pmc else pmc_why,
           "launches": prof["launches"], "avg_launch_us": 1e3 * net_ms / launches,
           "flops_per_launch": w_pass["gemm_flops"] / launches, "algorithmic_bytes_per_launch": w_pass["gemm_bytes"] / launches,
           "measured_on": measured_on,
           "note": "achieved / avg_launch_us: HIP event pair around every GEMM launch on its launch stream, minus event_pair_overhead_us "
                   "per launch (pairs around a kernel of known duration on the same stream); raw pair figures beside them; frac_wall: "
                   "the same FLOPs of the TIMED region over its wall clock (everything that is not a GEMM counts against it)",
           "event_pair_overhead_us": 1e3 * prof["pair_overhead_ms"], "avg_launch_us_raw_event_pairs": 1e3 * raw_ms / launches,
           "achieved_raw_event_pairs": ach_raw, "frac_raw_event_pairs": ach_raw / PEAK_F32_MATRIX_TFLOPS}
    if extra:
        rec.update(extra)
    return rec


class _SynthTokenizer:
    """The attributes of the reference's tokenizer the Lightning module reads (tokenizer_base.py:16-40)."""

    def __init__(self, ids, vocab):
        self.pad_token_idx, self.bos_token_idx, self.eos_token_idx, c_tok = ids
        self.n_tokens = vocab
        self.encoder_dict = {"c": c_tok}


def make_module(tta, sd, a, generation: str, tokenizer, layers: int = 4, **kw):
    mod = tta.VanillaEncoderDecoderTransformerLightning(
        src_tokenizer=tokenizer, tgt_tokenizer=tokenizer, embedding_dim=256, feedforward_dim=2048, num_encoder_layers=layers,
        num_decoder_layers=layers, num_heads=8, share_embeddings=True, generation=generation, max_len=a.max_len,
        draft_len=a.draft_len, report_prediction_time=False, **kw)
    mod.load_state_dict({"model." + k: v for k, v in sd.items()}, strict=True)
    mod.cuda()
    return mod


def predict_surface(tta, sd, timed, a, outs, raised, toks) -> dict:
    """The same K batches through the kept Lightning surface (src/model/lightning_model.py:209-243): a trainer stand-in that
    only calls on_predict_start -> predict_step per batch -> on_predict_end, i.e. what main.py's Trainer.predict does;
    predict_step serves the batches from windows decoded ahead (slot pools)."""
    PAD, BOS, EOS, C_TOK, V = toks
    from types import SimpleNamespace
    import contextlib
    mod = make_module(tta, sd, a, "greedy_speculative", _SynthTokenizer((PAD, BOS, EOS, C_TOK), V), n_drafts=a.n_drafts)
    os.environ["TTX_INFLIGHT"] = str(a.inflight)

    def loop(batches):
        loader = [{"src_tokens": b} for b in batches]
        mod.trainer = SimpleNamespace(datamodule=None, predict_dataloaders=loader)
        res = []
        with torch.inference_mode():
            mod.on_predict_start()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i, batch in enumerate(loader):
                try:
                    res.append(mod.predict_step(batch, i))
                except tta.ReferenceError_:
                    res.append(None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            mod.on_predict_end()
        return res, dt

    with contextlib.redirect_stdout(sys.stderr):      # the module prints its generator like the reference (lightning_model.py:73)
        loop(timed)                                       # sessions, workspaces and graphs of this module warm
        res, dt = loop(timed)
    ok = [i for i, o in enumerate(res) if o is not None]
    n = sum(int(timed[i].shape[0]) for i in ok)
    rec = {"value": n / dt, "unit": "reactions/s", "window_batches": mod.predict_window,
           "batches_served_from_look_ahead": mod._ahead.served if mod._ahead is not None else 0,
           "look_ahead_windows": mod._ahead.windows if mod._ahead is not None else 0,
           "look_ahead_decode_seconds": mod._ahead.decode_seconds if mod._ahead is not None else 0.0, "loop_seconds": dt,
           "identical_to_timed_outputs": sorted(set(range(len(timed))) - set(ok)) == sorted(raised)
           and all(torch.equal(res[i], outs[i]) for i in ok),
           "model_calls": mod.generator.model_calls_num}
    if mod.native is not None:
        mod.native.close()                # sessions (and their streams) go back before the next measurement builds its model
    return rec


def smiles_pipeline(tta, sd, a, src_rows, tgt_rows, token_batches) -> dict:
    """SMILES strings in -> CSV lines out: the reactions of the timed region as strings through
    NativeSmilesTokenizer.encode_batch (regex split + vocabulary + collate: tokenizer_smiles.py:8,34-39,
    seq2seq_wrappers.py:121-127), the module's predict_step (look-ahead, slot pools) and a writer that does what
    src/callbacks.py:49-64 does (ids -> strings with decode / decode_batch, one CSV row per reaction)."""
    from tools.synth import smiles_vocabulary, PAD, BOS, EOS
    from types import SimpleNamespace
    import contextlib
    import tempfile
    voc = smiles_vocabulary()
    inv = {v: k for k, v in voc.items()}
    tkz = tta.NativeSmilesTokenizer()
    tkz.assign_vocab(voc)
    src_lines = ["".join(inv[t] for t in r if t not in (PAD, BOS, EOS)) for r in src_rows]
    tgt_lines = ["".join(inv[t] for t in r if t not in (PAD, BOS, EOS)) for r in tgt_rows]
    bs = a.batch_size
    mod = make_module(tta, sd, a, "greedy_speculative", tkz, n_drafts=a.n_drafts)
    os.environ["TTX_INFLIGHT"] = str(a.inflight)

    class Writer:
        def __init__(self, path):
            self.path = path
            self.seconds = 0.0

        def write_on_batch_end(self, trainer, pl_module, prediction, batch_indices, batch, batch_idx, dataloader_idx):
            t0 = time.perf_counter()
            tk = pl_module.tgt_tokenizer
            p = prediction.cpu().numpy()
            with open(self.path, "a") as f:
                if f.tell() == 0:
                    print(",".join(["source", "target"] + [f"prediction_{i}" for i in range(1, p.shape[1] + 1)]), file=f)
                for i, (s, t) in enumerate(zip(batch["src_tokens"].cpu().numpy(), batch["tgt_tokens"].cpu().numpy())):
                    print(",".join([tk.decode(s), tk.decode(t)] + tk.decode_batch(p[i])), file=f)
            self.seconds += time.perf_counter() - t0

    def run():
        with tempfile.TemporaryDirectory() as td:
            w = Writer(os.path.join(td, "predictions.csv"))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            # data side: strings -> padded id batches on the device (what the DataModule's dataset + collate produce)
            loader = []
            for i in range(0, len(src_lines), bs):
                loader.append({"src_tokens": torch.from_numpy(tkz.encode_batch(src_lines[i:i + bs])).cuda(),
                               "tgt_tokens": torch.from_numpy(tkz.encode_batch(tgt_lines[i:i + bs])).cuda()})
            t_tok = time.perf_counter() - t0
            mod.trainer = SimpleNamespace(datamodule=None, predict_dataloaders=loader)
            with torch.inference_mode():
                mod.on_predict_start()
                for i, batch in enumerate(loader):
                    try:
                        p = mod.predict_step(batch, i)
                    except tta.ReferenceError_:
                        p = torch.full((batch["src_tokens"].shape[0], 1, a.max_len), PAD, dtype=torch.int64, device="cuda")
                    w.write_on_batch_end(mod.trainer, mod, p, None, batch, i, 0)
                torch.cuda.synchronize()
                mod.on_predict_end()
            dt = time.perf_counter() - t0
            n_lines = sum(1 for _ in open(w.path)) - 1
        return dt, t_tok, w.seconds, n_lines, loader

    with contextlib.redirect_stdout(sys.stderr):
        run()
        dt, t_tok, t_write, n_lines, loader = run()
    same_tokens = len(loader) == len(token_batches) and all(torch.equal(b["src_tokens"], t) for b, t in zip(loader, token_batches))
    n = len(src_lines)
    if mod.native is not None:
        mod.native.close()
    return {"value": n / dt, "unit": "reactions/s (SMILES strings in -> CSV rows out)", "reactions": n, "csv_rows": n_lines,
            "seconds": {"total": dt, "tokenize_and_collate": t_tok, "write_csv_incl_detokenize": t_write,
                        "decode_on_gpu_and_rest": dt - t_tok - t_write},
            "host_share": (t_tok + t_write) / dt,
            "tokenizer_lines_per_s": 2 * n / t_tok, "writer_rows_per_s": n / max(t_write, 1e-9),
            "tokenized_batches_equal_the_token_level_inputs": bool(same_tokens),
            "what": "C++ tokenizer (ttx_tokenizer_encode_batch) for source and target lines, look-ahead predict_step, "
                    "PredictionWriter restatement with ttx_tokenizer_decode; one host thread, CSV to a temporary file"}


# ------------------------------------------------------------------------------------------------------------------------
def measure_c2(ctx: Ctx, a, tta) -> dict:
    from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK, V
    rank, world, dev, dist = ctx.rank, ctx.world, ctx.dev, ctx.dist
    # ---- weights: rank 0 trains/loads and packs them into its HBM blob; every other rank receives that blob with ONE RCCL
    # broadcast straight into its own (empty) model's blob (dist.broadcast_model)
    sd = get_weights(a.train_steps, dev, info=ctx.setup) if rank == 0 else None
    model = tta.dist.broadcast_model(sd, 8, PAD, ctx.local_rank, dist)      # one RCCL broadcast of the packed blob into HBM
    cfg = {"d": model.emb_dim, "F": model.ff_dim, "V": model.tgt_vocab_size, "Le": model.num_enc_layers, "Ld": model.num_dec_layers}
    # ---- data: every rank takes its own contiguous shard of the synthetic test set (seed 123456)
    per_rank = (a.steps + a.warmup) * a.batch_size
    src_all, tgt_all = SynthReactions(123456, "mit").dataset(per_rank * world)
    mine = src_all[rank * per_rank:(rank + 1) * per_rank]
    mine_tgt = tgt_all[rank * per_rank:(rank + 1) * per_rank]
    dev_batches = [torch.from_numpy(b).to(dev) for b in batches(mine, a.batch_size)]
    warm, timed = dev_batches[:a.warmup], dev_batches[a.warmup:]
    rows_sched = a.schedule == "rows" and a.inflight > 1

    def make_gen(m, n_drafts=None):
        return tta.TranslationInferenceGreedySpeculative(m, a.max_len, a.draft_len, n_drafts or a.n_drafts, PAD, BOS, EOS, C_TOK)

    def decode(gen, bs):
        """The timed call.  A batch on which the reference itself raises (a row finishing at a width beyond max_len) is skipped
        and not counted: its rows are reported as all-PAD and listed under "batches_reference_raises"."""
        if a.inflight > 1:
            outs = gen.generate_many(bs, in_flight=a.inflight, reorder=rows_sched, on_error="skip")
            return outs, list(gen.last_failed_batches), gen
        outs, raised = [], []
        for i, b in enumerate(bs):
            try:
                outs.append(gen.generate(b))
            except tta.ReferenceError_:
                outs.append(None)
                raised.append(i)
        return outs, raised, gen

    log("model ready; warmup")
    g0 = make_gen(model)
    for b in warm:
        g0.generate(b)
    decode(make_gen(model), timed)          # sessions, workspaces and graphs at the timed region's own layout
    log("warmup done")
    sorted_s, (outs, raised, gen), run_s = ctx.timed(lambda: decode(make_gen(model), timed), a.repeats)
    elapsed = median_seconds(sorted_s)
    log("timed region done", sorted_s)

    # ---- prediction gather: one collective at the end
    outs = [o if o is not None else torch.full((timed[i].shape[0], 1, a.max_len), PAD, dtype=torch.int64, device=dev)
            for i, o in enumerate(outs)]
    preds = torch.cat([o[:, 0, :] for o in outs])
    gathered = tta.dist.gather_predictions(preds.unsqueeze(1), world * preds.shape[0], dist)
    if rank == 0:
        assert gathered.shape[0] == world * preds.shape[0]
        if os.environ.get("TTX_DUMP_PREDICTIONS"):        # tests/test_gpu_dist.py compares them with a single-process decode
            np.save(os.environ["TTX_DUMP_PREDICTIONS"], gathered.cpu().numpy())
    counted = tta.dist.sum_counters({"reactions": sum(int(timed[i].shape[0]) for i in range(len(timed)) if i not in raised),
                                     "raised": len(raised)}, dev, dist)
    n_reactions = int(counted["reactions"])
    stats = dict(gen.stats_total)
    stats["model_calls"] = gen.model_calls_num
    finished = int((preds == EOS).any(dim=1).sum())
    per_rank_reactions = n_reactions / world

    line = {
        "metric": "reactions/sec (SMILES decoded), greedy speculative", "value": n_reactions / elapsed,
        "unit": "reactions/s", "n_gpus": world, "steps": len(timed), "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / max(1, len(timed)), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": (n_reactions / elapsed) / BASELINE_REACTIONS_PER_S, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"USPTO-MIT-shaped synthetic SMILES, greedy speculative draft_len={a.draft_len} "
                               f"n_drafts={a.n_drafts} bs={a.batch_size} max_len={a.max_len}, d=256 8h FFN2048 4+4 fp32, "
                               f"weights trained {a.train_steps} steps on the synthetic task",
                   "baseline_config": "c2", "reactions": n_reactions, "parallelism": f"test-set shards x{world}, no per-step collective",
                   "batches_in_flight_per_gpu": a.inflight,
                   "schedule": ("rows of the K given batches in slot pools on the device, reference loop replayed per given batch "
                                "(outputs and model_calls identical to per-batch generate)") if rows_sched else "batches decoded as given"},
        "repeats": dict(spread(sorted_s, n_reactions), seconds_in_run_order=run_s,
                        protocol="the timed region run this many times, each bracketed by barrier + synchronize; value = median"),
        "batches_reference_raises": int(counted["raised"]),
        "model_calls": stats["model_calls"], "rows_finished_rank0": finished, "rows_rank0": int(preds.shape[0]),
        "accepted_per_step_per_row": stats["accepted_tokens"] / max(1, stats["produced_tokens"] - stats["accepted_tokens"]),
        "device_ms_encode_rank0": stats["encode_ms"], "device_ms_decode_rank0": stats["decode_ms"],
    }
    if rank != 0:
        return line
    # work the device executed: under the row schedule that is the slot pools', not the given batches'
    dstats = dict(stats["device"], encode_ms=stats["encode_ms"], decode_ms=stats["decode_ms"]) if rows_sched else stats

    def c2_work(ds, n_batches):
        return work(cfg, ds["verified_positions"], ds["src_tokens_padded"], ds["model_calls"], ds.get("batches", n_batches),
                    ds["kv_prefix_positions"] + ds["src_positions"], ds["produced_tokens"])

    w_timed = c2_work(dstats, len(timed))
    line["hbm_algorithmic"] = {"bytes_per_reaction": w_timed["bytes"] / max(1.0, per_rank_reactions),
                               "achieved_GBs": w_timed["bytes"] / elapsed / 1e9,
                               "frac_of_peak": w_timed["bytes"] / elapsed / 1e9 / PEAK_HBM_GBS,
                               "note": "KV-cached algorithm's bytes (SURVEY.md §8(d)) over the timed wall clock: the path is bound by the "
                                       "fp32 matrix pipe and launch latency, not by HBM — see roofline"}
    if rows_sched:
        line["device_model_calls"] = stats["device"]["model_calls"]
        line["device_src_tokens_padded"] = stats["device"]["src_tokens_padded"]
        line["config"]["pools"] = {"slots_per_pool": getattr(gen, "last_group_size", None)}
    if a.timed_only:
        model.close()
        return line
    # ---- beside the headline: N = 1 ("N=1 reported alongside"), the batches as given, one at a time
    decode(make_gen(model, 1), timed)
    s1, (o1, r1, g1), _ = ctx.timed(lambda: decode(make_gen(model, 1), timed), 1, local=True)
    line["n_drafts_1"] = {"value": sum(int(timed[i].shape[0]) for i in range(len(timed)) if i not in r1) / s1[0], "unit": "reactions/s",
                          "model_calls": g1.model_calls_num,
                          "rows_identical_to_n_drafts_3": int(sum(int((x[:, 0] == y[:, 0]).all(dim=1).sum()) for x, y in zip(o1, outs)
                                                                if x is not None)),
                          "rows": int(preds.shape[0])}
    if a.inflight > 1:
        gq = make_gen(model)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        seq = []
        for b in timed:
            try:
                seq.append(gq.generate(b))
            except tta.ReferenceError_:
                seq.append(None)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        ok1 = [i for i, o in enumerate(seq) if o is not None]
        line["one_batch_at_a_time"] = {"value": sum(int(timed[i].shape[0]) for i in ok1) / dt1, "unit": "reactions/s",
                                       "identical_to_timed_outputs": sorted(set(range(len(timed))) - set(ok1)) == sorted(raised)
                                       and all(torch.equal(seq[i], outs[i]) for i in ok1)}
    if rows_sched:
        make_gen(model).generate_many(timed, in_flight=a.inflight, on_error="skip")
        g2 = make_gen(model)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        given = g2.generate_many(timed, in_flight=a.inflight, on_error="skip")
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        ok2 = [i for i, o in enumerate(given) if o is not None]
        line["batches_as_given_in_flight"] = {"value": sum(int(timed[i].shape[0]) for i in ok2) / dt2, "unit": "reactions/s",
                                              "identical_to_row_scheduled_outputs": sorted(set(range(len(timed))) - set(ok2)) == sorted(raised)
                                              and all(torch.equal(given[i], outs[i]) for i in ok2),
                                              "model_calls": g2.model_calls_num}
    # every further leg builds a model of its own: this one's sessions, workspaces and streams go back first (a process's later
    # streams share hardware queues less evenly than its first ones, DESIGN.md §4.3 — each leg should see what a fresh process sees)
    model.close()
    torch.cuda.empty_cache()
    line["predict_step_surface"] = predict_surface(tta, sd, timed, a, outs, raised, (PAD, BOS, EOS, C_TOK, V))
    if not a.no_sub_records:
        n0 = a.warmup * a.batch_size
        line["smiles_pipeline"] = smiles_pipeline(tta, sd, a, mine[n0:], mine_tgt[n0:], timed)
    if not a.no_profile:
        # the dominant kernel family (fp32 MFMA GEMMs) on the timed region's OWN layout: the same generate_many call (same
        # pools, same number of sessions and streams) on a model whose sessions bracket every GEMM launch with a HIP event pair
        os.environ["TTX_PROFILE_GEMM"] = "1"
        pm = tta.NativeTransformer(sd, num_heads=8, pad_token_idx=PAD, device=ctx.local_rank)
        os.environ.pop("TTX_PROFILE_GEMM")
        decode(make_gen(pm), timed)
        pm.kernel_profile()
        _, _, pg = decode(make_gen(pm), timed)
        prof = pm.kernel_profile()
        pstats = dict(pg.stats_total)
        pstats["model_calls"] = pg.model_calls_num
        if rows_sched:
            pstats = dict(pstats["device"])
        pmc, pmc_why = pmc_traffic_for({"config": "c2", "steps": len(timed), "warmup": a.warmup, "schedule": a.schedule, "inflight": a.inflight,
                                        "batch_size": a.batch_size, "n_drafts": a.n_drafts, "draft_len": a.draft_len, "max_len": a.max_len})
        line["roofline"] = roofline_record(
            "k_gemm24<NT> / k_gemm2<NT> / k_gemm3 (fp32 v_mfma_f32_32x32x2_f32 GEMM family, canonical slice order: 128x64 / 64x64 tiles "
            "picked per launch from the live row count, one wave per slice for steps of few rows; every GEMM launch of the pass: "
            "encoder, cross K/V, verify steps)", prof, c2_work(pstats, len(timed)), w_timed, elapsed,
            "a further run of the timed region's own call (same slot pools, sessions and streams) right after it, on sessions that launch "
            "eagerly and bracket every GEMM launch with a HIP event pair on its launch stream; the pools of this pass run one after "
            "another (in the timed region they overlap), so that a pair times its own launch and not other pools' kernels", pmc, pmc_why)
        pm.close()
        log("profile pass done")
    if world == 1 and not a.no_cpu_baseline:
        from oracle.model import OracleTransformer, config_from_state
        from oracle.decoding import GreedySpeculativeOracle
        cores = usable_cores()
        torch.set_num_threads(cores)
        log("cpu baseline on", cores, "threads")
        om = OracleTransformer(config_from_state(sd, 8, PAD), sd)
        og = GreedySpeculativeOracle(om, a.max_len, a.draft_len, a.n_drafts, PAD, BOS, EOS, C_TOK)
        sample_idx = [i for i in range(len(timed)) if i not in raised][:a.cpu_batches or 3]
        sample = [timed[i] for i in sample_idx]
        with torch.inference_mode():
            t1 = time.perf_counter()
            cpu_out = [og.generate(b.cpu()) for b in sample]
            cpu_s = time.perf_counter() - t1
        same = sum(int(torch.equal(c[:, 0], outs[i][:, 0].cpu())) for c, i in zip(cpu_out, sample_idx))
        rows_same = sum(int((c[:, 0] == outs[i][:, 0].cpu()).all(dim=1).sum()) for c, i in zip(cpu_out, sample_idx))
        n_cpu = len(sample) * a.batch_size
        line["cpu_baseline"] = {"value": n_cpu / cpu_s, "unit": "reactions/s", "cores": cores, "kind": "port",
                                "sample": f"first {len(sample)} timed batch(es) = {n_cpu} reactions of the same workload, "
                                          f"oracle/ (full-prefix recompute like the reference), torch {torch.__version__} fp32",
                                "seconds": cpu_s, "model_calls": og.model_calls_num}
        line["parity"] = {"rows_token_identical_to_oracle": rows_same, "rows_checked": n_cpu, "batches_identical": same}
    return line


# ------------------------------------------------------------------------------------------------------------------------
def measure_beam(ctx: Ctx, a, tta, name: str, steps: int, warmup: int, full: bool) -> dict:
    """configs[2] / configs[3]: beam-search speculative decoding, whole loop native.  A step = one given batch; the K given
    batches go through generate_many: batch pools (whole batches admitted as slots free up, one verify step per iteration for
    every live candidate of every batch in the pool; outputs and counters per given batch are those of one-at-a-time calls,
    tests/test_gpu_beam_pool.py)."""
    from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK, V
    bc = BEAM_CONFIGS[name]
    rank, world, dev, dist = ctx.rank, ctx.world, ctx.dev, ctx.dist
    bs = a.batch_size if a.config == name and a.batch_size else bc["bs"]
    sd = get_weights(a.train_steps, dev, bc["kind"], bc["layers"], ctx.setup) if rank == 0 else None
    model = tta.dist.broadcast_model(sd, 8, PAD, ctx.local_rank, dist)
    cfg = {"d": model.emb_dim, "F": model.ff_dim, "V": model.tgt_vocab_size, "Le": model.num_enc_layers, "Ld": model.num_dec_layers}
    assert cfg["Le"] == bc["layers"] and cfg["Ld"] == bc["layers"]
    per_rank = (steps + warmup) * bs
    src_all, _ = SynthReactions(123456, bc["kind"]).dataset(per_rank * world)
    mine = src_all[rank * per_rank:(rank + 1) * per_rank]
    dev_batches = [torch.from_numpy(b).to(dev) for b in batches(mine, bs)]
    warm, timed = dev_batches[:warmup], dev_batches[warmup:]
    K, N, D, L = bc["n_best"], (a.n_drafts if a.config == name and a.n_drafts else bc["N"]), a.draft_len, a.max_len
    pooled = a.schedule == "rows" and a.inflight > 1

    def make_gen(m, smart):
        return tta.TranslationInferenceBeamSearchSpeculative(m, L, K, D, N, V, bool(smart), PAD, BOS, EOS, C_TOK, max_steps=4 * L)

    def run(gen, batch_list, inflight, pool):
        if inflight > 1:
            return gen.generate_many(batch_list, in_flight=inflight, pool=pool), gen
        return [gen.generate(b) for b in batch_list], gen

    smart = bool(a.smart) if a.config == name else False
    g0 = make_gen(model, smart)
    for b in warm:
        g0.generate(b)
    run(make_gen(model, smart), timed, a.inflight, pooled)              # pools, workspaces and graphs at the timed layout
    log(name, "warmup done")
    sorted_s, (outs, gen), run_s = ctx.timed(lambda: run(make_gen(model, smart), timed, a.inflight, pooled), a.repeats)
    elapsed = median_seconds(sorted_s)
    # prediction gather (C2): hypotheses padded to max_len, one collective at the end
    flat = torch.cat([torch.nn.functional.pad(o, (0, L - o.shape[2]), value=PAD) for o in outs])
    gathered = tta.dist.gather_predictions(flat, world * flat.shape[0], dist)
    if rank == 0:
        assert gathered.shape[0] == world * flat.shape[0]
    counted = tta.dist.sum_counters({"reactions": sum(int(b.shape[0]) for b in timed)}, dev, dist)
    n_reactions = int(counted["reactions"])
    value = n_reactions / elapsed
    rec = {
        "metric": "reactions/sec (SMILES decoded), beam-search speculative", "value": value, "unit": "reactions/s", "n_gpus": world,
        "steps": len(timed), "warmup": warmup, "ms_per_step": 1e3 * elapsed / max(1, len(timed)), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": value / bc["published"], "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{bc['name']} bs={bs} n_drafts={N} draft_len={D} max_len={L} smart_drafts_mode={smart}, d=256 8h "
                               f"FFN2048 {bc['layers']}+{bc['layers']} fp32, weights trained {a.train_steps} steps on the synthetic task",
                   "baseline_config": name, "reactions": n_reactions, "parallelism": f"test-set shards x{world}, no per-step collective",
                   "batches_in_flight_per_gpu": a.inflight,
                   "schedule": "batch pools on the device (whole batches admitted as slots free up), per-batch width / calls / counters "
                               "replayed from per-source traces" if pooled else "batches decoded as given"},
        "repeats": dict(spread(sorted_s, n_reactions), seconds_in_run_order=run_s),
        "model_calls": gen.model_calls_num, "acceptance_rate": gen.accepted_tokens_num / max(1, gen.produced_non_pad_tokens),
        "device_iterations_rank0": gen.stats_total.get("device_model_calls"),
        "top1_rows_with_eos_rank0": int((flat[:, 0] == EOS).any(dim=1).sum()), "rows_rank0": int(flat.shape[0]),
    }
    if rank != 0:
        return rec

    def beam_work(g):
        st = g.stats_total
        return work(cfg, st["verified_positions"], st["src_tokens_padded"], st.get("device_model_calls") or g.model_calls_num,
                    st["batches"], st["kv_prefix_positions"] + st["src_positions"], g.produced_non_pad_tokens)

    w_timed = beam_work(gen)
    rec["hbm_algorithmic"] = {"bytes_per_reaction": w_timed["bytes"] / max(1.0, n_reactions / world),
                              "achieved_GBs": w_timed["bytes"] / elapsed / 1e9, "frac_of_peak": w_timed["bytes"] / elapsed / 1e9 / PEAK_HBM_GBS}
    if a.timed_only:
        model.close()
        return rec
    if full:
        if pooled:
            run(make_gen(model, smart), timed, a.inflight, False)
            s2, (given, gg), _ = ctx.timed(lambda: run(make_gen(model, smart), timed, a.inflight, False), 1, local=True)
            rec["batches_as_given_in_flight"] = {"value": n_reactions / world / s2[0], "unit": "reactions/s",
                                                 "identical_to_pooled_outputs": all(torch.equal(x, y) for x, y in zip(given, outs))
                                                 and gg.model_calls_num == gen.model_calls_num}
        s1, (seq, g1), _ = ctx.timed(lambda: run(make_gen(model, smart), timed, 1, False), 1, local=True)
        rec["one_batch_at_a_time"] = {"value": n_reactions / world / s1[0], "unit": "reactions/s",
                                      "identical_to_timed_outputs": all(torch.equal(x, y) for x, y in zip(seq, outs))
                                      and g1.model_calls_num == gen.model_calls_num}
    run(make_gen(model, not smart), timed, a.inflight, pooled)
    s3, (other, g3), _ = ctx.timed(lambda: run(make_gen(model, not smart), timed, a.inflight, pooled), 1, local=True)
    rec["other_draft_mode"] = {"smart_drafts_mode": not smart, "value": n_reactions / world / s3[0], "unit": "reactions/s",
                               "model_calls": g3.model_calls_num,
                               "top1_identical_to_timed_run": sum(int(torch.equal(x[:, 0, :min(x.shape[2], y.shape[2])],
                                                                                  y[:, 0, :min(x.shape[2], y.shape[2])]))
                                                                  for x, y in zip(other, outs)), "batches": len(outs)}
    model.close()                         # the profiling model below takes over this one's sessions' streams
    torch.cuda.empty_cache()
    if not a.no_profile:
        os.environ["TTX_PROFILE_GEMM"] = "1"
        pm = tta.NativeTransformer(sd, num_heads=8, pad_token_idx=PAD, device=ctx.local_rank)
        os.environ.pop("TTX_PROFILE_GEMM")
        run(make_gen(pm, smart), timed, a.inflight, pooled)
        pm.kernel_profile()
        _, pg = run(make_gen(pm, smart), timed, a.inflight, pooled)
        prof = pm.kernel_profile()
        pmc, pmc_why = pmc_traffic_for({"config": name, "steps": len(timed), "warmup": warmup, "schedule": a.schedule, "inflight": a.inflight,
                                        "batch_size": bs, "n_drafts": N, "draft_len": D, "max_len": L, "smart": int(smart)})
        rec["roofline"] = roofline_record(
            "k_gemm24<NT> / k_gemm2<NT> / k_gemm3 (fp32 v_mfma_f32_32x32x2_f32 GEMM family, canonical slice order) — every GEMM launch "
            "of the pass: encoders of the admitted batches, cross K/V, the pools' verify steps", prof, beam_work(pg), w_timed, elapsed,
            "a further run of the timed region's own call (same batch pools, sessions and streams) on sessions that launch eagerly and "
            "bracket every GEMM launch with a HIP event pair on its launch stream; the pools of this pass run one after another (in the "
            "timed region they overlap), so that a pair times its own launch and not other pools' kernels", pmc, pmc_why,
            {"gemm_share_of_device_time": prof["gemm_ms"] / max(1e-9, pg.stats_total["encode_ms"] + pg.stats_total["decode_ms"])})
        pm.close()
    if world == 1 and not a.no_cpu_baseline:
        from oracle.model import OracleTransformer, config_from_state
        from oracle.spec_beam import BeamSearchSpeculativeOracle
        cores = usable_cores()
        torch.set_num_threads(cores)
        om = OracleTransformer(config_from_state(sd, 8, PAD), sd)
        og = BeamSearchSpeculativeOracle(om, L, K, D, N, V, smart, PAD, BOS, EOS, C_TOK, max_steps=4 * L)
        sample = timed[:max(1, a.cpu_batches or 1)]
        with torch.inference_mode():
            t1 = time.perf_counter()
            cpu_out = [og.generate(b.cpu()) for b in sample]
            cpu_s = time.perf_counter() - t1
        top1 = all_ranks = total = 0
        for c_, o in zip(cpu_out, outs):
            o = o.cpu()
            w = max(c_.shape[2], o.shape[2])
            c_ = torch.nn.functional.pad(c_, (0, w - c_.shape[2]), value=PAD)
            o = torch.nn.functional.pad(o, (0, w - o.shape[2]), value=PAD)
            top1 += int((c_[:, 0] == o[:, 0]).all(dim=1).sum())
            all_ranks += int((c_ == o).all(dim=2).sum())
            total += c_.shape[0] * c_.shape[1]
        n_cpu = sum(int(b.shape[0]) for b in sample)
        rec["cpu_baseline"] = {"value": n_cpu / cpu_s, "unit": "reactions/s", "cores": cores, "kind": "port",
                               "sample": f"first {len(sample)} timed batch(es) = {n_cpu} reactions of the same workload, oracle/ "
                                         f"(full-prefix recompute like the reference), torch {torch.__version__} fp32",
                               "seconds": cpu_s, "model_calls": og.model_calls_num}
        rec["parity"] = {"top1_rows_token_identical_to_oracle": top1, "rows_checked": n_cpu,
                         "hypotheses_token_identical_to_oracle": all_ranks, "hypotheses_checked": total}
    return rec


# ------------------------------------------------------------------------------------------------------------------------
def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, torch.distributed.run) from this
    process BEFORE it touches a GPU, and return their exit code.  Never prints an n_gpus: 1 line for --gpus N > 1."""
    n_dev = torch.cuda.device_count()              # counts devices without initialising the GPU in this process
    if n_dev < a.gpus and os.environ.get("TTX_SHARE_GPU") != "1":
        print(f"bench.py: --gpus {a.gpus} but {n_dev} GPU(s) visible (TTX_SHARE_GPU=1 rehearses the N>1 path on fewer)", file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT") or str(29500 + (os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < a.gpus:
        env.setdefault("TTX_DIST_BACKEND", "gloo")      # RCCL refuses two ranks on one device
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    ctx = Ctx()
    import translation_transformer_amd as tta
    if a.config in BEAM_CONFIGS:
        bc = BEAM_CONFIGS[a.config]
        line = measure_beam(ctx, a, tta, a.config, a.steps or bc["steps"], a.warmup, full=True)
    else:
        a.steps = a.steps or 256
        a.batch_size = a.batch_size or 32
        a.n_drafts = a.n_drafts or 3
        line = measure_c2(ctx, a, tta)
        if ctx.world == 1 and not a.no_sub_records:
            # the other two single-GPU BASELINE configs, so that one command shows all three (their own batch counts and warm-up)
            for name in ("c3", "c4"):
                log("sub-record", name)
                line[name] = measure_beam(ctx, a, tta, name, BEAM_CONFIGS[name]["steps"], 2, full=False)
    line["setup"] = ctx.setup
    if ctx.rank == 0:
        print(json.dumps(line))
    if ctx.dist:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
