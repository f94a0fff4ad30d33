#!/usr/bin/env python3
"""Benchmark of the MI355X-native greedy-speculative decoding path (BASELINE.json configs[1]):
reactions/sec at bs=32, draft_len=10, n_drafts=3, max_len=200 on USPTO-MIT-shaped synthetic SMILES.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one batch of 32 reactions through ``generate`` (encoder + the whole verify loop).  Inputs are
resident in HBM before the timed region; every rank decodes its own shard of the synthetic test set (weak
scaling, no data-path collective); weights are broadcast once from rank 0 and predictions gathered once at
the end over RCCL.  Rank 0 prints ONE JSON line.

Weights: there is no checkpoint offline, so a full-size (d=256, 8 heads, FFN 2048, 4+4) model is trained
on the synthetic task by tools/train_synth.py for a fixed step budget (cached under /tmp for later runs
on the same box).  That is set-up, outside every timed region.
"""
from __future__ import annotations

import argparse
import json
import os
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=("c2", "c3", "c4"), default="c2",
                    help="BASELINE.json configs[1] (greedy speculative bs=32, the headline), configs[2] (MIT beam speculative "
                         "n_best=5 bs=4 N=7) or configs[3] (50K 6+6 beam speculative n_best=10 bs=8 N=2)")
    ap.add_argument("--steps", type=int, default=None, help="timed batches (default 256 for c2, 32 for c3/c4)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--draft-len", type=int, default=10)
    ap.add_argument("--n-drafts", type=int, default=None)
    ap.add_argument("--smart", type=int, default=0, help="c3/c4: smart_drafts_mode of the timed run (the other mode is reported beside it)")
    ap.add_argument("--max-len", type=int, default=200)
    ap.add_argument("--train-steps", type=int, default=int(os.environ.get("TTX_TRAIN_STEPS", "1500")))
    ap.add_argument("--cpu-batches", type=int, default=3, help="batches of the workload timed on the host cores")
    ap.add_argument("--schedule", choices=("rows", "batches"), default=os.environ.get("TTX_SCHEDULE", "rows"),
                    help="rows: regroup the rows of the given batches by length (exact replay per batch); batches: as given")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("TTX_INFLIGHT", "8")),
                    help="batches decoded concurrently per GPU (1 = the reference's one-batch-at-a-time loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="skip the comparison passes (as-given, one at a time, event profile, CPU baseline): for rocprofv3 runs")
    a = ap.parse_args()
    beam = BEAM_CONFIGS.get(a.config)
    if a.steps is None:
        a.steps = 32 if beam else 256
    if a.batch_size is None:
        a.batch_size = beam["bs"] if beam else 32
    if a.n_drafts is None:
        a.n_drafts = beam["N"] if beam else 3
    return a


# BASELINE.json configs[2] / configs[3]: generator settings of the reference's own grid optimum for that batch size
# (results_grid_search/results_product_500_beam_search_speculative_bs_4_report.txt:31 -> 7.42 reactions/s;
#  results_retro_500_beam_search_speculative_bs_8_nbest_10_report.txt:5 -> 6.12 reactions/s; scripts/product_prediction.sh:197-198,
#  scripts/single_step_retrosynthesis.sh:166-174; model depth configs/cfg_standard_*:90-103)
BEAM_CONFIGS = {
    "c3": dict(kind="mit", layers=4, bs=4, n_best=5, N=7, published=7.42,
               name="USPTO-MIT-shaped synthetic SMILES, beam-search speculative n_best=5"),
    "c4": dict(kind="50k", layers=6, bs=8, n_best=10, N=2, published=6.12,
               name="USPTO-50K-shaped synthetic SMILES (retrosynthesis), beam-search speculative n_best=10"),
}


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
                info.update(weights="cache", path=cand)
            return torch.load(cand, weights_only=True, map_location="cpu")
    from tools.train_synth import train
    t0 = time.perf_counter()
    sd = train(kind, steps=train_steps, n_enc=layers, n_dec=layers, device=device, verbose=os.environ.get("TTX_BENCH_VERBOSE") == "1")
    if info is not None:
        info.update(weights="trained in this run", train_steps=train_steps, train_seconds=round(time.perf_counter() - t0, 1))
    try:
        torch.save(sd, path)
        extra = os.environ.get("TTX_SAVE_WEIGHTS")
        if extra:
            torch.save(sd, extra)
    except OSError:
        pass
    return sd


def flops_and_bytes(cfg: dict, stats: dict, B_total_src_tokens: int, n_batches: int) -> dict:
    """Algorithmic work of the KV-cached algorithm: dense FLOPs of every GEMM launch and
    HBM bytes (weights once per step / per batch, K/V cache reads and writes)."""
    d, F, V, Le, Ld = cfg["d"], cfg["F"], cfg["V"], cfg["Le"], cfg["Ld"]
    pos = stats["verified_positions"]
    dec_dense_per_pos = Ld * (12 * d * d + 4 * d * F) + 2 * d * V          # qkv+o+cq+co = 6 d^2 MAC -> 12 d^2 FLOP
    enc_dense_per_tok = Le * (8 * d * d + 4 * d * F)
    cross_kv_per_tok = Ld * 4 * d * d
    gemm_flops = pos * dec_dense_per_pos + B_total_src_tokens * (enc_dense_per_tok + cross_kv_per_tok)
    P_e = 4 * d * d + 4 * d + 2 * d * F + F + d + 4 * d
    P_d = 2 * (4 * d * d + 4 * d) + 2 * d * F + F + d + 6 * d
    W_enc = 4 * (Le * P_e + 2 * d + V * d)
    W_dec = 4 * (Ld * P_d + 2 * d + d * V + V)
    steps = stats["model_calls"]
    kv_read = (stats["kv_prefix_positions"] + stats["src_positions"]) * 2 * d * 4 * Ld
    kv_write = stats["produced_tokens"] * 2 * d * 4 * Ld
    bytes_total = n_batches * W_enc + steps * W_dec + kv_read + kv_write
    # operand bytes of the GEMM launches alone (activations in + out per position / source token, weights per pass)
    dec_io = 4 * (Ld * ((d + 3 * d) + 3 * (d + d) + (d + F) + (F + 2 * d)) + (d + V))
    enc_io = 4 * (Le * ((d + 3 * d) + (d + d) + (d + F) + (F + d)) + (d + Ld * 2 * d))
    gemm_bytes = pos * dec_io + B_total_src_tokens * enc_io + steps * W_dec + n_batches * (W_enc + 4 * Ld * 2 * d * d)
    return {"gemm_flops": float(gemm_flops), "bytes": float(bytes_total), "gemm_bytes": float(gemm_bytes)}


def pmc_traffic_for(key: dict):
    """HBM-side traffic per GEMM launch for exactly this command, from the committed rocprofv3 --pmc passes
    (profiles/gemm_pmc_traffic.json: one entry per profiled command, written by tools/pmc_summary.py).  Returns
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


class _SynthTokenizer:
    """The attributes of the reference's tokenizer the Lightning module reads (tokenizer_base.py:16-40)."""

    def __init__(self, ids, vocab):
        self.pad_token_idx, self.bos_token_idx, self.eos_token_idx, c_tok = ids
        self.n_tokens = vocab
        self.encoder_dict = {"c": c_tok}


def predict_surface(tta, sd, timed, warm, a, outs, raised, toks) -> dict:
    PAD, BOS, EOS, C_TOK, V = toks
    from types import SimpleNamespace
    tk = _SynthTokenizer((PAD, BOS, EOS, C_TOK), V)
    mod = tta.VanillaEncoderDecoderTransformerLightning(
        src_tokenizer=tk, tgt_tokenizer=tk, embedding_dim=256, feedforward_dim=2048, num_encoder_layers=4, num_decoder_layers=4,
        num_heads=8, share_embeddings=True, generation="greedy_speculative", max_len=a.max_len, n_drafts=a.n_drafts,
        draft_len=a.draft_len, report_prediction_time=False)
    mod.load_state_dict({"model." + k: v for k, v in sd.items()}, strict=True)
    mod.cuda()
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

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the module prints its generator like the reference (lightning_model.py:73)
        loop((warm * (-(-len(timed) // max(1, len(warm)))))[:len(timed)])    # sessions, workspaces and graphs of this module warm
        res, dt = loop(timed)
    ok = [i for i, o in enumerate(res) if o is not None]
    n = sum(int(timed[i].shape[0]) for i in ok)
    return {"value": n / dt, "unit": "reactions/s", "window_batches": mod.predict_window,
            "batches_served_from_look_ahead": mod._ahead.served if mod._ahead is not None else 0,
            "look_ahead_windows": mod._ahead.windows if mod._ahead is not None else 0,
            "look_ahead_decode_seconds": mod._ahead.decode_seconds if mod._ahead is not None else 0.0, "loop_seconds": dt,
            "row_schedule_ran": "device" in mod.generator.stats_total,
            "identical_to_timed_outputs": sorted(set(range(len(timed))) - set(ok)) == sorted(raised)
            and all(torch.equal(res[i], outs[i]) for i in ok),
            "model_calls": mod.generator.model_calls_num}


def beam_work(cfg: dict, st: dict, model_calls: int, positions_key: str) -> dict:
    """Algorithmic work of the KV-cached beam-speculative loop (SURVEY.md §8(d) formulas): GEMM FLOPs over the verified
    positions and the encoder tokens; HBM bytes = weights once per iteration / per batch + K/V reads + K/V writes."""
    d, F, V, Le, Ld = cfg["d"], cfg["F"], cfg["V"], cfg["Le"], cfg["Ld"]
    pos, src_tok = st[positions_key], st["src_tokens_padded"]
    dec_dense_per_pos = Ld * (12 * d * d + 4 * d * F) + 2 * d * V
    enc_dense_per_tok = Le * (8 * d * d + 4 * d * F)
    cross_kv_per_tok = Ld * 4 * d * d
    gemm_flops = pos * dec_dense_per_pos + src_tok * (enc_dense_per_tok + cross_kv_per_tok)
    P_e = 4 * d * d + 4 * d + 2 * d * F + F + d + 4 * d
    P_d = 2 * (4 * d * d + 4 * d) + 2 * d * F + F + d + 6 * d
    W_enc = 4 * (Le * P_e + 2 * d + V * d)
    W_dec = 4 * (Ld * P_d + 2 * d + d * V + V)
    kv_read = (st["kv_prefix_positions"] + st["src_positions"]) * 2 * d * 4 * Ld
    kv_write = st["produced"] * 2 * d * 4 * Ld
    bytes_total = st["batches"] * W_enc + model_calls * W_dec + kv_read + kv_write
    dec_io = 4 * (Ld * ((d + 3 * d) + 3 * (d + d) + (d + F) + (F + 2 * d)) + (d + V))
    enc_io = 4 * (Le * ((d + 3 * d) + (d + d) + (d + F) + (F + d)) + (d + Ld * 2 * d))
    gemm_bytes = pos * dec_io + src_tok * enc_io + model_calls * W_dec + st["batches"] * (W_enc + 4 * Ld * 2 * d * d)
    return {"gemm_flops": float(gemm_flops), "bytes": float(bytes_total), "gemm_bytes": float(gemm_bytes)}


def main_beam(a):
    """configs[2] / configs[3]: beam-search speculative decoding, whole loop native (ttx_beam_speculative_generate_many).
    A step = one given batch; `--inflight` batches are on the GPU at once (one session + stream each); outputs and counters
    per batch are those of one-at-a-time calls (tests/test_gpu_beam_native.py)."""
    bc = BEAM_CONFIGS[a.config]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and os.environ.get("TTX_SHARE_GPU") != "1":
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({n_dev} visible)")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(os.environ.get("TTX_DIST_BACKEND", "nccl"))
    import translation_transformer_amd as tta
    from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK, V

    setup = {}
    sd = get_weights(a.train_steps, dev, bc["kind"], bc["layers"], setup) if rank == 0 else None
    model = tta.dist.broadcast_model(sd, 8, PAD, local_rank, dist)
    cfg = {"d": model.emb_dim, "F": model.ff_dim, "V": model.tgt_vocab_size, "Le": model.num_enc_layers, "Ld": model.num_dec_layers}
    assert cfg["Le"] == bc["layers"] and cfg["Ld"] == bc["layers"]
    per_rank = (a.steps + a.warmup) * a.batch_size
    src_all, _ = SynthReactions(123456, bc["kind"]).dataset(per_rank * world)
    mine = src_all[rank * per_rank:(rank + 1) * per_rank]
    dev_batches = [torch.from_numpy(b).to(dev) for b in batches(mine, a.batch_size)]
    warm, timed = dev_batches[:a.warmup], dev_batches[a.warmup:]
    K, N, D, L = bc["n_best"], a.n_drafts, a.draft_len, a.max_len

    def make_gen(m, smart):
        return tta.TranslationInferenceBeamSearchSpeculative(m, L, K, D, N, V, bool(smart), PAD, BOS, EOS, C_TOK, max_steps=4 * L)

    def run(gen, bs, inflight):
        return gen.generate_many(bs, in_flight=inflight) if inflight > 1 else [gen.generate(b) for b in bs]

    smart = bool(a.smart)
    g0 = make_gen(model, smart)
    run(g0, (warm * a.inflight)[:max(a.inflight, len(warm))], a.inflight)     # every session of the pool sized and warm
    log("warmup done", g0.model_calls_num, "calls")
    gen = make_gen(model, smart)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run(gen, timed, a.inflight)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # prediction gather (C2): hypotheses padded to max_len, one collective at the end
    flat = torch.cat([torch.nn.functional.pad(o, (0, L - o.shape[2]), value=PAD) for o in outs])
    gathered = tta.dist.gather_predictions(flat, world * flat.shape[0], dist)
    if rank == 0:
        assert gathered.shape[0] == world * flat.shape[0]
    counted = tta.dist.sum_counters({"reactions": sum(int(b.shape[0]) for b in timed)}, dev, dist)
    n_reactions = int(counted["reactions"])
    top1_eos = int((flat[:, 0] == EOS).any(dim=1).sum())
    value = n_reactions / elapsed
    line = {
        "metric": "reactions/sec (SMILES decoded), beam-search speculative", "value": value, "unit": "reactions/s", "n_gpus": world,
        "steps": len(timed), "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / max(1, len(timed)), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": value / bc["published"], "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{bc['name']} bs={a.batch_size} n_drafts={N} draft_len={D} max_len={L} "
                               f"smart_drafts_mode={smart}, d=256 8h FFN2048 {bc['layers']}+{bc['layers']} fp32, weights trained "
                               f"{a.train_steps} steps on the synthetic task",
                   "baseline_config": a.config, "reactions": n_reactions,
                   "parallelism": f"test-set shards x{world}, no per-step collective", "batches_in_flight_per_gpu": a.inflight},
        "model_calls": gen.model_calls_num, "acceptance_rate": gen.accepted_tokens_num / max(1, gen.produced_non_pad_tokens),
        "top1_rows_with_eos_rank0": top1_eos, "rows_rank0": int(flat.shape[0]),
        "device_ms_encode_rank0": gen.stats_total["encode_ms"], "device_ms_decode_rank0": gen.stats_total["decode_ms"],
        "setup": setup,
    }
    if rank == 0:
        st = dict(gen.stats_total, produced=gen.produced_non_pad_tokens)
        work = beam_work(cfg, st, gen.model_calls_num, "verified_positions")
        line["hbm_algorithmic"] = {"bytes_per_reaction": work["bytes"] / max(1, len(timed) * a.batch_size),
                                   "achieved_GBs": work["bytes"] / elapsed / 1e9,
                                   "frac_of_peak": work["bytes"] / elapsed / 1e9 / PEAK_HBM_GBS}
        if not a.timed_only:
            if a.inflight > 1:
                g1 = make_gen(model, smart)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                seq = run(g1, timed, 1)
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
                line["one_batch_at_a_time"] = {"value": n_reactions / world / dt1, "unit": "reactions/s",
                                               "identical_to_timed_outputs": all(torch.equal(x, y) for x, y in zip(seq, outs))
                                               and g1.model_calls_num == gen.model_calls_num}
            g2 = make_gen(model, not smart)
            run(g2, warm, a.inflight)
            g2 = make_gen(model, not smart)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            other = run(g2, timed, a.inflight)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            line["other_draft_mode"] = {"smart_drafts_mode": not smart, "value": n_reactions / world / dt2, "unit": "reactions/s",
                                        "model_calls": g2.model_calls_num,
                                        "top1_identical_to_timed_run": sum(int(torch.equal(x[:, 0, :min(x.shape[2], y.shape[2])],
                                                                                           y[:, 0, :min(x.shape[2], y.shape[2])]))
                                                                           for x, y in zip(other, outs)),
                                        "batches": len(outs)}
        if not a.no_profile:
            # dominant kernel family = the fp32 MFMA GEMMs: HIP events on the launch stream around every GEMM launch of the
            # same batches, one after the other on a profiling session (raw event-pair time)
            import ctypes as C
            os.environ["TTX_PROFILE_GEMM"] = "1"
            pm = tta.NativeTransformer(sd, num_heads=8, pad_token_idx=PAD, device=local_rank)
            os.environ.pop("TTX_PROFILE_GEMM")
            pg = make_gen(pm, smart)
            for b in timed:
                pg.generate(b)
            ms, n, e = C.c_double(), C.c_int64(), C.c_double()
            pm._lib.ttx_last_kernel_profile(pm.session, C.byref(ms), C.byref(n), C.byref(e))
            gemm_ms, launches, empty_ms = ms.value, n.value, e.value
            pst = dict(pg.stats_total, produced=pg.produced_non_pad_tokens)
            pw = beam_work(cfg, pst, pg.model_calls_num, "verified_positions")
            pwx = beam_work(cfg, pst, pg.model_calls_num, "executed_positions")
            net_ms = max(1e-9, gemm_ms - launches * empty_ms)
            ach_raw = pw["gemm_flops"] / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            ach = pw["gemm_flops"] / (net_ms * 1e-3) / 1e12
            line["roofline"] = {"kernel": "fp32 MFMA GEMM family of the verify step under the small-row policy (k_gemm3<KW> 32x32 K-split for "
                                          "the K = 256 GEMMs up to N = 768, k_gemm2<NT> 64x64 for FFN1 and for FFN2 as 8 K-slices; v_mfma_f32_32x32x2_f32), every launch of the run",
                                "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / PEAK_F32_MATRIX_TFLOPS, "traffic": None,
                                "traffic_note": "no PMC pass was collected for this configuration",
                                "launches": launches, "avg_launch_us": 1e3 * net_ms / max(1, launches),
                                "avg_launch_us_raw_event_pairs": 1e3 * gemm_ms / max(1, launches), "achieved_raw_event_pairs": ach_raw,
                                "flops_per_launch": pw["gemm_flops"] / max(1, launches),
                                "algorithmic_bytes_per_launch": pw["gemm_bytes"] / max(1, launches),
                                "achieved_counting_executed_rows": pwx["gemm_flops"] / (net_ms * 1e-3) / 1e12,
                                "event_pair_overhead_us": 1e3 * empty_ms,
                                "gemm_share_of_device_time": gemm_ms / max(1e-9, pst["encode_ms"] + pst["decode_ms"]),
                                "note": "one batch at a time on the profiling session: a verify step has a few hundred to ~1 700 rows, "
                                        "so these launches are latency-bound; the fraction is what the step's GEMMs reach, not the chip"}
            pm.close()
        if world == 1 and not a.no_cpu_baseline:
            from oracle.model import OracleTransformer, config_from_state
            from oracle.spec_beam import BeamSearchSpeculativeOracle
            cores = usable_cores()
            torch.set_num_threads(cores)
            om = OracleTransformer(config_from_state(sd, 8, PAD), sd)
            og = BeamSearchSpeculativeOracle(om, L, K, D, N, V, smart, PAD, BOS, EOS, C_TOK, max_steps=4 * L)
            sample = timed[:max(1, a.cpu_batches if a.cpu_batches != 3 else 1)]
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
            line["cpu_baseline"] = {"value": n_cpu / cpu_s, "unit": "reactions/s", "cores": cores, "kind": "port",
                                    "sample": f"first {len(sample)} timed batch(es) = {n_cpu} reactions of the same workload, oracle/ "
                                              f"(full-prefix recompute like the reference), torch {torch.__version__} fp32",
                                    "seconds": cpu_s, "model_calls": og.model_calls_num}
            line["parity"] = {"top1_rows_token_identical_to_oracle": top1, "rows_checked": n_cpu,
                              "hypotheses_token_identical_to_oracle": all_ranks, "hypotheses_checked": total}
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.timed_only:
        a.no_profile = a.no_cpu_baseline = True
    if a.config in BEAM_CONFIGS:
        return main_beam(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and os.environ.get("TTX_SHARE_GPU") != "1":
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({n_dev} visible)")
    local_rank = local_rank % n_dev          # TTX_SHARE_GPU=1: rehearsal of the N>1 path on a 1-GPU box
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(os.environ.get("TTX_DIST_BACKEND", "nccl"))   # "nccl" = RCCL over xGMI

    import translation_transformer_amd as tta
    from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK, V

    # ---- weights: rank 0 trains/loads and packs them into its HBM blob; every other rank receives that blob with ONE RCCL
    # broadcast straight into its own (empty) model's blob (dist.broadcast_model)
    setup = {}
    sd = get_weights(a.train_steps, dev, info=setup) if rank == 0 else None
    model = tta.dist.broadcast_model(sd, 8, PAD, local_rank, dist)      # one RCCL broadcast of the packed blob into HBM
    cfg = {"d": model.emb_dim, "F": model.ff_dim, "V": model.tgt_vocab_size, "Le": model.num_enc_layers,
           "Ld": model.num_dec_layers}

    # ---- data: every rank takes its own contiguous shard of the synthetic test set (seed 123456)
    per_rank = (a.steps + a.warmup) * a.batch_size
    src_all, _ = SynthReactions(123456, "mit").dataset(per_rank * world)
    mine = src_all[rank * per_rank:(rank + 1) * per_rank]
    dev_batches = [torch.from_numpy(b).to(dev) for b in batches(mine, a.batch_size)]
    warm, timed = dev_batches[:a.warmup], dev_batches[a.warmup:]

    def make_gen(m):
        return tta.TranslationInferenceGreedySpeculative(m, a.max_len, a.draft_len, a.n_drafts, PAD, BOS, EOS, C_TOK)

    # "rows": the K given batches are decoded as length-sorted row groups and replayed per given batch (exact);
    # "batches": every given batch is decoded as given.  Both keep a.inflight groups/batches on the GPU at once.
    rows_sched = a.schedule == "rows" and a.inflight > 1
    gen = make_gen(model)
    log("model ready; warmup")
    for b in warm:
        gen.generate(b)
    if a.inflight > 1:     # warm every session of the pool (workspaces, graph capture) at the timed region's grouping
        reps = max(a.inflight, -(-len(timed) // max(1, len(warm)))) if rows_sched else a.inflight
        gen.generate_many((warm * reps)[:max(a.inflight, len(timed))], in_flight=a.inflight, reorder=rows_sched)
    log("warmup done", gen.model_calls_num, "calls")
    gen = make_gen(model)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # A batch on which the reference itself raises (a row finishing at a width beyond max_len) is skipped and not
    # counted: its rows are reported as all-PAD and listed under "batches_reference_raises".
    raised = []
    if a.inflight > 1:
        outs = gen.generate_many(timed, in_flight=a.inflight, reorder=rows_sched, on_error="skip")
        raised = list(gen.last_failed_batches)
    else:
        outs = []
        for i, b in enumerate(timed):
            try:
                outs.append(gen.generate(b))
            except tta.ReferenceError_:
                outs.append(None)
                raised.append(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log("timed region done", elapsed, "s", gen.model_calls_num, "calls")
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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

    line = {
        "metric": "reactions/sec (SMILES decoded), greedy speculative", "value": n_reactions / elapsed,
        "unit": "reactions/s", "n_gpus": world, "steps": len(timed), "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / max(1, len(timed)), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": (n_reactions / elapsed) / BASELINE_REACTIONS_PER_S, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"USPTO-MIT-shaped synthetic SMILES, greedy speculative draft_len={a.draft_len} "
                               f"n_drafts={a.n_drafts} bs={a.batch_size} max_len={a.max_len}, d=256 8h FFN2048 4+4 fp32, "
                               f"weights trained {a.train_steps} steps on the synthetic task",
                   "reactions": n_reactions, "parallelism": f"test-set shards x{world}, no per-step collective",
                   "batches_in_flight_per_gpu": a.inflight,
                   "schedule": ("rows of the K given batches regrouped by source length on the device, reference loop replayed "
                                "per given batch (outputs and model_calls identical to per-batch generate)") if rows_sched
                               else "batches decoded as given"},
        "batches_reference_raises": int(counted["raised"]),
        "model_calls": stats["model_calls"], "rows_finished_rank0": finished, "rows_rank0": int(preds.shape[0]),
        "accepted_per_step_per_row": stats["accepted_tokens"] / max(1, stats["produced_tokens"] - stats["accepted_tokens"]),
        "device_ms_encode_rank0": stats["encode_ms"], "device_ms_decode_rank0": stats["decode_ms"],
        "setup": setup,
    }

    if rank == 0:
        # work the device executed: under the row schedule that is the row groups', not the given batches'
        dstats = dict(stats["device"], encode_ms=stats["encode_ms"], decode_ms=stats["decode_ms"]) if rows_sched else stats
        work = flops_and_bytes(cfg, dstats, dstats["src_tokens_padded"], dstats.get("batches", len(timed)))
        line["hbm_algorithmic"] = {"bytes_per_reaction": work["bytes"] / (len(timed) * a.batch_size),
                                   "achieved_GBs": work["bytes"] / elapsed / 1e9,
                                   "frac_of_peak": work["bytes"] / elapsed / 1e9 / PEAK_HBM_GBS}
        if a.inflight > 1 and not a.timed_only:
            # the same K batches strictly one at a time (the reference's predict loop), for comparison
            g1 = make_gen(model)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            seq = []
            for i, b in enumerate(timed):
                try:
                    seq.append(g1.generate(b))
                except tta.ReferenceError_:
                    seq.append(None)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            ok1 = [i for i, o in enumerate(seq) if o is not None]
            line["one_batch_at_a_time"] = {"value": sum(int(timed[i].shape[0]) for i in ok1) / dt1, "unit": "reactions/s",
                                           "identical_to_timed_outputs": sorted(set(range(len(timed))) - set(ok1)) == sorted(raised)
                                           and all(torch.equal(seq[i], outs[i]) for i in ok1)}
        if rows_sched and not a.timed_only:
            # the same K batches decoded as given (no regrouping), a.inflight of them at a time
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
            line["device_model_calls"] = stats["device"]["model_calls"]
            line["device_src_tokens_padded"] = stats["device"]["src_tokens_padded"]
        if not a.timed_only:
            # The same K batches through the kept Lightning surface (src/model/lightning_model.py:209-243): a trainer
            # stand-in that only calls on_predict_start -> predict_step per batch -> on_predict_end, i.e. what main.py's
            # Trainer.predict does; predict_step serves the batches from windows decoded ahead (slot pools).
            line["predict_step_surface"] = predict_surface(tta, sd, timed, warm, a, outs, raised, (PAD, BOS, EOS, C_TOK, V))
        if not a.no_profile:
            # dominant kernel = k_gemm_tn (fp32 MFMA GEMM): HIP events on the launch stream around every launch,
            # same batches, same process, right after the timed region
            os.environ["TTX_PROFILE_GEMM"] = "1"
            pm = tta.NativeTransformer(sd, num_heads=8, pad_token_idx=PAD, device=local_rank)
            os.environ.pop("TTX_PROFILE_GEMM")
            pg = make_gen(pm)
            import ctypes as C
            gemm_ms, launches, empty_ms = 0.0, 0, 0.0
            ms, n, e = C.c_double(), C.c_int64(), C.c_double()
            if rows_sched:
                # the timed region's workload: the same row groups, one after the other on the profiling session
                pg.generate_many(timed, in_flight=1, reorder=True, group_size=gen.last_group_size, on_error="skip")
                pm._lib.ttx_last_kernel_profile(pm.session, C.byref(ms), C.byref(n), C.byref(e))
                gemm_ms, launches, empty_ms = ms.value, n.value, e.value
            else:
                for b in timed:
                    try:
                        pg.generate(b)
                    except tta.ReferenceError_:
                        pass
                    pm._lib.ttx_last_kernel_profile(pm.session, C.byref(ms), C.byref(n), C.byref(e))
                    gemm_ms += ms.value
                    launches += n.value
                    empty_ms = e.value
            raw_ms = gemm_ms
            net_ms = max(1e-9, gemm_ms - launches * empty_ms)      # with the cost of an empty event pair removed
            pstats = dict(pg.stats_total)
            pstats["model_calls"] = pg.model_calls_num
            if rows_sched:
                pstats = dict(pstats["device"], encode_ms=pstats["encode_ms"], decode_ms=pstats["decode_ms"])
            pw = flops_and_bytes(cfg, pstats, pstats["src_tokens_padded"], pstats.get("batches", len(timed)))
            # a launch's duration = its event pair minus what an EMPTY pair measures on the same stream in the same run (the
            # cost of recording the two events, ~4.7 us); the rocprofv3 kernel trace of this very pass agrees with the net
            # figure (profiles/r02_s20_roofline_pass_from_trace.txt: 28.4 us by the trace, 27-28 us net, 32 us raw)
            ach_raw = pw["gemm_flops"] / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            ach = pw["gemm_flops"] / (net_ms * 1e-3) / 1e12
            pmc, pmc_why = pmc_traffic_for({"config": a.config, "steps": len(timed), "warmup": a.warmup, "schedule": a.schedule,
                                            "inflight": a.inflight, "batch_size": a.batch_size, "n_drafts": a.n_drafts,
                                            "draft_len": a.draft_len, "max_len": a.max_len})
            line["roofline"] = {"kernel": "k_gemm24<NT> / k_gemm2<NT> (fp32 v_mfma_f32_32x32x2_f32 GEMM family: 128x128 or 64x64 tiles picked "
                                          "per launch from the live row count; every GEMM launch of the pass: encoder, cross K/V, verify steps)",
                                "bound": "mfma", "achieved": ach,
                                "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MATRIX_TFLOPS,
                                "traffic": pmc.get("bytes_per_launch") if pmc else None, "launches": launches,
                                "traffic_note": (f"HBM-side bytes per GEMM launch (2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included) from "
                                                 f"rocprofv3 --pmc passes of this very command: {pmc.get('file')}") if pmc else pmc_why,
                                "algorithmic_bytes_per_launch": pw["gemm_bytes"] / max(1, launches), "avg_launch_us": 1e3 * net_ms / max(1, launches),
                                "measured_on": "a second pass over the same rows right after the timed region: one profiling session (its own "
                                               "stream, one slot pool of up to 512 rows, eager launches), a HIP event pair around every GEMM "
                                               "launch on that stream",
                                "note": "achieved/avg_launch_us = event-pair time minus event_pair_overhead_us per launch (what an empty pair "
                                        "measures on the same stream in this run); the raw pair figures are beside them; the rocprofv3 "
                                        "kernel trace of the same pass is summarised under profiles/ (tools/roofline_from_trace.py)",
                                "event_pair_overhead_us": 1e3 * empty_ms,
                                "avg_launch_us_raw_event_pairs": 1e3 * raw_ms / max(1, launches),
                                "achieved_raw_event_pairs": ach_raw, "frac_raw_event_pairs": ach_raw / PEAK_F32_MATRIX_TFLOPS,
                                "flops_per_launch": pw["gemm_flops"] / max(1, launches),
                                "gemm_share_of_decode_time": gemm_ms / max(1e-9, pstats["encode_ms"] + pstats["decode_ms"])}
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
            sample_idx = [i for i in range(len(timed)) if i not in raised][:a.cpu_batches]
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
            line["parity"] = {"rows_token_identical_to_oracle": rows_same, "rows_checked": n_cpu,
                              "batches_identical": same}
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
