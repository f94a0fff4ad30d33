"""Generators with the reference's constructor keywords, counters and ``generate`` contract
(src/model/lightning_model.py:92-137 shows how they are built), running on the
HIP library.  ``model`` must be a ``NativeTransformer``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _native as N
from .model import NativeTransformer


def _need_native(model) -> NativeTransformer:
    if not isinstance(model, NativeTransformer):
        raise TypeError("the native generators run on a NativeTransformer (no eager/PyTorch fallback exists)")
    return model


class TranslationInferenceGreedySpeculative:
    """Drop-in for src/decoding/speculative_decoding.py:8-174: copy-drafts, one parallel verify pass per
    step, longest accepted prefix + 1 bonus token — the whole loop runs in ttx_greedy_speculative_generate."""

    def __init__(self, model, max_len: int, draft_len: int, n_drafts: int, pad_token: int, bos_token: int,
                 eos_token: int, replace_token: int) -> None:
        self.model = _need_native(model)
        self.max_len = max_len
        self.pad_token, self.bos_token, self.eos_token = pad_token, bos_token, eos_token
        self.replace_token = replace_token
        self.draft_len = draft_len
        self.n_drafts = n_drafts
        self.accepted_tokens_num = 0   # left at 0 by the reference's greedy-speculative loop as well
        self.model_calls_num = 0
        self.stats_total = {"accepted_tokens": 0, "produced_tokens": 0, "verified_positions": 0,
                            "kv_prefix_positions": 0, "src_positions": 0, "encode_ms": 0.0, "decode_ms": 0.0,
                            "src_tokens_padded": 0, "batches": 0}
        self.last_stats: N.GenStats | None = None
        self.last_failed_batches: list = []    # generate_many(on_error="skip"): batches on which the reference raises
        self.last_batch_counters: list = []    # generate_many: what each batch added to the counter attributes (None: failed)
        self.record_step = 0           # parity tests: k > 0 keeps the logits of verify step k (see step_snapshot)

    def step_snapshot(self) -> dict:
        """The verify step recorded through ``record_step`` during the last ``generate``: numpy arrays
        logits [n_active, rps, V], rows [n_active] (batch row of every slot), front [B], gen [B, gen_ld]."""
        import numpy as np
        m = self.model
        info = (C.c_int32 * 6)()
        N.check(m._lib.ttx_debug_step_snapshot(m.session, info, None, None, None, None))
        n_act, rps, B, gen_ld, V, step = list(info)
        logits = np.empty((n_act, rps, V), dtype=np.float32)
        act = np.empty(n_act, dtype=np.int32)
        front = np.empty(B, dtype=np.int32)
        gen = np.empty((B, gen_ld), dtype=np.int32)
        N.check(m._lib.ttx_debug_step_snapshot(m.session, info, logits.ctypes.data, act.ctypes.data, front.ctypes.data,
                                               gen.ctypes.data))
        return {"logits": logits, "rows": act, "front": front, "gen": gen, "step": step}

    def __str__(self):
        return (f"Greedy speculative decoding (draft_len={self.draft_len}, n_drafts={self.n_drafts}, "
                f"max_len={self.max_len})")

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        m = self.model
        src = src.to(m.device, torch.int64).contiguous()
        m.check_tokens(src)
        B, Ls = src.shape
        out = torch.empty((B, 1, self.max_len), dtype=torch.int64, device=m.device)
        p = N.GenParams(self.max_len, self.draft_len, self.n_drafts, self.pad_token, self.bos_token, self.eos_token,
                        self.replace_token, int(self.record_step))
        st = N.GenStats()
        N.check(m._lib.ttx_greedy_speculative_generate(m.session, src.data_ptr(), B, Ls, C.byref(p), out.data_ptr(),
                                                       C.byref(st), m._stream()))
        self.model_calls_num += int(st.model_calls)
        t = self.stats_total
        for k in ("accepted_tokens", "produced_tokens", "verified_positions", "kv_prefix_positions", "src_positions",
                  "encode_ms", "decode_ms"):
            t[k] += getattr(st, k)
        t["src_tokens_padded"] += B * Ls
        t["batches"] += 1
        self.last_stats = st
        return out

    def generate_many(self, batches: list, in_flight: int = 4, reorder: bool = False, group_size: int | None = None,
                      on_error: str = "raise", pool: bool | None = None) -> list:
        """Decode several batches with up to `in_flight` of them on the GPU at once (one session + stream each;
        ttx_greedy_speculative_generate_many).  Returns one [B,1,max_len] tensor per batch, each identical to what
        ``generate`` returns for that batch; counters accumulate as if ``generate`` had been called per batch.

        ``reorder=True`` decodes the rows in groups sorted by source length (less padding, rows of similar length
        finish together) through ttx_greedy_speculative_generate_rows and then replays the reference's width rule
        over the batches as passed (scheduling.replay_batch): outputs, ``model_calls_num`` and the error behaviour
        stay those of per-batch ``generate`` calls.

        ``on_error="skip"``: a batch on which the reference raises (a row finishing at a width beyond max_len,
        speculative_decoding.py:158) yields ``None`` in the returned list instead of ending the call; the indices are
        kept in ``self.last_failed_batches``."""
        m = self.model
        if on_error not in ("raise", "skip"):
            raise ValueError("on_error must be 'raise' or 'skip'")
        self.last_failed_batches = []
        self.last_batch_counters = [None] * len(batches)
        if not batches:
            return []
        if reorder:
            try:
                if pool is None:
                    pool = True
                return self._generate_reordered(batches, in_flight, group_size, on_error, pool)
            except N.TtxError as e:
                if e.code != N.TTX_ERR_ROW_REPLAY:
                    raise            # otherwise: a PAD inside a sequence; decode the batches as given
        srcs = [b.to(m.device, torch.int64).contiguous() for b in batches]
        for b in srcs:
            m.check_tokens(b)
        outs = [torch.empty((s.shape[0], 1, self.max_len), dtype=torch.int64, device=m.device) for s in srcs]
        n = len(srcs)
        pool = m.session_pool(max(1, min(in_flight, n)))
        sess = (C.c_void_p * len(pool))(*[p.value for p in pool])
        src_p = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
        out_p = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        Bs = (C.c_int * n)(*[s.shape[0] for s in srcs])
        Ls = (C.c_int * n)(*[s.shape[1] for s in srcs])
        p = N.GenParams(self.max_len, self.draft_len, self.n_drafts, self.pad_token, self.bos_token, self.eos_token,
                        self.replace_token, 0)
        stats = (N.GenStats * n)()
        rc = m._lib.ttx_greedy_speculative_generate_many(sess, len(pool), n, src_p, Bs, Ls, C.byref(p), out_p, stats, m._stream())
        if not (rc == N.TTX_ERR_REFERENCE and on_error == "skip"):
            N.check(rc)
        t = self.stats_total
        for i, st in enumerate(stats):
            if st.status != N.TTX_OK:                  # only reachable in skip mode: every batch was decoded, this one raises
                if st.status != N.TTX_ERR_REFERENCE:
                    N.check(int(st.status))
                outs[i] = None
                self.last_failed_batches.append(i)
                continue
            self.model_calls_num += int(st.model_calls)
            self.last_batch_counters[i] = {"model_calls_num": int(st.model_calls)}
            for k in ("accepted_tokens", "produced_tokens", "verified_positions", "kv_prefix_positions", "src_positions",
                      "encode_ms", "decode_ms"):
                t[k] += getattr(st, k)
            t["src_tokens_padded"] += srcs[i].shape[0] * srcs[i].shape[1]
            t["batches"] += 1
        return outs

    def _generate_reordered(self, batches: list, in_flight: int, group_size: int | None, on_error: str = "raise",
                            pool: bool = True) -> list:
        from .scheduling import plan_row_groups, replay_batch
        m = self.model
        L, T = self.max_len, self.max_len + 1
        srcs = [b.to(m.device, torch.int64) for b in batches]
        sizes = [int(s.shape[0]) for s in srcs]
        R = sum(sizes)
        # device group size: the given batch size is not binding any more; larger groups run the GEMMs at better MFMA
        # occupancy, but at least `in_flight` groups should exist so that tails overlap
        if pool:      # slot pool: 512 slots x 5 pools for long lists (measured); short lists are split evenly
            gsz = int(group_size or os.environ.get("TTX_POOL_CAPACITY") or 512)
        else:
            gsz = int(group_size or min(256, max(max(sizes), -(-R // max(1, in_flight)))))
        self.last_group_size = gsz
        # all rows in one right-padded matrix; a row's length is the position after its last non-PAD token
        Lmax = max(int(s.shape[1]) for s in srcs)
        allsrc = torch.full((R, Lmax), self.pad_token, dtype=torch.int64, device=m.device)
        r0 = 0
        for s in srcs:
            allsrc[r0:r0 + s.shape[0], :s.shape[1]] = s
            r0 += s.shape[0]
        m.check_tokens(allsrc)
        pos = torch.arange(1, Lmax + 1, device=m.device)
        lengths = ((allsrc != self.pad_token) * pos).amax(dim=1)
        order, groups = plan_row_groups(lengths.cpu().numpy(), gsz)
        order_t = torch.from_numpy(order).to(m.device)
        sorted_src = allsrc[order_t]
        sorted_len = lengths[order_t].cpu().numpy()
        out_sorted = torch.empty((R, L), dtype=torch.int64, device=m.device)
        traj_sorted = torch.empty((R, T), dtype=torch.int16, device=m.device)
        fin_sorted = torch.empty((R,), dtype=torch.int32, device=m.device)
        p = N.GenParams(L, self.draft_len, self.n_drafts, self.pad_token, self.bos_token, self.eos_token, self.replace_token, 0)
        if pool:
            # continuous batching: every session keeps `gsz` slots filled from the sorted work list
            # (ttx_greedy_speculative_generate_pool)
            n_sess = max(1, min(in_flight, max(1, 2560 // gsz), -(-R // 32)))     # five pools (sweeps: profiles/r03_sweep_c2_pools.txt)
            if os.environ.get("TTX_POOL_SESSIONS"):                               # experiments
                n_sess = max(1, int(os.environ["TTX_POOL_SESSIONS"]))
            sessions = m.session_pool(n_sess)
            sess = (C.c_void_p * len(sessions))(*[q.value for q in sessions])
            width = max(2, int(sorted_len.max()))
            src_mat = sorted_src[:, :width].contiguous() if width <= sorted_src.shape[1] else torch.nn.functional.pad(
                sorted_src, (0, width - sorted_src.shape[1]), value=self.pad_token).contiguous()
            h_len = (C.c_int32 * R)(*[int(x) for x in sorted_len])
            one = N.GenStats()
            N.check(m._lib.ttx_greedy_speculative_generate_pool(sess, len(sessions), src_mat.data_ptr(), R, width, h_len, gsz,
                                                                C.byref(p), out_sorted.data_ptr(), traj_sorted.data_ptr(),
                                                                fin_sorted.data_ptr(), C.byref(one), m._stream()))
            stats, gsrc = [one], [src_mat]
        else:
            gsrc = [sorted_src[g, :max(2, int(sorted_len[g].max()))].contiguous() for g in groups]
            n = len(gsrc)
            sessions = m.session_pool(max(1, min(in_flight, n)))
            sess = (C.c_void_p * len(sessions))(*[q.value for q in sessions])
            src_p = (C.c_void_p * n)(*[s.data_ptr() for s in gsrc])
            out_p = (C.c_void_p * n)(*[out_sorted[g].data_ptr() for g in groups])
            traj_p = (C.c_void_p * n)(*[traj_sorted[g].data_ptr() for g in groups])
            fin_p = (C.c_void_p * n)(*[fin_sorted[g].data_ptr() for g in groups])
            Bs = (C.c_int * n)(*[s.shape[0] for s in gsrc])
            Ls = (C.c_int * n)(*[s.shape[1] for s in gsrc])
            stats = (N.GenStats * n)()
            N.check(m._lib.ttx_greedy_speculative_generate_rows(sess, len(sessions), n, src_p, Bs, Ls, C.byref(p), out_p, traj_p, fin_p,
                                                                stats, m._stream()))
        # back to the caller's row order, then the reference's per-batch loop over the traces
        inv = torch.empty_like(order_t)
        inv[order_t] = torch.arange(R, device=m.device)
        out_rows = out_sorted[inv]
        traj = traj_sorted[inv].cpu().numpy()
        fin = fin_sorted[inv].cpu().numpy()
        keep = torch.zeros(R, dtype=torch.bool)
        t = self.stats_total
        failed = None
        r0 = 0
        for bi, B in enumerate(sizes):
            rep = replay_batch(traj[r0:r0 + B], fin[r0:r0 + B], L, self.draft_len, self.n_drafts)
            if rep.error:
                if on_error == "skip":
                    self.last_failed_batches.append(bi)
                    r0 += B
                    continue
                failed = bi
                break
            keep[r0:r0 + B] = torch.from_numpy(rep.finished)
            self.model_calls_num += rep.model_calls
            self.last_batch_counters[bi] = {"model_calls_num": rep.model_calls}
            t["accepted_tokens"] += rep.accepted_tokens
            t["produced_tokens"] += rep.produced_tokens
            t["verified_positions"] += rep.verified_positions
            t["kv_prefix_positions"] += rep.kv_prefix_positions
            t["src_positions"] += rep.rows_iterations * int(srcs[bi].shape[1])
            t["src_tokens_padded"] += B * int(srcs[bi].shape[1])
            t["batches"] += 1
            r0 += B
        for st in stats:
            t["encode_ms"] += st.encode_ms
            t["decode_ms"] += st.decode_ms
        # what the device actually executed (row groups), beside the reference-equivalent per-batch counters above
        dv = t.setdefault("device", {"model_calls": 0, "accepted_tokens": 0, "produced_tokens": 0, "verified_positions": 0,
                                     "kv_prefix_positions": 0, "src_positions": 0, "src_tokens_padded": 0, "batches": 0})
        for st, gs in zip(stats, gsrc):
            for k in ("model_calls", "accepted_tokens", "produced_tokens", "verified_positions", "kv_prefix_positions",
                      "src_positions"):
                dv[k] += int(getattr(st, k))
            dv["src_tokens_padded"] += int(st.src_tokens_padded) if pool else int(gs.numel())
            dv["batches"] += 1
        t["device_model_calls"] = dv["model_calls"]
        if failed is not None:
            raise N.ReferenceError_(f"batch {failed}: a row finished at a width beyond max_len: shape mismatch in the reference "
                                    "(speculative_decoding.py:158)")
        out_rows = torch.where(keep.to(m.device)[:, None], out_rows, torch.full_like(out_rows, self.pad_token))
        outs, r0 = [], 0
        skipped = set(self.last_failed_batches)
        for bi, B in enumerate(sizes):
            outs.append(None if bi in skipped else out_rows[r0:r0 + B].unsqueeze(1).contiguous())
            r0 += B
        return outs


class TranslationInferenceGreedy:
    """Drop-in for src/decoding/standard_decoding.py:4-55; the loop runs in ttx_greedy_generate (KV cache)."""

    def __init__(self, model, max_len: int, pad_token: int, bos_token: int, eos_token: int) -> None:
        self.model = _need_native(model)
        self.max_len = max_len
        self.pad_token, self.bos_token, self.eos_token = pad_token, bos_token, eos_token
        self.model_calls_num = 0
        self.given_tokens = 0

    def __str__(self):
        return f"Greedy decoding (max_len={self.max_len})"

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        m = self.model
        src = src.to(m.device, torch.int64).contiguous()
        m.check_tokens(src)
        B, Ls = src.shape
        out = torch.empty((B, 1, self.max_len), dtype=torch.int64, device=m.device)
        p = N.GenParams(self.max_len, 0, 1, self.pad_token, self.bos_token, self.eos_token, self.pad_token, 0)
        st = N.GenStats()
        N.check(m._lib.ttx_greedy_generate(m.session, src.data_ptr(), B, Ls, C.byref(p), out.data_ptr(), C.byref(st),
                                           m._stream()))
        self.model_calls_num += int(st.model_calls)
        self.given_tokens += int((src != m.src_pad_token_i).sum())
        return out


class TranslationInferenceBeamSearch:
    """Drop-in for src/decoding/standard_decoding.py:58-174: the whole loop runs in ttx_beam_generate (per-hypothesis KV
    cache, encoder and cross K/V once per source, log-softmax + top-beam + row assembly in one kernel per step)."""

    def __init__(self, model, beam_size: int, max_len: int, pad_token: int, bos_token: int, eos_token: int):
        assert max_len > 1
        assert beam_size > 0
        self.model = _need_native(model)
        self.beam_size, self.max_len = beam_size, max_len
        self.pad_token, self.bos_token, self.eos_token = pad_token, bos_token, eos_token
        self.model_calls_num = 0
        self.given_tokens = 0
        self.b_sz = 0

    def __str__(self):
        return f"Beam search decoding (beam_size={self.beam_size}, max_len={self.max_len})"

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        m, K = self.model, self.beam_size
        src = src.to(m.device, torch.int64).contiguous()
        m.check_tokens(src)
        B, Ls = src.shape
        out = torch.empty((B, K, self.max_len), dtype=torch.int64, device=m.device)
        p = N.BeamSearchParams(self.max_len, K, self.pad_token, self.bos_token, self.eos_token)
        st = N.BeamSearchStats()
        N.check(m._lib.ttx_beam_generate(m.session, src.data_ptr(), B, Ls, C.byref(p), out.data_ptr(), C.byref(st), m._stream()))
        self.model_calls_num += int(st.model_calls)
        self.b_sz += int(st.running_rows)
        self.given_tokens += int((src != m.src_pad_token_i).sum())
        return out[:, :, :int(st.out_width)].contiguous()


class TranslationInferenceBeamSearchSpeculative:
    """Drop-in for src/decoding/speculative_decoding.py:241-869 (both draft modes): the whole loop — candidate rows, draft
    slots, KV-cached verify step, accepted lengths, best draft, leaf enumeration and scoring, per-source selection,
    termination — runs in ttx_beam_speculative_generate; Python allocates the output tensor and keeps the counters."""

    def __init__(self, model, max_len: int, n_best: int, draft_len: int, n_drafts: int, vocab_size: int,
                 smart_drafts_mode: bool, pad_token: int, bos_token: int, eos_token: int, C_token: int,
                 max_steps: int | None = None) -> None:
        self.model = _need_native(model)
        self.max_len, self.vocab_size = max_len, vocab_size
        self.smart_drafts_mode = smart_drafts_mode
        self.pad_token_idx, self.bos_token_idx, self.eos_token_idx, self.C_token_idx = pad_token, bos_token, eos_token, C_token
        self.n_best = n_best
        self.accepted_tokens_num = 0
        self.model_calls_num = 0
        self.model_input_lines_num = 0
        self.max_drafts_num = n_drafts
        self.n_drafts = 0
        self.requested_drafts_num = n_drafts
        self.produced_non_pad_tokens = 0
        self.max_draft_len, self.min_draft_len = 200, 5
        clamped = min(max(self.min_draft_len, draft_len), self.max_draft_len)
        if clamped != draft_len:
            print(f"The draft length should be in range [{self.min_draft_len}: {self.max_draft_len}], so it was changed to {clamped}")
        self.draft_len = clamped
        self.b_sz = 0
        # Safety valve absent from the reference, whose loop never ends when a candidate keeps emitting PAD before any
        # EOS; None = reference behaviour.
        self.max_steps = max_steps
        self.last_failed_batches: list = []    # generate_many(on_error="skip"): batches on which the reference raises / the guard trips
        self.last_batch_counters: list = []    # generate_many: what each batch added to the counter attributes (None: failed)
        # work the device executed, for bench.py's roofline
        self.stats_total = {"verified_positions": 0, "executed_positions": 0, "kv_prefix_positions": 0, "running_candidates": 0,
                            "src_tokens_padded": 0, "src_positions": 0, "encode_ms": 0.0, "decode_ms": 0.0, "batches": 0}
        if vocab_size != self.model.tgt_vocab_size:
            raise ValueError("vocab_size differs from the model's target vocabulary")

    def __str__(self):
        return (f"SpeculativeSampling decoding (n_best={self.n_best}, max_len={self.max_len}, "
                f"max_num_of_drafts={self.max_drafts_num}, draft_len={self.draft_len})")

    def _params(self) -> N.BeamParams:
        return N.BeamParams(self.max_len, self.n_best, self.draft_len, self.requested_drafts_num, int(bool(self.smart_drafts_mode)),
                            self.pad_token_idx, self.bos_token_idx, self.eos_token_idx, self.C_token_idx, int(self.max_steps or 0))

    _COUNTERS = ("model_calls_num", "accepted_tokens_num", "produced_non_pad_tokens", "model_input_lines_num", "b_sz", "n_drafts")

    def _counter_values(self) -> dict:
        return {k: getattr(self, k) for k in self._COUNTERS}

    def _account(self, st: N.BeamStats, B: int) -> None:
        t = self.stats_total
        for k in ("verified_positions", "executed_positions", "kv_prefix_positions", "running_candidates", "src_tokens_padded",
                  "encode_ms", "decode_ms"):
            t[k] += getattr(st, k)
        t["src_positions"] += int(st.running_candidates) * (int(st.src_tokens_padded) // max(1, B))
        t["batches"] += 1
        self.model_calls_num += int(st.model_calls)
        if self.smart_drafts_mode:                                               # only the smart-drafts loop counts these (:741, :760)
            self.model_input_lines_num += int(st.input_lines)
            self.b_sz += int(st.running_rows)
        self.accepted_tokens_num += int(st.accepted_tokens)
        self.produced_non_pad_tokens += int(st.produced_non_pad_tokens)
        if not self.smart_drafts_mode:
            self.n_drafts += B * self.requested_drafts_num                       # :434

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        m = self.model
        src = src.to(m.device, torch.int64).contiguous()
        m.check_tokens(src)
        B, Ls = src.shape
        out = torch.empty((B, self.n_best, self.max_len), dtype=torch.int64, device=m.device)
        p, st = self._params(), N.BeamStats()
        N.check(m._lib.ttx_beam_speculative_generate(m.session, src.data_ptr(), B, Ls, C.byref(p), out.data_ptr(), C.byref(st),
                                                     m._stream()))
        self._account(st, B)
        return out[:, :, :int(st.out_width)].contiguous()

    def generate_many(self, batches: list, in_flight: int = 4, pool: bool | None = None, on_error: str = "raise",
                      capacity: int | None = None) -> list:
        """Several batches on the GPU at once; every returned tensor and the counters are those of per-batch ``generate`` calls.

        ``pool`` (default: on for two or more batches, ``TTX_BEAM_POOL=0`` turns it off): the given batches are decoded in slot
        pools — continuous batching: whole batches are admitted as slots free up, one verify step per iteration serves a few
        hundred candidates of many batches, and the reference's batch-wide loop scalars (draft length, stop rule, smart-mode
        table width) are kept per batch on the device (ttx_beam_speculative_generate_pool); result width, model calls and
        counters per given batch come from per-source traces (scheduling.replay_beam_batch).  Otherwise the batches run as
        given, ``in_flight`` of them at once, one session + stream each (ttx_beam_speculative_generate_many).

        ``on_error="skip"``: a batch on which the reference raises (or the ``max_steps`` guard trips) yields ``None`` instead of
        ending the call; the indices are kept in ``self.last_failed_batches`` — ``generate`` on that batch raises the error."""
        m = self.model
        if on_error not in ("raise", "skip"):
            raise ValueError("on_error must be 'raise' or 'skip'")
        self.last_failed_batches = []
        self.last_batch_counters = [None] * len(batches)
        if not batches:
            return []
        srcs = [b.to(m.device, torch.int64).contiguous() for b in batches]
        for b in srcs:
            m.check_tokens(b)
        if pool is None:
            pool = len(srcs) > 1
        # smart mode needs a window library for every batch (Ls - 5 > 0, drafting.py:39); max_len < 3 never enters the loop
        d0 = self.draft_len if not self.smart_drafts_mode else min(max(5, self.draft_len + 1), 200) - 1
        if pool and self.max_len >= 3 and all(int(b.shape[1]) >= 2 and (not self.smart_drafts_mode or int(b.shape[1]) > 5) for b in srcs):
            return self._generate_pooled(srcs, in_flight, on_error, capacity, d0)
        return self._generate_as_given(srcs, list(range(len(srcs))), in_flight, on_error)

    def _generate_as_given(self, srcs: list, index: list, in_flight: int, on_error: str) -> list:
        m = self.model
        n = len(srcs)
        outs = [torch.empty((s.shape[0], self.n_best, self.max_len), dtype=torch.int64, device=m.device) for s in srcs]
        sessions = m.session_pool(max(1, min(in_flight, n)))
        sess = (C.c_void_p * len(sessions))(*[q.value for q in sessions])
        src_p = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
        out_p = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        Bs = (C.c_int * n)(*[s.shape[0] for s in srcs])
        Ls = (C.c_int * n)(*[s.shape[1] for s in srcs])
        p = self._params()
        stats = (N.BeamStats * n)()
        rc = m._lib.ttx_beam_speculative_generate_many(sess, len(sessions), n, src_p, Bs, Ls, C.byref(p), out_p, stats, m._stream())
        if rc not in (N.TTX_OK, N.TTX_ERR_REFERENCE, N.TTX_ERR_MAX_STEPS):
            N.check(rc)                            # not a per-batch outcome: the call itself failed
        res = []
        for i, (s, o, st) in enumerate(zip(srcs, outs, stats)):
            if int(st.status) != N.TTX_OK:         # every batch was decoded; this one raises in the reference / trips the guard
                if on_error == "raise" or int(st.status) not in (N.TTX_ERR_REFERENCE, N.TTX_ERR_MAX_STEPS):
                    if int(st.status) == N.TTX_ERR_MAX_STEPS:
                        raise RuntimeError("beam-speculative loop exceeded max_steps (non-terminating input)")
                    if int(st.status) == N.TTX_ERR_REFERENCE:
                        raise N.ReferenceError_(f"batch {index[i]}: fewer candidate leaves than n_best for a source (the reference "
                                                "asserts here, speculative_decoding.py:195)")
                    N.check(int(st.status))
                self.last_failed_batches.append(index[i])
                res.append(None)
                continue
            before = self._counter_values()
            self._account(st, s.shape[0])
            self.last_batch_counters[index[i]] = {k: v - before[k] for k, v in self._counter_values().items()}
            res.append(o[:, :, :int(st.out_width)].contiguous())
        return res

    def _generate_pooled(self, srcs: list, in_flight: int, on_error: str, capacity: int | None, d0: int) -> list:
        from .scheduling import replay_beam_batch
        m, K, L = self.model, self.n_best, self.max_len
        n = len(srcs)
        sizes = np.array([int(s.shape[0]) for s in srcs])
        widths = np.array([int(s.shape[1]) for s in srcs], dtype=np.int32)
        R = int(sizes.sum())
        Lmax = int(widths.max())
        allsrc = torch.full((R, Lmax), self.pad_token_idx, dtype=torch.int64, device=m.device)
        starts = np.concatenate([[0], np.cumsum(sizes)])
        for s, r0 in zip(srcs, starts):
            allsrc[r0:r0 + s.shape[0], :s.shape[1]] = s
        pos = torch.arange(1, Lmax + 1, device=m.device)
        lengths = ((allsrc != self.pad_token_idx) * pos).amax(dim=1).clamp(min=2).cpu().numpy().astype(np.int32)
        # work list: whole batches, the one with the longest source first (a chunk is encoded at its longest row's width)
        batch_long = np.array([lengths[starts[i]:starts[i + 1]].max() for i in range(n)])
        border = np.argsort(-batch_long, kind="stable")
        order = np.concatenate([np.arange(starts[b], starts[b + 1]) for b in border])
        order_t = torch.from_numpy(order).to(m.device)
        width = int(lengths.max())
        src_mat = allsrc[order_t][:, :width].contiguous()
        h_len = np.ascontiguousarray(lengths[order])
        h_batch = np.ascontiguousarray(np.repeat(np.arange(n, dtype=np.int32), sizes[border]))
        h_given = np.ascontiguousarray(widths[border])
        # pool size: about 16 k step rows per iteration (where the step GEMMs run on their large tilings); a few pools in flight
        rps = 1 + self.requested_drafts_num * d0
        cap = int(capacity or os.environ.get("TTX_BEAM_POOL_CAPACITY") or max(4, min(512, 16384 // max(1, K * rps))))
        cap = max(cap, int(sizes.max()))
        n_sess = max(1, min(in_flight, 4, -(-R // cap)))
        if os.environ.get("TTX_POOL_SESSIONS"):
            n_sess = max(1, int(os.environ["TTX_POOL_SESSIONS"]))
        sessions = m.session_pool(n_sess)
        sess = (C.c_void_p * len(sessions))(*[q.value for q in sessions])
        T_cap = L + 8
        out_sorted = torch.empty((R, K, L), dtype=torch.int64, device=m.device)
        tlen = torch.empty((R, T_cap), dtype=torch.int16, device=m.device)
        summ = torch.empty((R, 8), dtype=torch.int32, device=m.device)
        st = N.BeamStats()
        p = self._params()
        i32 = C.POINTER(C.c_int32)
        N.check(m._lib.ttx_beam_speculative_generate_pool(
            sess, len(sessions), src_mat.data_ptr(), R, width, h_len.ctypes.data_as(i32), h_batch.ctypes.data_as(i32), n,
            h_given.ctypes.data_as(i32), cap, C.byref(p), out_sorted.data_ptr(), tlen.data_ptr(), summ.data_ptr(), T_cap, C.byref(st),
            m._stream()))
        inv = torch.empty_like(order_t)
        inv[order_t] = torch.arange(R, device=m.device)
        out_rows = out_sorted[inv]
        tlen_h, summ_h = tlen[inv].cpu().numpy(), summ[inv].cpu().numpy()
        # what the device executed (for bench.py's roofline), once for the whole call
        t = self.stats_total
        for k in ("verified_positions", "executed_positions", "kv_prefix_positions", "running_candidates", "src_tokens_padded",
                  "encode_ms", "decode_ms"):
            t[k] += getattr(st, k)
        t["src_positions"] += int((summ_h[:, 7].astype(np.int64) * lengths.astype(np.int64)).sum())   # cross-attention keys read
        t["device_model_calls"] = t.get("device_model_calls", 0) + int(st.model_calls)
        t["pool_calls"] = t.get("pool_calls", 0) + 1
        res: list = [None] * n
        for bi in range(n):
            r0, r1 = int(starts[bi]), int(starts[bi + 1])
            rep = replay_beam_batch(tlen_h[r0:r1], summ_h[r0:r1], L, d0, K)
            if rep.error is not None:
                if on_error == "raise":
                    if rep.error == "max_steps":
                        raise RuntimeError("beam-speculative loop exceeded max_steps (non-terminating input)")
                    raise N.ReferenceError_(f"batch {bi}: fewer candidate leaves than n_best for a source (the reference asserts here, "
                                            "speculative_decoding.py:195)")
                self.last_failed_batches.append(bi)
                continue
            res[bi] = out_rows[r0:r1, :, :rep.out_width].contiguous()
            before = self._counter_values()
            self.model_calls_num += rep.model_calls
            self.accepted_tokens_num += rep.accepted_tokens
            self.produced_non_pad_tokens += rep.produced_non_pad_tokens
            if self.smart_drafts_mode:
                self.model_input_lines_num += rep.input_lines
                self.b_sz += rep.running_rows
            else:
                self.n_drafts += (r1 - r0) * self.requested_drafts_num
            self.last_batch_counters[bi] = {k: v - before[k] for k, v in self._counter_values().items()}
            t["batches"] += 1
        return res
