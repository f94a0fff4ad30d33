import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import translation_transformer_amd as tta, bench
from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK, V
bc = bench.BEAM_CONFIGS["c4"]
sd = bench.get_weights(1500, "cuda:0", bc["kind"], bc["layers"], {})
os.environ["TTX_HOST_TIMING"] = "1"
model = tta.NativeTransformer(sd, num_heads=8, pad_token_idx=PAD, device=0)
src_all, _ = SynthReactions(123456, bc["kind"]).dataset(66 * 8)
bs = [torch.from_numpy(b).cuda() for b in batches(src_all, 8)]
for smart in (False, True):
    mk = lambda: tta.TranslationInferenceBeamSearchSpeculative(model, 200, 10, 10, 2, V, smart, PAD, BOS, EOS, C_TOK, max_steps=800)
    mk().generate_many(bs[:8], in_flight=8)
    print("=== smart", smart, file=sys.stderr, flush=True)
    g = mk(); torch.cuda.synchronize(); t0 = time.perf_counter(); g.generate_many(bs[2:], in_flight=8); torch.cuda.synchronize()
    print("=== done smart", smart, (time.perf_counter() - t0) * 1e3, "ms", g.stats_total, file=sys.stderr, flush=True)
