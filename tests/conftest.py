"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import sys
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def trained_full_state():
    """Full-size (d=256, 8 heads, FFN 2048) weights overfit on the 10 fixture pairs, trained ONCE per test session and depth
    (stock torch training on the GPU: set-up only).  Returns get(n_layers) -> state dict (CPU tensors)."""
    cache = {}

    def get(n_layers: int) -> dict:
        if n_layers in cache:
            return cache[n_layers]
        import torch
        from tools.train_synth import TrainModel
        from util_models import fixture_tokens
        src, tgt, _, V = fixture_tokens()
        torch.manual_seed(1234 if n_layers == 4 else 4321)
        model = TrainModel(vocab=V, n_enc=n_layers, n_dec=n_layers).cuda()
        opt = torch.optim.Adam(model.parameters(), lr=3e-4 if n_layers == 4 else 2e-4)
        crit = torch.nn.CrossEntropyLoss()
        s, t = src.cuda(), tgt.cuda()
        model.train()
        t0 = time.time()
        for step in range(900):
            loss = crit(model(s, t[:, :-1]).reshape(-1, V), t[:, 1:].reshape(-1))
            opt.zero_grad()
            loss.backward()
            opt.step()
            if loss.item() < 5e-3:
                break
        print(f"full-size {n_layers}+{n_layers} fixture model: steps", step, "loss", loss.item(), f"({time.time() - t0:.1f} s)")
        assert loss.item() < 0.05
        cache[n_layers] = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        return cache[n_layers]

    return get
