import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """libivit.so, built in-tree if the prebuilt one is stale or absent (hipcc cross-compiles)."""
    from interactive_vit_amd import build
    return build.build()


@pytest.fixture(autouse=True)
def oracle_mirrors_engine_rounding_points():
    """The oracle's rounding-aware mode mirrors the rounding points of the engine build under test: LayerNorm
    folded into the next GEMM (the bf16 default, IVIT_FOLD_LN unset or 1) or a LayerNorm kernel (IVIT_FOLD_LN=0).
    Engine.ln_fold reports what an engine actually does; the GPU tests assert that it agrees with this."""
    from oracle import vit_oracle
    import torch
    vit_oracle.LN_FOLD = os.environ.get("IVIT_FOLD_LN", "1") != "0"
    vit_oracle.OPERAND_DTYPE = torch.bfloat16
    yield
    vit_oracle.LN_FOLD = False
    vit_oracle.OPERAND_DTYPE = torch.bfloat16
