"""not-gpu: the CPU oracle (oracle/dsrg_oracle.py) reproduces the golden vectors that the
reference's own model files produced (tools/make_golden.py). Tolerance 1e-4 max-abs (fp32 vs fp32,
different op order only); BASELINE north_star asks 1e-3 on logits."""
import pytest
import torch

from dsrg_amd.config import DSRGConfig
from oracle import dsrg_oracle
from tests import common


@pytest.mark.parametrize('tag,reverie', [('r2r_c1', False), ('reverie_c3', True)])
def test_oracle_matches_reference_golden(tag, reverie):
    torch.set_num_threads(8)
    gold = common.load_gold(tag)
    sd = common.synth_sd(tag)
    cfg = DSRGConfig()
    fwd = lambda mode, b: dsrg_oracle.forward(sd, cfg, mode, b, reverie=reverie)
    out = common.run_chain(fwd, gold['meta'])
    errs = common.compare(out, gold, tol=1e-4)
    print(tag, {k: '%.1e' % v for k, v in errs.items()})
