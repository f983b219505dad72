import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dsrg_amd import synth
from dsrg_amd.config import DSRGConfig
from dsrg_amd.pretrain_reverie import GlocalTextPathCMTPreTraining
from tests import common
from tests.test_oracle_pt_golden import PT_CFG_REVERIE
gold = common.load_gold('reverie_pt')
B, L, O, seed = [int(x) for x in gold['meta']]
cfg = DSRGConfig(**PT_CFG_REVERIE)
for order in (('mlm', 'mrc', 'sap', 'og'), ('og',), ('sap', 'og'), ('mrc', 'og'), ('mlm', 'og')):
    m = GlocalTextPathCMTPreTraining(cfg, compute_dtype=torch.bfloat16)
    m.load_state_dict(common.synth_sd('reverie_pt'))
    m = m.cuda().train()
    named = dict(m.named_parameters())
    for task in order:
        b = common.to_t(synth.pretrain_batch(B, L, O, seed, task, reverie=True), 'cuda')
        m.zero_grad()
        loss = m(b, task, compute_loss=True)
        loss.mean().backward()
        worst = (0, None)
        for k, gg in gold.items():
            if k.startswith(task + '_grad:'):
                mine = named[k.split(':', 1)[1]].grad.reshape(-1)[:64].float().cpu().numpy()
                r = np.abs(mine - gg).max() / (np.abs(gg).max() + 1e-12)
                if r > worst[0]: worst = (r, k)
        print(order, task, 'worst rel', worst, flush=True)
        with torch.no_grad():
            m(b, task, compute_loss=False)
