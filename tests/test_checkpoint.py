"""not-gpu: checkpoint compatibility.  The module tree only HOLDS parameters, so it is built on CPU.

  * pre-training -> fine-tune hand-off (finetune_src/models/vlnbert_init.py:58-66,110-113): a PT state dict saved with DDP's
    'module.' prefixes and bare '*_head' / 'sap_fuse*' keys is loaded through get_vlnbert_models(args.bert_ckpt_file):
    every tensor the two models share must land, PT-only keys are dropped, fine-tune-only modules (RMF, IGL) keep
    their initialisation, and a shape mismatch is an error, not a silent skip;
  * agent snapshots (finetune_src/r2r/agent_base.py:225-274): {'vln_bert': {'epoch', 'state_dict', 'optimizer'}, 'critic': ...}
    round trip with and without 'module.' prefixes and with extra keys."""
from types import SimpleNamespace

import pytest
import torch

from dsrg_amd.config import DSRGConfig
from dsrg_amd.model import Critic, VLNBert, get_vlnbert_models, load_agent_checkpoint
from dsrg_amd.pretrain import GlocalTextPathCMTPreTraining

SMALL = dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, vocab_size=512, num_l_layers=2, num_x_layers=2,
             num_pano_layers=1, image_feat_size=128, image_prob_size=16, max_position_embeddings=64)


def _args(**kw):
    return SimpleNamespace(**{**dict(model='r2r', dataset='r2r', bert_ckpt_file=None, feat_dropout=0.4, fusion='dynamic', graph_sprels=True), **kw})


def test_pretrain_to_finetune_remap_lands_every_shared_tensor(tmp_path):
    torch.manual_seed(3)
    pt = GlocalTextPathCMTPreTraining(DSRGConfig(use_lang2visn_attn=True, obj_feat_size=0, pretrain_tasks=('mlm', 'mrc', 'sap'), **SMALL))
    with torch.no_grad():
        for p in pt.parameters():
            p.copy_(torch.randn_like(p))
    src = pt.state_dict()
    path = str(tmp_path / 'model_step_best_1.pt')
    torch.save({'module.' + k: v for k, v in src.items()}, path)          # as saved from under DDP
    ft_cfg = DSRGConfig(obj_feat_size=0, **SMALL)
    torch.manual_seed(4)
    ft = get_vlnbert_models(_args(bert_ckpt_file=path), config=ft_cfg)
    own = ft.state_dict()
    landed, pt_only = 0, []
    for k, v in src.items():
        name = k[5:] if k.startswith('bert.') else k          # heads are bare in the PT file, 'bert.'-prefixed by the remap, stripped on load
        if name in own:
            assert torch.equal(own[name], v), name
            landed += 1
        else:
            pt_only.append(name)
    assert landed >= 0.8 * len(own)
    assert any(k.startswith('global_sap_head.') for k in own) and any(k.startswith('sap_fuse_linear.') for k in own)
    assert all(k.split('.')[0] in ('mlm_head', 'image_classifier') or '.lang_' in k or 'instr_act_encoder' in k or 'instr_obj_encoder' in k
               for k in pt_only), pt_only[:5]
    # fine-tune-only modules are not in the PT file: they keep their (random) initialisation
    torch.manual_seed(4)
    fresh = get_vlnbert_models(_args(), config=ft_cfg).state_dict()
    for k in ('gmap_pooler.dense.weight', 'local_his_map.weight', 'lang_encoder.instr_actobj_encoder.query.weight'):
        assert torch.equal(own[k], fresh[k]), k
    assert not ft.training                                       # from_pretrained returns .eval()


def test_shape_mismatch_is_loud(tmp_path):
    pt = GlocalTextPathCMTPreTraining(DSRGConfig(use_lang2visn_attn=True, obj_feat_size=0, pretrain_tasks=('sap',), **SMALL))
    path = str(tmp_path / 'pt.pt')
    torch.save(pt.state_dict(), path)
    wrong = DSRGConfig(obj_feat_size=0, **{**SMALL, 'intermediate_size': 384})
    with pytest.raises(RuntimeError, match='size mismatch'):
        get_vlnbert_models(_args(bert_ckpt_file=path), config=wrong)


@pytest.mark.parametrize('prefix', ['', 'module.'])
def test_agent_snapshot_round_trip(tmp_path, prefix, monkeypatch):
    import dsrg_amd.model as M
    cfg = DSRGConfig(obj_feat_size=0, **SMALL)
    monkeypatch.setattr(M.DSRGConfig, 'from_args', classmethod(lambda cls, a: cfg))
    torch.manual_seed(5)
    a, ca = VLNBert(_args()), Critic(_args())
    with torch.no_grad():
        for p in list(a.parameters()) + list(ca.parameters()):
            p.copy_(torch.randn_like(p))
    opt = torch.optim.AdamW(a.parameters(), lr=1e-5)
    snap = {'vln_bert': {'epoch': 7, 'state_dict': {prefix + k: v for k, v in a.state_dict().items()}, 'optimizer': opt.state_dict()},
            'critic': {'epoch': 7, 'state_dict': {prefix + k: v for k, v in ca.state_dict().items()}, 'optimizer': {}}}
    snap['vln_bert']['state_dict'][prefix + 'vln_bert.some_removed_layer.weight'] = torch.zeros(3)       # an extra key
    path = str(tmp_path / 'best_val_unseen')
    torch.save(snap, path)
    b, cb = VLNBert(_args()), Critic(_args())
    epoch, extra = load_agent_checkpoint({'vln_bert': b, 'critic': cb}, path)
    assert epoch == 6 and extra['vln_bert'] == ['vln_bert.some_removed_layer.weight']
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    for (k, x), (_, y) in zip(ca.state_dict().items(), cb.state_dict().items()):
        assert torch.equal(x, y), k


def test_weight_plan_views_follow_the_grad_mode_of_each_resolve():
    """A weight plan is built once and resolved every step.  Reshaped weights (the [1,H] score layers) must be re-viewed at
    every resolve: a view taken while the plan was built would freeze the grad mode of that first call (no gradient when the
    first forward ran under no_grad) and would keep the parameter's AccumulateGrad node alive for the plan's lifetime."""
    from dsrg_amd import functional as Fn
    w32 = torch.nn.Parameter(torch.randn(1, 8))
    w16 = torch.nn.Parameter(torch.randn(1, 8))
    pk = Fn.WeightPacker()
    W0 = {'a': pk.view(w32, (-1,)), 'b': pk.view(pk.cast(w16, torch.bfloat16), (-1,))}
    with torch.no_grad():
        W = pk.resolve(dict(W0))
    assert W['a'].grad_fn is None and W['b'].grad_fn is None and W['a'].shape == (8,) and W['b'].dtype == torch.bfloat16
    W = pk.resolve(dict(W0))
    assert W['a'].grad_fn is not None and W['b'].grad_fn is not None
    (W['a'].sum() + W['b'].float().sum()).backward()
    assert torch.equal(w32.grad, torch.ones(1, 8)) and torch.equal(w16.grad, torch.ones(1, 8))
    # nothing the plan holds (placeholders, packer sources) carries an autograd node of a parameter
    assert all(not isinstance(v, torch.Tensor) or v.grad_fn is None for v in list(W0.values()) + pk.srcs)
