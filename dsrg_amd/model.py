"""`VLNBert` — the module the reference agents instantiate (finetune_src/models/model.py:12-41).

Same constructor argument (the parsed `args` namespace), same forward(mode, batch): the batch is wrapped in a
defaultdict(None) so absent keys read as None, 'panorama' applies feature dropout to view_img_fts (and to
obj_img_fts for REVERIE), unknown modes raise NotImplementedError."""
import collections
import torch
import torch.nn as nn

from .config import DSRGConfig


def get_vlnbert_models(args, config=None, state_dict=None, compute_dtype=None):
    """Counterpart of vlnbert_init.get_vlnbert_models (vlnbert_init.py:14-115) without the by-name hub
    fetches: the config is derived from args locally; a checkpoint path in args.bert_ckpt_file is loaded
    with the reference's key remap ('module.' stripped; '_head' / 'sap_fuse' keys get 'bert.')."""
    if getattr(args, 'model', 'r2r') == 'reverie':
        from .vilmodel_reverie import GlocalTextPathNavCMT
    else:
        from .vilmodel import GlocalTextPathNavCMT
    cfg = config or DSRGConfig.from_args(args)
    ckpt = getattr(args, 'bert_ckpt_file', None)
    new_sd = dict(state_dict or {})
    if ckpt not in (None, 'bert', 'lxmert') and not new_sd:
        for k, v in torch.load(ckpt, map_location='cpu', weights_only=True).items():
            if k.startswith('module'):
                k = k[7:]
            new_sd[('bert.' + k) if ('_head' in k or 'sap_fuse' in k) else k] = v
    kw = {} if compute_dtype is None else {'compute_dtype': compute_dtype}
    return GlocalTextPathNavCMT.from_pretrained(None, config=cfg, state_dict=new_sd, **kw)


def load_agent_checkpoint(models, path, strict_shapes=True):
    """Counterpart of Seq2SeqAgent.load (finetune_src/r2r/agent_base.py:242-274) for the parameter part: `models` maps the
    snapshot's names ('vln_bert', 'critic') to modules; keys are matched tolerantly (a 'module.' prefix on either side is
    fixed, extra keys in the file are reported and skipped, keys missing from the file keep the module's values).  The file
    is read with weights_only=True.  Returns (epoch - 1, {name: [extra keys]}) like the reference returns the epoch."""
    states = torch.load(path, map_location='cpu', weights_only=True)
    extra = {}
    for name, model in models.items():
        state = model.state_dict()
        model_keys = set(state.keys())
        sd = states[name]['state_dict']
        if model_keys != set(sd.keys()):
            mk0, lk0 = next(iter(model_keys)), next(iter(sd.keys()))
            if not mk0.startswith('module.') and lk0.startswith('module.'):
                sd = {k.replace('module.', ''): v for k, v in sd.items()}
            if mk0.startswith('module.') and not lk0.startswith('module.'):
                sd = {'module.' + k: v for k, v in sd.items()}
            extra[name] = [k for k in sd if k not in model_keys]
            sd = {k: v for k, v in sd.items() if k in model_keys}
        if strict_shapes:
            bad = [k for k, v in sd.items() if tuple(v.shape) != tuple(state[k].shape)]
            if bad:
                raise ValueError('load_agent_checkpoint: size mismatch for %s' % ', '.join(bad[:8]))
        state.update(sd)
        model.load_state_dict(state)
    return states['vln_bert']['epoch'] - 1, extra


class VLNBert(nn.Module):
    def __init__(self, args, compute_dtype=None):
        super().__init__()
        self.args = args
        self.vln_bert = get_vlnbert_models(args, config=None, compute_dtype=compute_dtype)
        self.drop_env = nn.Dropout(p=getattr(args, 'feat_dropout', 0.4))

    def forward(self, mode, batch):
        batch = collections.defaultdict(lambda: None, batch)
        if mode == 'language':
            return self.vln_bert(mode, batch)
        elif mode == 'panorama':
            batch['view_img_fts'] = self.drop_env(batch['view_img_fts'])
            if getattr(self.args, 'dataset', 'r2r') == 'reverie' and 'obj_img_fts' in batch:
                batch['obj_img_fts'] = self.drop_env(batch['obj_img_fts'])
            return self.vln_bert(mode, batch)
        elif mode == 'navigation':
            return self.vln_bert(mode, batch)
        raise NotImplementedError('wrong mode: %s' % mode)


class Critic(nn.Module):
    """768 -> 512 -> 1 value head (model.py:44-55); built by the agents, never used in a loss."""

    def __init__(self, args):
        super().__init__()
        self.state2value = nn.Sequential(nn.Linear(768, 512), nn.ReLU(), nn.Dropout(getattr(args, 'dropout', 0.5)),
                                         nn.Linear(512, 1))

    def forward(self, state):
        return self.state2value(state).squeeze()
