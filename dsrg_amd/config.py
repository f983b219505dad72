"""Model configuration for the DSRG nav-policy network.

Mirrors the attribute set the reference hangs on a HuggingFace PretrainedConfig
(finetune_src/models/vlnbert_init.py:72-108; bert-base-uncased architecture defaults)
and the pre-training JSON (pretrain_src/config/r2r_model_config_objv2.json).
Any object exposing the same attributes (e.g. a transformers config) is accepted by the models.
"""
import json


class DSRGConfig:
    # bert-base-uncased architecture (what PretrainedConfig.from_pretrained('bert-base-uncased') yields)
    _BERT_BASE = dict(
        vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
        intermediate_size=3072, hidden_act='gelu', hidden_dropout_prob=0.1,
        attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
        initializer_range=0.02, layer_norm_eps=1e-12,
    )
    # DSRG fields, vlnbert_init.py:77-108 (values = finetune_src/r2r/parser.py defaults)
    _DSRG = dict(
        max_action_steps=100, image_feat_size=768, angle_feat_size=4, obj_feat_size=768,
        obj_loc_size=3, num_l_layers=9, num_pano_layers=2, num_x_layers=4,
        graph_sprels=True, glocal_fuse=True,
        fix_lang_embedding=False, fix_pano_embedding=False, fix_local_branch=False,
        update_lang_bert=True, output_attentions=True, output_hidden_states=False,
        pred_head_dropout_prob=0.1, use_lang2visn_attn=False, obj_v2_size=128,
        max_instr_len=200, word_max_action=37, word_max_object=44, feat_dropout=0.4,
        obj_name_vocab_size=45, image_dropout_prob=0.1,
        # pre-training only (pretrain_src/config/r2r_model_config_objv2.json)
        image_prob_size=1000, obj_prob_size=0, pretrain_tasks=('mlm', 'mrc', 'sap'),
    )

    def __init__(self, **kw):
        for k, v in {**self._BERT_BASE, **self._DSRG, **kw}.items():
            setattr(self, k, v)

    @classmethod
    def from_json_file(cls, path, **kw):
        with open(path) as f:
            return cls(**{**json.load(f), **kw})

    @classmethod
    def from_args(cls, args):
        """Same derivation as get_vlnbert_models (vlnbert_init.py:77-108)."""
        g = lambda k, d: getattr(args, k, d)
        kw = dict(
            image_feat_size=g('image_feat_size', 768), angle_feat_size=g('angle_feat_size', 4),
            obj_feat_size=g('obj_feat_size', 768), num_l_layers=g('num_l_layers', 9),
            num_pano_layers=g('num_pano_layers', 2), num_x_layers=g('num_x_layers', 4),
            graph_sprels=g('graph_sprels', True), glocal_fuse=g('fusion', 'dynamic') == 'dynamic',
            fix_lang_embedding=g('fix_lang_embedding', False),
            fix_pano_embedding=g('fix_pano_embedding', False),
            fix_local_branch=g('fix_local_branch', False),
            update_lang_bert=not g('fix_lang_embedding', False),
            obj_v2_size=g('obj_v2_size', 128), max_instr_len=g('max_instr_len', 200),
            word_max_action=g('word_max_action', 37), word_max_object=g('word_max_object', 44),
            feat_dropout=g('feat_dropout', 0.4),
        )
        if g('dataset', 'r2r') == 'reverie':
            kw['image_dropout_prob'] = g('image_dropout_prob', 0.1)
        return cls(**kw)

    def to_dict(self):
        return dict(self.__dict__)
