"""REVERIE pre-training model: drop-in for `GlocalTextPathCMTPreTraining` of pretrain_src/model/pretrain_cmt_reverie.py
over `GlocalTextPathCMT` of pretrain_src/model/vilmodel_reverie.py.

What differs from the R2R twin (dsrg_amd/pretrain.py), all of it in the trajectory embedding (vilmodel_reverie.py:520-593):
  * the ASV object embedding is fused into each view BEFORE the panorama encoder (:547-548);
  * every step's 0..20 768-d "objv1" object tokens go through obj_reverie_linear + LayerNorm and are appended to that
    step's views (:551-564), so the panorama encoder, the GAA pooling (over all padded slots, :586-591) and the viewpoint
    encoder run over views + objects;
  * the language encoder is the plain 9-layer BERT (no instr_* parameters, :436-457);
  * tasks: mlm, mrc with the object-region branch (pretrain_cmt.py:202-226), sap, og (:296-327), valid_sap_og (:329-392).
That trajectory embedding is arithmetic-for-arithmetic the fine-tune REVERIE panorama mode, so it IS that code path
(dsrg_amd/vilmodel_reverie.py); state_dict keys match the reference's (tests/golden/reverie_pt_state_dict_keys.txt)."""
import torch

from .pretrain import GlocalTextPathCMT as _R2RBackbone, GlocalTextPathCMTPreTraining as _R2RPreTraining
from .vilmodel_reverie import GlocalTextPathNavCMT as _ReverieNav


class GlocalTextPathCMT(_R2RBackbone, _ReverieNav):
    PRETRAIN, REVERIE = True, True

    def _trajectory(self, b):
        V = b['traj_view_img_fts'].shape[1]
        obj, ol = b.get('traj_obj_img_fts'), b.get('traj_vp_obj_lens')
        pano, masks, fused = self.forward_panorama_per_step(
            b['traj_view_img_fts'], b['traj_objv2_fts'], b['traj_loc_fts'], b['traj_nav_types'], b['traj_vp_view_lens'],
            b['traj_objv2_lens'], obj_img_fts=obj, obj_lens=ol, objv2_nav_types=b['traj_nav_types'][:, :V])
        lens = b['traj_vp_view_lens'] if obj is None else b['traj_vp_view_lens'] + ol
        return pano, fused, lens


class GlocalTextPathCMTPreTraining(_R2RPreTraining):
    BACKBONE = GlocalTextPathCMT
