"""The in-tree SNAG MMEA model on the HIP library: encoder + fusion + contrastive objective, as one drop-in module.

Mirrors ``MultiModalEncoder`` (model/SNAG_tools.py:53-156) and ``SNAG`` (model/SNAG.py:18-229): same constructors (``SNAG(kgs, args)``),
same duck-typed driver interface (``model(batch) -> (loss, {"loss_dic", "emb"})``, ``joint_emb_generat``, ``update_noise``,
``Iter_new_links``, ``data_refresh``; main.py:58-75,246-289,359-455), same state_dict keys, so ``Runner`` can swap
``from model import SNAG`` for ``from snag_amd.mmea import SNAG`` and load existing checkpoints.  Every tensor operation of the
forward/backward runs in libsnag_hip.so (projections and Mformer linears on the matrix cores (fp32 in, fp32 accumulate, split-bf16 products), sparse GAT, noise blend,
InfoNCE); torch is used for parameter containers, index bookkeeping and six-scalar loss weighting.

Not built (never exercised by run_snag.sh): GCN structural encoder, ``use_project_head``, ``ial_loss`` (constructed but unused in
the reference, model/SNAG.py:53), ``awloss`` weighting beyond holding its parameters, name/char (surface) modalities' data
readers -- the name/char projections themselves work when features are supplied.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib, evalsim, ops
from ._lib import call, ptr, stream
from .fusion import MformerFusion
from .gat import GAT
from .losses import CustomMultiLossLayer, icl_loss
from .noise import NoiseInjector
from .projection import Linear

f32 = torch.float32


class _RowNormalizeFn(torch.autograd.Function):
    """F.normalize(x) (dim=1, eps 1e-12) via the gather+normalise kernels with the identity index."""

    @staticmethod
    def forward(ctx, x):
        ops._rows(x, "x")
        N, d = x.shape
        idx = torch.arange(N, device=x.device, dtype=torch.int32)
        z = torch.empty(N, d, device=x.device, dtype=f32)
        den = torch.empty(N, device=x.device, dtype=f32)
        call("snag_gather_normalize_fwd", ptr(x), x.stride(0), N, d, ptr(idx), 4, N, 1, ptr(z), ptr(den), stream())
        ctx.save_for_backward(z, den, idx)
        return z

    @staticmethod
    def backward(ctx, dz):
        z, den, idx = ctx.saved_tensors
        N, d = z.shape
        dx = torch.zeros_like(z)
        call("snag_gather_normalize_bwd", ptr(dz.contiguous()), ptr(z), ptr(den), d, ptr(idx), 4, N, 1, ptr(dx), d, stream())
        return dx


def normalize_rows(x):
    return _RowNormalizeFn.apply(x)          # a row-strided view (one token slot of hidden) is read in place


class _EntityBlendFn(torch.autograd.Function):
    """e[mask] = (1 - a) e[mask] + a noise[mask], a = mask_ratio / 2 (model/SNAG_tools.py:127-128); noise is a constant."""

    @staticmethod
    def forward(ctx, e, noise, mask_u8, a):
        N, D = e.shape
        out = torch.empty_like(e)
        call("snag_entity_token_fwd", ptr(e), N, D, ptr(noise), ptr(mask_u8), float(a), ptr(out), D, stream())
        ctx.save_for_backward(mask_u8)
        ctx.a = a
        return out

    @staticmethod
    def backward(ctx, d_out):
        (mask_u8,) = ctx.saved_tensors
        d_out = d_out.contiguous()
        N, D = d_out.shape
        d_e = torch.empty_like(d_out)
        call("snag_entity_token_bwd", ptr(d_out), D, N, D, ptr(mask_u8), float(ctx.a), ptr(d_e), 0, stream())
        return d_e, None, None, None


class AutomaticWeightedLoss(nn.Module):
    """Parameter holder + formula of model/Tool_model.py:14-39 (only used when ``--awloss 1``)."""

    def __init__(self, num=2):
        super().__init__()
        self.params = nn.Parameter(torch.ones(num, requires_grad=True))

    def forward(self, *x):
        loss_sum = 0
        for i, loss in enumerate(x):
            loss_sum = loss_sum + 0.5 / (self.params[i] ** 2) * loss + torch.log(1 + self.params[i] ** 2)
        return loss_sum


class MultiModalEncoder(nn.Module):
    """HIP drop-in for model/SNAG_tools.py:53-156."""

    def __init__(self, args, ent_num, img_feature_dim, char_feature_dim=None, use_project_head=False, attr_input_dim=1000):
        super().__init__()
        if use_project_head:
            raise NotImplementedError("use_project_head is not wired in the reference's SNAG encoder either (SNAG_tools.py:75)")
        self.args = args
        self.ENT_NUM = ent_num
        self.n_units = [int(x) for x in args.hidden_units.strip().split(",")]
        self.n_heads = [int(x) for x in args.heads.strip().split(",")]
        self.input_dim = self.n_units[0]
        self.entity_emb = nn.Embedding(ent_num, self.input_dim)                                    # :84-86
        nn.init.normal_(self.entity_emb.weight, std=1.0 / math.sqrt(ent_num))
        self.rel_fc = Linear(1000, args.attr_dim)                                                  # :92-96, same construction order
        self.att_fc = Linear(attr_input_dim, args.attr_dim)
        self.img_fc = Linear(img_feature_dim, args.img_dim)
        self.name_fc = Linear(300, args.char_dim)
        self.char_fc = Linear(char_feature_dim, args.char_dim)
        if args.structure_encoder == "gcn":                                                        # :98-100
            from .gcn import GCN
            self.cross_graph_model = GCN(self.n_units[0], self.n_units[1], self.n_units[2], dropout=args.dropout)
        elif args.structure_encoder != "gat":
            raise NotImplementedError(f"structure_encoder {args.structure_encoder!r} is not a reference option")
        else:
            self.cross_graph_model = GAT(n_units=self.n_units, n_heads=self.n_heads, dropout=args.dropout, attn_dropout=args.attn_dropout,
                                         instance_normalization=args.instance_normalization, diag=True)   # :101-104
        self.fusion = MformerFusion(args, modal_num=args.inner_view_num, with_weight=args.with_weight)  # :105-106

    def forward(self, input_idx, adj, img_features=None, rel_features=None, att_features=None, name_features=None, char_features=None,
                ent_wo_img=None, entity_noise=None, entity_noise_mask=None, _test=False, drop_masks=None):
        a = self.args
        if a.w_gcn:                                                                                # :122-129
            ent = nn.functional.embedding(input_idx, self.entity_emb.weight)
            if entity_noise is not None and entity_noise_mask is not None:
                ent = _EntityBlendFn.apply(ent.contiguous(), entity_noise, entity_noise_mask.to(torch.uint8).contiguous(), a.mask_ratio * 0.5)
            gph_emb = self.cross_graph_model(ent, adj)
        else:
            gph_emb = None
        img_emb = self.img_fc(img_features) if a.w_img else None                                  # :133-152
        rel_emb = self.rel_fc(rel_features) if a.w_rel else None
        att_emb = self.att_fc(att_features) if a.w_attr else None
        name_emb = self.name_fc(name_features) if (a.w_name and name_features is not None) else None
        char_emb = self.char_fc(char_features) if (a.w_char and char_features is not None) else None
        joint_emb, joint_emb_fz, hidden_states, weight_norm = self.fusion([img_emb, att_emb, rel_emb, gph_emb, name_emb, char_emb],
                                                                          drop_masks=drop_masks)  # :154
        return gph_emb, img_emb, rel_emb, att_emb, name_emb, char_emb, joint_emb, joint_emb_fz, hidden_states, weight_norm


class SNAG(nn.Module):
    """HIP drop-in for model/SNAG.py:18-229."""

    def __init__(self, kgs, args, device="cuda"):
        super().__init__()
        _lib.require_gpu()
        self.kgs, self.args = kgs, args
        dev = torch.device(device)
        img = torch.as_tensor(np.asarray(kgs["images_list"]), dtype=f32)
        self.img_features = (img / img.norm(dim=1, keepdim=True).clamp_min(1e-12)).to(dev)        # F.normalize at load (:23)
        self.input_idx = kgs["input_idx"].to(dev)
        self.adj = kgs["adj"].to(dev)
        self.rel_features = torch.as_tensor(np.asarray(kgs["rel_features"]), dtype=f32).to(dev)
        self.att_features = torch.as_tensor(np.asarray(kgs["att_features"]), dtype=f32).to(dev)
        self.ent_wo_img = torch.as_tensor(kgs["ent_wo_img"]).to(dev)
        self.ent_w_img = torch.as_tensor(kgs["ent_w_img"]).to(dev)
        self.name_features = self.char_features = None
        self.train_ill = kgs["train_ill"]
        self.args.modal_num = 3
        if kgs["name_features"] is not None:
            self.name_features = kgs["name_features"].to(dev)
            self.char_features = kgs["char_features"].to(dev)
            self.args.modal_num = 4
        char_dim = kgs["char_features"].shape[1] if self.char_features is not None else 100
        self.multimodal_encoder = MultiModalEncoder(args=self.args, ent_num=kgs["ent_num"], img_feature_dim=self.img_features.shape[1],
                                                    char_feature_dim=char_dim, use_project_head=self.args.use_project_head,
                                                    attr_input_dim=self.att_features.shape[1])
        self.multi_loss_layer = CustomMultiLossLayer(loss_num=6)
        self.multi_loss_layer_2 = AutomaticWeightedLoss(num=7)
        self.criterion_cl = icl_loss(tau=self.args.tau, ab_weight=self.args.ab_weight, n_view=2)
        self.criterion_cl_joint = icl_loss(tau=self.args.tau, ab_weight=self.args.ab_weight, n_view=2, neg_cross_kg=self.args.neg_cross_kg)
        self.noise_mode = "parity"             # "philox" keeps the per-epoch noise draw on the device
        self._noise = None
        self.to(dev)
        if self.args.add_noise:
            self.get_mean_std()

    # ---- noise (model/SNAG.py:66-98) -------------------------------------------------------------------------------------------
    def get_mean_std(self):
        self._noise = NoiseInjector({"rel": self.rel_features, "att": self.att_features, "img": self.img_features},
                                    self.args.noise_ratio, self.args.mask_ratio, exclude_rows={"img": self.ent_wo_img.cpu().tolist()},
                                    mode=self.noise_mode)
        (self.img_mean, self.img_std), (self.rel_mean, self.rel_std), (self.att_mean, self.att_std) = (
            self._noise.stats["img"], self._noise.stats["rel"], self._noise.stats["att"])

    def update_noise(self, seed=0):
        self._noise.mode = self.noise_mode
        noisy = self._noise.update_noise(entity_weight=self.multimodal_encoder.entity_emb.weight, seed=seed)
        self.rel_noisy_features, self.att_noisy_features, self.img_noisy_features = noisy["rel"], noisy["att"], noisy["img"]
        self.entity_noise, self.entity_noise_mask = self._noise.entity_noise, self._noise.entity_noise_mask

    # ---- forward (model/SNAG.py:101-162) ---------------------------------------------------------------------------------------
    def forward(self, batch, drop_masks=None):
        (gph_emb, img_emb, rel_emb, att_emb, name_emb, char_emb, joint_emb, joint_emb_fz, hidden_states,
         weight_norm) = self.joint_emb_generat(only_joint=False, drop_masks=drop_masks)
        hid = self.generate_hidden_emb(hidden_states)
        GMI_loss = self.criterion_cl_joint(joint_emb, batch) + self.criterion_cl_joint(joint_emb_fz, batch)          # :106
        ECIA_loss = self.inner_view_loss(gph_emb, rel_emb, att_emb, img_emb, name_emb, char_emb, batch, weight_norm=weight_norm)   # :109
        IIR_loss = self.inner_view_loss(*hid[:6], batch)                                                              # :112
        loss_list = [GMI_loss, ECIA_loss, IIR_loss]
        loss_all = self.multi_loss_layer_2(*loss_list) if self.args.awloss else sum(loss_list)
        loss_dic = {"joint_Intra_modal": GMI_loss.item(), "Intra_modal": ECIA_loss.item(), "IIR_loss": IIR_loss.item()}
        return loss_all, {"loss_dic": loss_dic, "emb": joint_emb}

    def generate_hidden_emb(self, hidden):
        """:124-141 -- including the reference's slot labelling (slot 0 is called gph although the stack order is img, att, rel, gph)."""
        nm = lambda k: normalize_rows(hidden[:, k, :])
        gph_emb, rel_emb, att_emb = nm(0), nm(1), nm(2)
        img_emb = nm(3) if self.args.w_img else None
        name_emb = char_emb = None
        if hidden.shape[1] >= 6:
            name_emb, char_emb = nm(4), nm(5)
        return gph_emb, rel_emb, att_emb, img_emb, name_emb, char_emb

    def inner_view_loss(self, gph_emb, rel_emb, att_emb, img_emb, name_emb, char_emb, train_ill, weight_norm=None):
        embs = (gph_emb, rel_emb, att_emb, img_emb, name_emb, char_emb)
        cols = (3, 2, 1, 0, 4, 5)                                                                                     # :147-152
        losses = []
        wn = weight_norm * weight_norm.shape[1] if weight_norm is not None else None                                   # :145-146
        for e, c in zip(embs, cols):
            if e is None:
                losses.append(0)
            elif wn is not None:
                losses.append(self.criterion_cl(e, train_ill, weight_norm=wn[:, c]))
            else:
                losses.append(self.criterion_cl(e, train_ill))
        return self.multi_loss_layer(losses)                                                                          # :160

    def joint_emb_generat(self, only_joint=True, test=False, drop_masks=None):
        enc = self.multimodal_encoder
        if self.args.add_noise and enc.training:                                                                      # :168-170
            outs = enc(self.input_idx, self.adj, self.img_noisy_features, self.rel_noisy_features, self.att_noisy_features,
                       self.name_features, self.char_features, ent_wo_img=self.ent_wo_img, entity_noise=self.entity_noise,
                       entity_noise_mask=self.entity_noise_mask, _test=test, drop_masks=drop_masks)
        else:
            outs = enc(self.input_idx, self.adj, self.img_features, self.rel_features, self.att_features, self.name_features,
                       self.char_features, ent_wo_img=self.ent_wo_img, _test=test, drop_masks=drop_masks)
        if only_joint:
            return outs[7], outs[9]                                                                                   # joint_emb_fz, weight_norm
        return outs

    # ---- iterative learning (model/SNAG.py:192-229) ----------------------------------------------------------------------------
    def Iter_new_links(self, epoch, left_non_train, final_emb, right_non_train, new_links=[]):
        if len(left_non_train) == 0 or len(right_non_train) == 0:
            return new_links
        _, _, mutual = evalsim.mutual_nearest(final_emb, left_non_train, right_non_train)
        if (epoch + 1) % (self.args.semi_learn_step * 5) == self.args.semi_learn_step:
            return mutual
        keep = set(map(tuple, new_links))
        return [p for p in mutual if p in keep]

    def data_refresh(self, logger, train_ill, test_ill_, left_non_train, right_non_train, new_links=[]):
        if len(new_links) != 0 and len(left_non_train) != 0 and len(right_non_train) != 0:
            train_ill = np.vstack((train_ill, np.array(new_links)))
            test_set = set(map(tuple, np.asarray(test_ill_).tolist()))
            num_true = sum(1 for nl in new_links if tuple(nl) in test_set)
            for nl in new_links:
                left_non_train.remove(nl[0])
                right_non_train.remove(nl[1])
            logger.info(f"#new_links_select:{len(new_links)}")
            logger.info(f"train_ill.shape:{train_ill.shape}")
            logger.info(f"#true_links: {num_true}")
            logger.info(f"true link ratio: {(100 * num_true / len(new_links)):.1f}%")
            logger.info(f"#entity not in train set: {len(left_non_train)} (left) {len(right_non_train)} (right)")
            new_links = []
        else:
            logger.info("len(new_links) is 0")
        return left_non_train, right_non_train, train_ill, new_links
