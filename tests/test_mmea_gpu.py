"""GPU parity: the whole in-tree SNAG model (encoder + GAT + fusion + contrastive losses) vs golden G10 (reference SNAG.forward)."""
import types

import numpy as np
import pytest
import torch

from helpers import T, assert_close, load

pytestmark = pytest.mark.gpu
DEV = "cuda"


def snag_args(D=64, add_noise=1):
    return types.SimpleNamespace(
        attr_dim=D, img_dim=D, name_dim=D, char_dim=D, dropout=0.0, attn_dropout=0.0, hidden_units=f"{D},{D},{D}", heads="2,2",
        structure_encoder="gat", instance_normalization=False, inner_view_num=4, with_weight=1,
        w_gcn=True, w_img=True, w_rel=True, w_attr=True, w_name=False, w_char=False, use_project_head=False,
        hidden_size=D, num_attention_heads=1, num_hidden_layers=1, use_intermediate=1, intermediate_size=96,
        add_noise=add_noise, noise_ratio=0.2, mask_ratio=0.7, tau=0.1, tau2=4.0, ab_weight=0.5, neg_cross_kg=0, zoom=0.1,
        reduction="mean", batch_size=60, awloss=0, semi_learn_step=5, modal_num=3)


def build(z):
    from snag_amd.mmea import SNAG
    N = int(z["N"])
    adj = torch.sparse_coo_tensor(T(z["edges"]), torch.ones(z["edges"].shape[1]), (N, N))
    wo = z["ent_wo_img"].tolist()
    kgs = {"ent_num": N, "images_list": z["images_list"], "rel_features": z["rel_features"], "att_features": z["att_features"],
           "name_features": None, "char_features": None, "input_idx": torch.arange(N), "ent_wo_img": wo,
           "ent_w_img": [i for i in range(N) if i not in set(wo)], "adj": adj, "train_ill": z["train_ill"]}
    model = SNAG(kgs, snag_args(int(z["D"])), device=DEV)
    sd = {k[2:]: T(v) for k, v in z.items() if k.startswith("p.")}
    assert set(model.state_dict()) == set(sd), sorted(set(model.state_dict()) ^ set(sd))        # checkpoint compatibility
    model.load_state_dict(sd, strict=True)
    return model


def test_snag_model_golden():
    z = load("g10_snag_model.npz")
    model = build(z)
    # eval: the embedding Runner._test consumes (main.py:376)
    model.eval()
    with torch.no_grad():
        final, wn = model.joint_emb_generat()
    assert_close(final, z["eval.joint_emb_fz"], atol=2e-5, rtol=1e-5)
    assert_close(wn, z["eval.weight_norm"], atol=2e-5)
    # train: update_noise() replays the reference's RNG stream, then one forward/backward with the recorded Bert dropout masks
    model.train()
    torch.manual_seed(int(z["noise_seed"]))
    model.update_noise()
    for nm in ("img", "rel", "att"):
        assert_close(getattr(model, f"{nm}_noisy_features"), z[f"{nm}_noisy"], atol=2e-6, what=f"{nm}_noisy")
    assert_close(model.entity_noise, z["entity_noise"], atol=1e-6)
    assert np.array_equal(model.entity_noise_mask.cpu().numpy().astype(bool), z["entity_noise_mask"])
    masks = [{"probs": T(z["mask.probs"], DEV), "attn_out": T(z["mask.attn_out"], DEV), "ffn_out": T(z["mask.ffn_out"], DEV)}]
    loss, out = model(z["train_ill"], drop_masks=masks)
    assert_close(loss, z["train.loss"], atol=2e-3, rtol=2e-5)
    for k in ("joint_Intra_modal", "Intra_modal", "IIR_loss"):
        assert abs(out["loss_dic"][k] - float(z["train.loss." + k])) < 1e-3 * max(1.0, abs(float(z["train.loss." + k]))), k
    assert_close(out["emb"], z["train.joint_emb"], atol=2e-5)
    loss.backward()
    checked = 0
    for k, p in model.named_parameters():
        key = "g.p." + k
        if key in z:
            ref = z[key]
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            assert_close(g, ref, atol=3e-4 * max(1.0, float(np.abs(ref).max())), rtol=2e-3, what=key)
            checked += 1
    assert checked >= 25, checked


def test_snag_model_trains_and_evaluates():
    """A few optimiser steps on the golden KG lower the loss; the eval path returns sane Hits@k; IL returns mutual pairs."""
    from snag_amd import evalsim
    z = load("g10_snag_model.npz")
    model = build(z)
    model.noise_mode = "philox"
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3)
    model.train()
    model.update_noise(seed=1)
    first = None
    for it in range(12):
        loss, _ = model(z["train_ill"])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        first = float(loss.detach()) if first is None else first
    assert float(loss.detach()) < first
    model.eval()
    with torch.no_grad():
        final, _ = model.joint_emb_generat()
        from snag_amd.mmea import normalize_rows
        final = normalize_rows(final)
        left, right = T(z["train_ill"][:, 0].astype(np.int64), DEV), T(z["train_ill"][:, 1].astype(np.int64), DEV)
        res = evalsim.rank_metrics(final[left].contiguous(), final[right].contiguous(), csls_k=3)
    assert res["hits_l2r"][1] > 0.5                     # training pairs rank well after training on them
    links = model.Iter_new_links(4, left.cpu().tolist(), final, right.cpu().tolist(), new_links=[])
    assert len(links) > 0 and all(len(p) == 2 for p in links)


def test_dataset_directory_to_trained_model(tmp_path):
    """SURVEY 8(f) #4 end to end: a DBP15K-layout directory (golden G11's synthetic files) -> snag_amd.data.load_data -> the HIP SNAG model ->
    optimiser steps through a DataLoader with the reference's collator -> checkpoint round trip -> eval on the test split."""
    import logging
    import os
    import pickle
    from snag_amd import data as D, evalsim
    from snag_amd.mmea import SNAG, normalize_rows
    z = load("g11_data.npz")
    root = str(tmp_path)
    file_dir = os.path.join(root, "DBP15K", "zh_en")
    os.makedirs(file_dir)
    os.makedirs(os.path.join(root, "pkls"))
    for name in z["file_names"]:
        with open(os.path.join(file_dir, str(name)), "wb") as f:
            f.write(z["file." + str(name)].tobytes())
    with open(os.path.join(root, "pkls", "zh_en_GA_id_img_feature_dict.pkl"), "wb") as f:
        pickle.dump({int(i): v for i, v in zip(z["img_ids"], z["img_vecs"])}, f)
    args = snag_args(32)
    args.__dict__.update(data_path=root, data_choice="DBP15K", data_split="zh_en", ratio="1.0", data_rate=0.3, word_embedding="glove", unsup=0,
                         batch_size=8, intermediate_size=48)
    np.random.seed(int(z["seed"]))
    kgs, non_train, train_set, test_set, _, test_ill_ = D.load_data(logging.getLogger("t"), args)
    model = SNAG(kgs, args, device=DEV)
    model.noise_mode = "philox"
    loader = torch.utils.data.DataLoader(train_set, batch_size=args.batch_size, shuffle=False, collate_fn=D.Collator_base(args))
    opt = torch.optim.AdamW(model.parameters(), lr=5e-3)
    model.train()
    losses = []
    for epoch in range(6):
        model.update_noise(seed=epoch)
        tot = 0.0
        for batch in loader:
            loss, _ = model(batch)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            tot += float(loss.detach())
        losses.append(tot)
    assert losses[-1] < losses[0], losses
    path = D.save_model(model, os.path.join(root, "SNAG", "save", "m.pkl"))
    clone = SNAG(kgs, args, device=DEV)
    D.load_model(clone, path, map_location=DEV)
    model.eval(), clone.eval()
    with torch.no_grad():
        fa, fb = normalize_rows(model.joint_emb_generat()[0]), normalize_rows(clone.joint_emb_generat()[0])
        assert torch.equal(fa, fb)
        test = np.array(test_ill_)
        res = evalsim.rank_metrics(fa[T(test[:, 0].astype(np.int64), DEV)].contiguous(), fa[T(test[:, 1].astype(np.int64), DEV)].contiguous(), csls_k=3)
    assert 0.0 <= res["hits_l2r"][0] <= 1.0 and res["mr_l2r"] >= 1.0


def test_snag_model_with_gcn_structure_encoder():
    """structure_encoder == "gcn" (model/SNAG_tools.py:98-100): the whole model builds on the GCN drop-in and trains."""
    from snag_amd.mmea import SNAG
    z = load("g10_snag_model.npz")
    N = int(z["N"])
    g = torch.Generator().manual_seed(4)
    adj = torch.sparse_coo_tensor(T(z["edges"]), torch.rand(z["edges"].shape[1], generator=g) * 0.2, (N, N))
    wo = z["ent_wo_img"].tolist()
    kgs = {"ent_num": N, "images_list": z["images_list"], "rel_features": z["rel_features"], "att_features": z["att_features"],
           "name_features": None, "char_features": None, "input_idx": torch.arange(N), "ent_wo_img": wo,
           "ent_w_img": [i for i in range(N) if i not in set(wo)], "adj": adj, "train_ill": z["train_ill"]}
    args = snag_args(int(z["D"]))
    args.structure_encoder = "gcn"
    model = SNAG(kgs, args, device=DEV)
    assert {"multimodal_encoder.cross_graph_model.gc1.weight", "multimodal_encoder.cross_graph_model.gc2.bias"} <= set(model.state_dict())
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3)
    model.train()
    model.noise_mode = "philox"
    model.update_noise(seed=1)
    losses = []
    for _ in range(10):
        loss, _ = model(z["train_ill"])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    assert model.multimodal_encoder.cross_graph_model.gc1.weight.grad.abs().sum() > 0


def test_surface_form_modalities_end_to_end(tmp_path):
    """--use_surface 1 (run_snag.sh $6): the loader's GloVe name and character-bigram features become two more modality tokens
    (6 in all: img, att, rel, gph, name, char); the HIP model trains on them."""
    import logging
    import os
    import pickle
    from snag_amd import data as D
    from snag_amd.mmea import SNAG
    z11, z13 = load("g11_data.npz"), load("g13_data_surface_unsup.npz")
    root = str(tmp_path)
    file_dir = os.path.join(root, "DBP15K", "zh_en")
    for d in (file_dir, os.path.join(root, "pkls"), os.path.join(root, "embedding"), os.path.join(root, "DBP15K", "translated_ent_name")):
        os.makedirs(d)
    for name in z11["file_names"]:
        with open(os.path.join(file_dir, str(name)), "wb") as f:
            f.write(z11["file." + str(name)].tobytes())
    with open(os.path.join(root, "pkls", "zh_en_GA_id_img_feature_dict.pkl"), "wb") as f:
        pickle.dump({int(i): v for i, v in zip(z11["img_ids"], z11["img_vecs"])}, f)
    with open(os.path.join(root, "embedding", "glove.6B.300d.txt"), "wb") as f:
        f.write(z13["glove"].tobytes())
    with open(os.path.join(root, "DBP15K", "translated_ent_name", "dbp_zh_en.json"), "wb") as f:
        f.write(z13["names_json"].tobytes())
    args = snag_args(32)
    args.__dict__.update(data_path=root, data_choice="DBP15K", data_split="zh_en", ratio="1.0", data_rate=0.3, word_embedding="glove", unsup=0,
                         batch_size=9, intermediate_size=48, w_name=True, w_char=True, inner_view_num=6)
    np.random.seed(5)
    kgs, _, train_set, _, _, _ = D.load_data(logging.getLogger("t"), args)
    assert kgs["name_features"].shape == (kgs["ent_num"], 300) and kgs["char_features"].shape[0] == kgs["ent_num"]
    model = SNAG(kgs, args, device=DEV)
    model.noise_mode = "philox"
    opt = torch.optim.AdamW(model.parameters(), lr=5e-3)
    model.train()
    model.update_noise(seed=0)
    batch = np.array(train_set.data)
    losses = []
    for _ in range(8):
        loss, out = model(batch)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    enc = model.multimodal_encoder
    assert enc.name_fc.weight.grad.abs().sum() > 0 and enc.char_fc.weight.grad.abs().sum() > 0
    model.eval()
    with torch.no_grad():
        final, _ = model.joint_emb_generat()
    assert final.shape[1] == 6 * 32


def test_set_optim_groups_and_schedule_match_reference():
    """snag_amd.optim.set_optim on the drop-in SNAG model == the reference's set_optim on its model (golden G14): the same parameter
    names in the same three groups (decayed / undecayed / 5x-lr loss weights), the same cosine schedule."""
    from snag_amd.optim import set_optim
    z10, z14 = load("g10_snag_model.npz"), load("g14_set_optim.npz")
    model = build(z10)
    opt = types.SimpleNamespace(model_name="SNAG", lr=5e-4, weight_decay=1e-4, adam_epsilon=1e-8, optim="adamw", scheduler="cos",
                                warmup_steps=40, total_steps=400, accumulation_steps=2)
    optimizer, scheduler = set_optim(opt, [model], [], [], None)
    name_of = {id(p): n for n, p in model.named_parameters()}
    assert len(optimizer.param_groups) == 3
    for gi, gr in enumerate(optimizer.param_groups):
        assert [name_of[id(p)] for p in gr["params"]] == [str(s) for s in z14[f"group{gi}.names"]], gi
        assert gr["initial_lr"] == pytest.approx(float(z14[f"group{gi}.initial_lr"])) and gr["weight_decay"] == float(z14[f"group{gi}.weight_decay"])
    for row in z14["lrs"]:
        assert [g_["lr"] for g_ in optimizer.param_groups] == pytest.approx(row.tolist(), rel=1e-12, abs=1e-18)
        optimizer.step()
        scheduler.step()


def test_mmea_training_is_reproducible_run_to_run():
    """Same seeds, same data: two runs of the drop-in SNAG model give identical parameters (alignment seeds name each entity once, so
    the InfoNCE gradient scatter never has two adds to one address; every other reduction in the path is fixed-order)."""
    z = load("g10_snag_model.npz")

    def run():
        torch.manual_seed(123)                       # seeds the Bert-dropout mask draws, as for the reference's nn.Dropout
        model = build(z)
        model.noise_mode = "philox"
        opt = torch.optim.AdamW(model.parameters(), lr=2e-3)
        model.train()
        for it in range(5):
            model.update_noise(seed=it)
            loss, _ = model(z["train_ill"])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        return {k: v.detach().clone() for k, v in model.state_dict().items()}, float(loss.detach())
    a, la = run()
    b, lb = run()
    assert la == lb
    for k in a:
        assert torch.equal(a[k], b[k]), k
