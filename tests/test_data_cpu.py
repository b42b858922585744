"""SURVEY.md par. 8(f) #4: dataset readers / checkpoints (snag_amd.data) against golden G11 -- the outputs of the reference's own
readers (src/data.py, src/utils.py:327-362) on a tiny synthetic DBP15K-layout directory.  Host-side only: runs without a GPU."""
import logging
import os
import pickle
import types

import numpy as np
import torch

from helpers import load
from snag_amd import data as D


def _materialise(z, root):
    file_dir = os.path.join(root, "DBP15K", "zh_en")
    os.makedirs(file_dir)
    os.makedirs(os.path.join(root, "pkls"))
    for name in z["file_names"]:
        with open(os.path.join(file_dir, str(name)), "wb") as f:
            f.write(z["file." + str(name)].tobytes())
    with open(os.path.join(root, "pkls", "zh_en_GA_id_img_feature_dict.pkl"), "wb") as f:
        pickle.dump({int(i): v for i, v in zip(z["img_ids"], z["img_vecs"])}, f)
    return file_dir


def test_raw_readers_match_reference(tmp_path):
    z = load("g11_data.npz")
    file_dir = _materialise(z, str(tmp_path))
    ent2id, ills, triples, r_hs, r_ts, ids = D.read_raw_data(file_dir, [1, 2])
    assert np.array_equal(np.array(ills), z["raw_ills"])
    assert np.array_equal(np.array(triples), z["raw_triples"])
    assert len(r_hs) == len(r_ts) == int(z["n_rel"])
    assert sorted(ent2id) == [str(s) for s in z["ent_names"]]
    assert [ent2id[k] for k in sorted(ent2id)] == z["ent_ids"].tolist()
    assert D.get_ids(os.path.join(file_dir, "ent_ids_1")) == z["left_ids"].tolist()
    assert D.get_ent2id([os.path.join(file_dir, "ent_ids_1"), os.path.join(file_dir, "ent_ids_2")]) == ent2id
    assert [len(s) for s in ids] == [len(z["left_ids"]), len(ent2id) - len(z["left_ids"])]
    assert D.loadfile(os.path.join(file_dir, "triples_1"), 3) == triples[:len(D.loadfile(os.path.join(file_dir, "triples_1"), 1))]
    adj0 = D.get_adjr(len(ent2id), triples, norm=False)
    assert np.array_equal(adj0._indices().numpy(), z["adj0_indices"])
    assert np.array_equal(adj0._values().numpy(), z["adj0_values"])


def test_load_eva_data_matches_reference(tmp_path):
    z = load("g11_data.npz")
    _materialise(z, str(tmp_path))
    args = types.SimpleNamespace(data_path=str(tmp_path), data_choice="DBP15K", data_split="zh_en", ratio="1.0", data_rate=float(z["data_rate"]),
                                 word_embedding="glove", w_name=0, w_char=0, unsup=0)
    np.random.seed(int(z["seed"]))
    kgs, non_train, train_ds, test_ds, eval_ill, test_ill_ = D.load_data(logging.getLogger("t"), args)
    assert kgs["ent_num"] == int(z["ent_num"]) and kgs["rel_num"] == int(z["rel_num"])
    for k in ("images_list", "rel_features", "att_features", "train_ill"):
        assert kgs[k].dtype == z[k].dtype and np.array_equal(kgs[k], z[k]), k       # bit-exact, RNG-drawn rows included
    assert kgs["ent_wo_img"] == z["ent_wo_img"].tolist() and kgs["ent_w_img"] == z["ent_w_img"].tolist()
    assert np.array_equal(kgs["adj"]._indices().numpy(), z["adj_indices"])
    assert np.array_equal(kgs["adj"]._values().numpy(), z["adj_values"])
    assert np.array_equal(np.array(test_ill_), z["test_ill"])
    assert non_train["left"] == z["left_non_train"].tolist() and non_train["right"] == z["right_non_train"].tolist()
    assert torch.equal(kgs["input_idx"], torch.from_numpy(z["input_idx"]))
    assert kgs["name_features"] is None and kgs["char_features"] is None and eval_ill is None
    assert len(train_ds) == len(z["train_ill"]) and np.array_equal(train_ds[0], z["train_ill"][0]) and len(test_ds) == len(z["test_ill"])


def test_readers_edge_cases(tmp_path):
    assert D.load_relation(5, [], 7).shape == (5, 7)
    adj = D.get_adjr(4, [(1, 0, 1), (2, 0, 2)], norm=True)          # only self-referencing triples: identity
    assert torch.equal(adj.to_dense(), torch.eye(4))
    rel = D.load_relation(3, [(0, 9, 1), (1, 9, 2), (0, 4, 2)], 1)  # topR smaller than the relation count: relation 4 is dropped
    assert rel.tolist() == [[1.0], [2.0], [1.0]]


def test_checkpoint_round_trip_with_reference_keys(tmp_path):
    """main.py:457-500: state_dict .pkl; DistributedDataParallel 'module.' prefixes are stripped on load."""
    from snag_amd.fusion import MformerFusion
    a = types.SimpleNamespace(hidden_size=32, num_attention_heads=2, num_hidden_layers=1, use_intermediate=1, intermediate_size=48)
    m1, m2 = MformerFusion(a, modal_num=3), MformerFusion(a, modal_num=3)
    path = D.save_model(m1, os.path.join(str(tmp_path), "save", "model.pkl"))
    D.load_model(m2, path)
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert "fusion_layer.0.attention.self.query.weight" in m1.state_dict() and "weight_raw" in m1.state_dict()
    torch.save({"module." + k: v for k, v in m1.state_dict().items()}, path)
    m3 = MformerFusion(a, modal_num=3)
    D.load_model(m3, path)
    assert all(torch.equal(v, m3.state_dict()[k]) for k, v in m1.state_dict().items())


def _materialise_surface(z11, z13, root):
    _materialise(z11, root)
    os.makedirs(os.path.join(root, "embedding"))
    os.makedirs(os.path.join(root, "DBP15K", "translated_ent_name"))
    with open(os.path.join(root, "embedding", "glove.6B.300d.txt"), "wb") as f:
        f.write(z13["glove"].tobytes())
    with open(os.path.join(root, "DBP15K", "translated_ent_name", "dbp_zh_en.json"), "wb") as f:
        f.write(z13["names_json"].tobytes())


def test_surface_features_and_unsupervised_split_match_reference(tmp_path):
    """--use_surface (GloVe name vectors + character bigrams, src/data.py:274-365) and --unsup (visual pivots, :367-404), golden G13."""
    z11, z13 = load("g11_data.npz"), load("g13_data_surface_unsup.npz")
    root = str(tmp_path)
    _materialise_surface(z11, z13, root)
    args = types.SimpleNamespace(data_path=root, data_choice="DBP15K", data_split="zh_en", ratio="1.0", data_rate=0.3, word_embedding="glove",
                                 w_name=1, w_char=1, unsup=0, unsup_mode="img", unsup_k=10)
    wv = D.load_word2vec(os.path.join(root, "embedding", "glove.6B.300d.txt"))
    assert len(wv) == 40 and "badline" not in wv and "w_nan" not in wv
    np.random.seed(int(z13["seed"]))
    kgs, *_ = D.load_data(logging.getLogger("t"), args)
    assert torch.equal(kgs["name_features"], torch.from_numpy(z13["name_features"]))
    assert torch.equal(kgs["char_features"], torch.from_numpy(z13["char_features"]))
    assert np.array_equal(kgs["images_list"], z13["images_list"]) and np.array_equal(kgs["train_ill"], z13["train_ill"])
    with open(os.path.join(root, "embedding", "dbp_zh_en_name.pkl"), "rb") as f:          # the cache files the reference would read back
        assert np.array_equal(pickle.load(f), z13["cached_name"])
    with open(os.path.join(root, "embedding", "dbp_zh_en_char.pkl"), "rb") as f:
        assert np.array_equal(pickle.load(f), z13["cached_char"])
    for mode in ("img", "name", "char"):
        args.unsup, args.unsup_mode = 1, mode
        np.random.seed(int(z13["seed"]))
        kgs_u, non_u, _, _, _, test_u = D.load_data(logging.getLogger("t"), args)
        assert np.array_equal(kgs_u["train_ill"], z13[f"unsup_{mode}.train_ill"]), mode
        assert np.array_equal(np.array(test_u), z13[f"unsup_{mode}.test_ill"])
        assert non_u["left"] == z13[f"unsup_{mode}.left_non_train"].tolist()
