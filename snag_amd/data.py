"""Dataset readers, on-disk formats and checkpoints either side of the hot path -- SURVEY.md par. 8(f) "next" #4.

Host-side only (no kernels): the DBP15K / FB15K-style directory layout the reference trains on and the ``.pkl`` state_dict
checkpoints it writes, restated with the reference's function names, argument meaning and return values so that
``load_data(logger, args)`` output feeds ``snag_amd.mmea.SNAG(kgs, args)`` exactly as it feeds the reference model.

  <file_dir>/ent_ids_{1,2}        "<id>\\t<entity name>"                     read_raw_data / get_ids / get_ent2id  (src/data.py:406-457,472-487)
  <file_dir>/triples_{1,2}        "<h>\\t<r>\\t<t>"                          read_raw_data
  <file_dir>/ill_ent_ids          "<left id>\\t<right id>"                   read_raw_data
  <file_dir>/training_attrs_{1,2} "<entity name>\\t<attr>\\t<attr>..."       load_attr      (src/data.py:489-522)
  <data_path>/pkls/*_id_img_feature_dict*.pkl   {entity id: float vector}   load_img       (src/data.py:551-581)
  relation bag-of-words [N, 1000]                                           load_relation  (src/data.py:524-540)
  normalised adjacency as a torch sparse COO tensor                         get_adjr       (src/utils.py:327-362, normalize_adj :220-226)
  whole loader                                                              load_eva_data / load_data (src/data.py:46-76,135-272)
  checkpoints: torch.save(model.state_dict()) as <name>.pkl                 save_model / load_model   (main.py:457-500)

  surface forms (``--use_surface 1``): GloVe word vectors + character bigrams of the translated entity names
      load_word2vec / load_char_bigram / load_word_char_features (src/data.py:274-365)
  unsupervised seed links (``--unsup``): mutual-best visual pivots          visual_pivot_induction (src/data.py:367-404), get_topk_indices (src/utils.py:437-443)
"""
import json
import os
import os.path as osp
import pickle
from collections import Counter

import numpy as np
import torch


class EADataset(torch.utils.data.Dataset):
    """Index dataset over an array of aligned pairs (src/data.py:16-24)."""

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        return self.data[index]


class Collator_base(object):
    """DataLoader collate_fn: a batch of pairs -> one int array [B, 2] (src/data.py:37-44)."""

    def __init__(self, args):
        self.args = args

    def __call__(self, batch):
        return np.array(batch)


def _int_rows(path, ncols=None):
    rows = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            rows.append(tuple(int(x) for x in (parts if ncols is None else parts[:ncols])))
    return rows


def loadfile(fn, num=1):
    """First ``num`` integer columns of every line, as tuples (src/data.py:459-470)."""
    return _int_rows(fn, num)


def get_ids(fn):
    """Entity ids (first column) in file order (src/data.py:472-478)."""
    return [row[0] for row in _int_rows(fn, 1)]


def get_ent2id(fns):
    """name -> id over several ent_ids files; later files win (src/data.py:480-487)."""
    ent2id = {}
    for fn in fns:
        with open(fn, "r", encoding="utf-8") as f:
            for line in f:
                parts = line.rstrip("\n").split("\t")
                ent2id[parts[1]] = int(parts[0])
    return ent2id


def read_raw_data(file_dir, lang=(1, 2)):
    """-> (ent2id_dict, ills, triples, r_hs, r_ts, ids), as src/data.py:406-457.

    ``ids`` is one id set per language file; ``r_hs`` / ``r_ts`` map a relation to the set of its head / tail entities.
    """
    ent2id, ids = {}, []
    for i in lang:
        seen = set()
        with open(osp.join(file_dir, f"ent_ids_{i}"), "r", encoding="utf-8") as f:
            for line in f:
                parts = line.rstrip("\n").split("\t")
                ent2id[parts[1]] = int(parts[0])
                seen.add(int(parts[0]))
        ids.append(seen)
    ills = _int_rows(osp.join(file_dir, "ill_ent_ids"))
    triples = []
    for i in lang:
        triples += _int_rows(osp.join(file_dir, f"triples_{i}"))
    r_hs, r_ts = {}, {}
    for h, r, t in triples:
        r_hs.setdefault(r, set()).add(h)
        r_ts.setdefault(r, set()).add(t)
    return ent2id, ills, triples, r_hs, r_ts, ids


def load_attr(fns, e, ent2id, topA=1000):
    """Binary bag of the (at most 1000) most frequent attributes, [e, topA'] float32 (src/data.py:489-522).

    As in the reference the cap is a fixed 1000 whatever ``topA`` says, and ties in frequency keep first-seen order.
    """
    counts = Counter()
    rows = []
    for fn in fns:
        with open(fn, "r", encoding="utf-8") as f:
            for line in f:
                parts = line[:-1].split("\t")
                if parts[0] not in ent2id:
                    continue
                counts.update(parts[1:])
                rows.append(parts)
    width = min(1000, len(counts))
    attr2id = {name: i for i, name in enumerate(sorted(counts, key=counts.get, reverse=True)[:width])}
    attr = np.zeros((e, width), dtype=np.float32)
    for parts in rows:
        row = ent2id[parts[0]]
        for a in parts[1:]:
            j = attr2id.get(a)
            if j is not None:
                attr[row, j] = 1.0
    return attr


def load_relation(e, KG, topR=1000):
    """Relation bag of words: rel_mat[h, j] += 1 and rel_mat[t, j] += 1 for every triple whose relation is the j-th most
    frequent one (src/data.py:524-540).  Ties keep first-seen order (Counter.most_common)."""
    kg = np.asarray(KG, dtype=np.int64).reshape(-1, 3)
    rel_mat = np.zeros((e, topR), dtype=np.float32)
    if kg.shape[0] == 0:
        return rel_mat
    top = Counter(kg[:, 1].tolist()).most_common(topR)
    col_of = np.full(int(kg[:, 1].max()) + 1, -1, dtype=np.int64)
    for j, (r, _) in enumerate(top):
        col_of[r] = j
    cols = col_of[kg[:, 1]]
    keep = cols >= 0
    np.add.at(rel_mat, (kg[keep, 0], cols[keep]), 1.0)
    np.add.at(rel_mat, (kg[keep, 2], cols[keep]), 1.0)
    return rel_mat


def load_img(logger, e_num, path, ills):
    """Image feature table [e_num, F]: stored vectors, and for entities without an image a draw from N(mean, std) of the known
    ones through numpy's GLOBAL RNG, one call per missing entity in id order (src/data.py:551-581) -- seed numpy to reproduce.
    -> (img_embd, ent_wo_img, ent_w_img)."""
    with open(path, "rb") as f:
        img_dict = pickle.load(f)
    known = np.array(list(img_dict.values()))
    mean, std = np.mean(known, axis=0), np.std(known, axis=0)
    rows, ent_wo_img, ent_w_img = [], [], []
    for i in range(e_num):
        if i in img_dict:
            rows.append(img_dict[i])
            ent_w_img.append(i)
        else:
            rows.append(np.random.normal(mean, std, mean.shape[0]))
            ent_wo_img.append(i)
    img_embd = np.array(rows)
    if logger is not None:
        have = set(ent_w_img)
        linked = [x for pair in ills for x in pair[:2]]
        logger.info(f"{(100 * len(img_dict) / e_num):.2f}% entities have images")
        if linked:
            logger.info(f"{(100 * sum(1 for x in linked if x in have) / len(linked)):.2f}% entities in EA dataset have images")
    return img_embd, ent_wo_img, ent_w_img


def load_word2vec(path, dim=300):
    """GloVe / fastText text file -> {lower-cased word: float64 vector}; lines that do not have dim + 1 fields or do not parse are
    skipped (src/data.py:274-298)."""
    word2vec = {}
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            parts = line.strip("\n").split(" ")
            if len(parts) != dim + 1:
                continue
            try:
                word2vec[parts[0].lower()] = np.array([float(x) for x in parts[1:]], dtype=np.float64)
            except ValueError:
                continue
    return word2vec


def load_char_bigram(path):
    """translated entity names [[id, [word, ...]], ...] -> (names, {character bigram: index in first-seen order}) (src/data.py:300-316)."""
    with open(path, "r") as f:
        ent_names = json.load(f)
    char2id = {}
    for _, name in ent_names:
        for word in name:
            word = word.lower()
            for i in range(len(word) - 1):
                char2id.setdefault(word[i:i + 2], len(char2id))
    return ent_names, char2id


def load_word_char_features(node_size, word2vec_path, args, logger):
    """Name features (mean GloVe vector of the name's known words) and character-bigram counts, both L2-normalised per entity
    (src/data.py:318-365).  Entities without a known word / without a bigram get ``np.random.random(.) - 0.5`` from numpy's GLOBAL RNG,
    in entity order, name draw before char draw.  The two arrays are cached as pickles under <data_path>/embedding/ exactly as the
    reference does (and read back from there when present)."""
    name_path = osp.join(args.data_path, "DBP15K", "translated_ent_name", "dbp_" + args.data_split + ".json")
    if not osp.exists(name_path):
        raise FileNotFoundError(name_path)
    save_name = osp.join(args.data_path, "embedding", f"dbp_{args.data_split}_name.pkl")
    save_char = osp.join(args.data_path, "embedding", f"dbp_{args.data_split}_char.pkl")
    if osp.exists(save_name) and osp.exists(save_char):
        with open(save_name, "rb") as f1, open(save_char, "rb") as f2:
            return pickle.load(f1), pickle.load(f2)
    word_vecs = load_word2vec(word2vec_path)
    ent_names, char2id = load_char_bigram(name_path)
    n_gram = len(char2id)
    ent_vec = np.zeros((node_size, 300))
    char_vec = np.zeros((node_size, n_gram))

    def unit(v):
        return v / np.linalg.norm(v)

    for ent, name in ent_names:
        words = [w.lower() for w in name]
        # name row: what the row holds so far (zeros unless the id repeats) + the known words' vectors, added top to bottom, then the mean
        known = [word_vecs[w] for w in words if w in word_vecs]
        if known:
            row = np.add.reduce(np.vstack([ent_vec[ent]] + known), axis=0) / len(known)
        else:
            row = np.random.random(300) - 0.5                     # global RNG, name draw first (src/data.py:353)
        # bigram row: occurrence counts (integers: any order gives the same floats)
        grams = np.fromiter((char2id[w[j:j + 2]] for w in words for j in range(len(w) - 1)), dtype=np.int64)
        counts = char_vec[ent] + np.bincount(grams, minlength=n_gram)
        if not counts.sum():
            counts = np.random.random(n_gram) - 0.5               # ... then the char draw (:356)
        ent_vec[ent], char_vec[ent] = unit(row), unit(counts)
    os.makedirs(osp.dirname(save_name), exist_ok=True)
    with open(save_name, "wb") as f:
        pickle.dump(ent_vec, f)
    with open(save_char, "wb") as f:
        pickle.dump(char_vec, f)
    if logger is not None:
        logger.info("save entity emb done. ")
    return ent_vec, char_vec


def get_topk_indices(M, K=1000):
    """(row, col) of the K largest entries of a matrix, best first (src/utils.py:437-443)."""
    H, W = M.shape
    vals, indices = M.reshape(-1).topk(K)
    return torch.cat(((indices // W).unsqueeze(1), (indices % W).unsqueeze(1)), dim=1)


def visual_pivot_induction(args, left_ents, right_ents, img_features, ills, logger):
    """Unsupervised seed alignment (src/data.py:367-404): walk the unsup_k * 100 most similar (left, right) pairs best first and keep a
    pair when neither entity is taken yet, until unsup_k pairs are found.  The similarity product runs wherever ``img_features`` lives
    (on the GPU it is one split-bf16 MFMA GEMM (fp32 in, fp32 accumulate) through snag_amd.ops.gemm)."""
    l_img, r_img = img_features[left_ents], img_features[right_ents]
    if l_img.is_cuda:
        from . import ops
        sim = ops.gemm(l_img.contiguous().float(), r_img.contiguous().float())
    else:
        sim = l_img.mm(r_img.t())
    topk = args.unsup_k
    pairs = get_topk_indices(sim, min(topk * 100, sim.numel())).cpu().tolist()
    del sim
    used, links = set(), []
    for li, ri in pairs:
        a, b = left_ents[li], right_ents[ri]
        if a in used or b in used:
            continue
        used.add(a)
        used.add(b)
        links.append((a, b))
        if len(links) == topk:
            break
    if logger is not None and links:
        truth = set(tuple(x) for x in ills)
        logger.info(f"{(100.0 * sum(1 for l in links if l in truth) / len(links)):.2f}% in true links")
        logger.info(f"visual links length: {len(links)}")
    return np.array(links, dtype=np.int32)


def normalize_adj(mx):
    """D^-1/2 A^T D^-1/2 with D = row sums (src/utils.py:220-226; the reference's product order transposes A)."""
    import scipy.sparse as sp
    rowsum = np.array(mx.sum(1))
    with np.errstate(divide="ignore"):
        r_inv_sqrt = np.power(rowsum, -0.5).flatten()
    r_inv_sqrt[np.isinf(r_inv_sqrt)] = 0.0
    d = sp.diags(r_inv_sqrt)
    return mx.dot(d).transpose().dot(d)


def get_adjr(ent_size, triples, norm=False):
    """Symmetrised relation-count adjacency plus unit self loops as a torch sparse COO tensor (src/utils.py:327-362).

    Every ordered pair (h, t), h != t, that occurs in the triples contributes its multiplicity to BOTH (h, t) and (t, h); when the
    reverse pair also occurs both contributions are stored (duplicate coordinates, summed by coalescing / scipy).  ``norm``
    applies ``normalize_adj``.  The GAT reads only ``adj._indices()`` (model/layers.py:78).
    """
    import scipy.sparse as sp
    tri = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    tri = tri[tri[:, 0] != tri[:, 2]]
    if tri.shape[0]:
        pairs, first, mult = np.unique(tri[:, [0, 2]], axis=0, return_index=True, return_counts=True)
        order = np.argsort(first, kind="stable")        # first-seen order of the pairs, as the reference's dict
        pairs, mult = pairs[order], mult[order]
    else:
        pairs, mult = np.empty((0, 2), dtype=np.int64), np.empty(0, dtype=np.int64)
    ind = np.empty((2 * len(pairs) + ent_size, 2), dtype=np.int64)
    val = np.empty(2 * len(pairs) + ent_size, dtype=np.float32)
    ind[0:2 * len(pairs):2] = pairs
    ind[1:2 * len(pairs):2] = pairs[:, ::-1]
    val[0:2 * len(pairs):2] = mult
    val[1:2 * len(pairs):2] = mult
    ind[2 * len(pairs):, 0] = ind[2 * len(pairs):, 1] = np.arange(ent_size)
    val[2 * len(pairs):] = 1.0
    if not norm:
        return torch.sparse_coo_tensor(torch.from_numpy(ind.T.copy()), torch.from_numpy(val), (ent_size, ent_size))
    adj = sp.coo_matrix((val, (ind[:, 0].astype(np.int32), ind[:, 1].astype(np.int32))), shape=(ent_size, ent_size), dtype=np.float32)
    mx = normalize_adj(adj).tocoo().astype(np.float32)
    indices = torch.from_numpy(np.vstack((mx.row, mx.col)).astype(np.int64))
    return torch.sparse_coo_tensor(indices, torch.from_numpy(mx.data), mx.shape)


def _img_vec_path(args, file_dir):
    prefix = "" if args.ratio == "1.0" else f"_{args.ratio}"
    if "V1" in file_dir:
        args.data_split = "norm"
        return osp.join(args.data_path, f"OpenEA/pkl/{args.data_choice}_id_img_feature_dict{prefix}.pkl")
    if "V2" in file_dir:
        args.data_split = "dense"
        return osp.join(args.data_path, f"OpenEA/pkl/{args.data_choice}_id_img_feature_dict{prefix}.pkl")
    if "FB" in file_dir:
        return osp.join(args.data_path, f"pkls/{args.data_choice}_id_img_feature_dict{prefix}.pkl")
    return osp.join(args.data_path, "pkls", args.data_split + f"_GA_id_img_feature_dict{prefix}.pkl")


def load_eva_data(logger, args):
    """The reference's dataset entry point (src/data.py:135-272): same ``args`` fields (data_path, data_choice, data_split, ratio,
    data_rate, w_name, w_char, unsup), same return tuple ``(KGs, non_train, train_ill, test_ill, eval_ill, test_ill_)``.
    The ILL shuffle uses numpy's global RNG exactly once before ``load_img`` (seed numpy to reproduce a split)."""
    if "OEA" in args.data_choice:
        file_dir = osp.join(args.data_path, "OpenEA", args.data_choice)
    else:
        file_dir = osp.join(args.data_path, args.data_choice, args.data_split)
    ent2id, ills, triples, r_hs, _, _ = read_raw_data(file_dir, [1, 2])
    left_ents = get_ids(osp.join(file_dir, "ent_ids_1"))
    right_ents = get_ids(osp.join(file_dir, "ent_ids_2"))
    ent_num, rel_num = len(ent2id), len(r_hs)
    np.random.shuffle(ills)
    img_vec_path = _img_vec_path(args, file_dir)
    if not osp.exists(img_vec_path):
        raise FileNotFoundError(img_vec_path)
    img_features, ent_wo_img, ent_w_img = load_img(logger, ent_num, img_vec_path, ills)
    name_features = char_features = None
    if args.data_choice == "DBP15K" and (getattr(args, "w_name", False) or getattr(args, "w_char", False)):      # :176-193
        if getattr(args, "word_embedding", "glove") != "glove":
            raise Exception("error word embedding")
        word2vec_path = osp.join(args.data_path, "embedding", "glove.6B.300d.txt")
        ent_vec, char_vec = load_word_char_features(ent_num, word2vec_path, args, logger)
        name_features = torch.nn.functional.normalize(torch.Tensor(ent_vec))
        char_features = torch.nn.functional.normalize(torch.Tensor(char_vec))
    cut = int(len(ills) // 1 * args.data_rate)
    if getattr(args, "unsup", False):                                                                            # :195-205
        mode = getattr(args, "unsup_mode", "img")
        if mode == "char":
            input_features = char_features
        elif mode == "name":
            input_features = name_features
        else:
            input_features = torch.nn.functional.normalize(torch.Tensor(img_features))
        train_ill = visual_pivot_induction(args, left_ents, right_ents, input_features, ills, logger)
    else:
        train_ill = np.array(ills[:cut], dtype=np.int32)
    test_ill_ = ills[cut:]
    test_ill = np.array(test_ill_, dtype=np.int32)
    left_non_train = list(set(left_ents) - set(train_ill[:, 0].tolist()))
    right_non_train = list(set(right_ents) - set(train_ill[:, 1].tolist()))
    rel_features = load_relation(ent_num, triples, 1000)
    att_features = load_attr([osp.join(file_dir, "training_attrs_1"), osp.join(file_dir, "training_attrs_2")], ent_num, ent2id, 1000)
    if logger is not None:
        logger.info(f"triple num: {len(triples)}  entity num: {ent_num}  relation num: {rel_num}  "
                    f"train ill: {train_ill.shape[0]}  test ill: {test_ill.shape[0]}  without image: {len(ent_wo_img)}")
    kgs = {
        "ent_num": ent_num, "rel_num": rel_num, "images_list": img_features, "rel_features": rel_features,
        "att_features": att_features, "name_features": name_features, "char_features": char_features,
        "input_idx": torch.LongTensor(np.arange(ent_num)), "ent_wo_img": ent_wo_img, "ent_w_img": ent_w_img,
        "adj": get_adjr(ent_num, triples, norm=True), "train_ill": train_ill,
    }
    return kgs, {"left": left_non_train, "right": right_non_train}, EADataset(train_ill), EADataset(test_ill), None, test_ill_


def load_data(logger, args):
    """src/data.py:46-76."""
    return load_eva_data(logger, args)


def save_model(model, save_path):
    """``torch.save(model.state_dict(), <...>.pkl)`` (main.py:482-500).  Keys are the reference's (snag_amd.mmea / .fusion keep them)."""
    os.makedirs(osp.dirname(osp.abspath(save_path)), exist_ok=True)
    torch.save(model.state_dict(), save_path)
    return save_path


def load_model(model, save_path, map_location="cpu", strip_module_prefix=True):
    """Load a reference checkpoint into a drop-in module (main.py:457-480); ``module.`` prefixes written by a DistributedDataParallel
    run are stripped the way the reference does for its 'Dist' models.  A missing file is an error here, not a silent random init."""
    state = torch.load(save_path, map_location=map_location)
    if strip_module_prefix:
        state = {k.replace("module.", ""): v for k, v in state.items()}
    model.load_state_dict(state)
    return model
