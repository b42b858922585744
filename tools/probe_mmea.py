"""DBP15K-shaped MMEA training step + eval on the HIP library (run_snag.sh shape: N = 39,594, hidden 300, batch 3,500, GAT 2x2 heads,
1-layer/1-head Mformer with FFN 400, 4 modalities).  SURVEY.md par. 6 probed the reference at this shape on CPU: 8.6 s forward, ~16 s backward."""
import sys, time, types
import numpy as np, torch
sys.path.insert(0, '.')
from snag_amd.mmea import SNAG, normalize_rows
from snag_amd import evalsim

D, N, B = 300, 39594, 3500
args = types.SimpleNamespace(
    attr_dim=D, img_dim=D, name_dim=D, char_dim=D, dropout=0.0, attn_dropout=0.0, hidden_units=f"{D},{D},{D}", heads="2,2",
    structure_encoder="gat", instance_normalization=False, inner_view_num=4, with_weight=1,
    w_gcn=True, w_img=True, w_rel=True, w_attr=True, w_name=False, w_char=False, use_project_head=False,
    hidden_size=D, num_attention_heads=1, num_hidden_layers=1, use_intermediate=1, intermediate_size=400,
    add_noise=1, noise_ratio=0.2, mask_ratio=0.7, tau=0.1, tau2=4.0, ab_weight=0.5, neg_cross_kg=0, zoom=0.1,
    reduction="mean", batch_size=B, awloss=0, semi_learn_step=5, modal_num=3)
g = torch.Generator().manual_seed(0)
T_ = 200000
a, b = torch.randint(0, N, (T_,), generator=g), torch.randint(0, N, (T_,), generator=g)
keep = a != b
edges = torch.cat([torch.stack([a[keep], b[keep]]), torch.stack([b[keep], a[keep]]), torch.arange(N).repeat(2, 1)], 1)
adj = torch.sparse_coo_tensor(edges, torch.ones(edges.shape[1]), (N, N))
perm = torch.randperm(N, generator=g)
train_ill = torch.stack([perm[:4500], perm[4500:9000]], 1).numpy().astype(np.int32)
kgs = {"ent_num": N, "images_list": torch.rand(N, 2048, generator=g).numpy(), "rel_features": (torch.rand(N, 1000, generator=g) < 0.01).float().numpy(),
       "att_features": (torch.rand(N, 1000, generator=g) < 0.01).float().numpy(), "name_features": None, "char_features": None,
       "input_idx": torch.arange(N), "ent_wo_img": list(range(0, N, 50)), "ent_w_img": [], "adj": adj, "train_ill": train_ill}
model = SNAG(kgs, args)
model.noise_mode = "philox"
opt = torch.optim.AdamW(model.parameters(), lr=5e-4)
model.train()
model.update_noise(seed=1)
batch = train_ill[:B]
def step():
    loss, _ = model(batch)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n): loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"MMEA train step (fwd+bwd+AdamW, N={N}, D={D}, B={B}): {dt * 1e3:.1f} ms/step, loss {float(loss):.3f}")
model.eval()
test_l, test_r = torch.tensor(perm[4500:15000].numpy()), torch.tensor(perm[15000:25500].numpy())
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.no_grad():
    final, _ = model.joint_emb_generat()
    final = normalize_rows(final)
    res = evalsim.rank_metrics(final[test_l.cuda()].contiguous(), final[test_r.cuda()].contiguous(), csls_k=3)
torch.cuda.synchronize()
print(f"eval (full-table encode + 10,500^2 x 1,200 similarity + CSLS + ranks): {(time.perf_counter() - t0) * 1e3:.1f} ms; hits@1 {res['hits_l2r'][0]}")
