import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import aread_oracle as O
from tests import util as U

which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
fn, mk, seed = U.GOLDEN_MODELS[which]
G, spec = U.load_golden(fn), mk()
model, P = U.build_model(spec, seed)
model.train()
p = "single_ones"
d = int(G[f"{p}/domain"])
masks = U.golden_masks(spec, G, "ones")
x = G[f"{p}/x"]
st, gate = model._run(torch.from_numpy(x).cuda(), 0, 1, d, model._masks_dev([masks[d] if i == d else None for i in range(spec.n_domain)], "cuda"), False)
torch.cuda.synchronize()
r = O.forward(P, O.split_buffers(P), spec, x, mode="domain_mask_bagging", mask=masks[d], train=True)
cap = r["cap"]
B = x.shape[0]
def cmp(name, got, ref):
    got = got.cpu().numpy()[:B]; ref = ref.detach().numpy()
    print(f"{name:12s} max|diff| {np.abs(got-ref).max():.3e}  ref max {np.abs(ref).max():.3e}")
D = spec.d
cmp("e", st.e, cap["e"])
cmp("lin", model.debug_ws(st, "lin", 1), cap["lin"])
cmp("cn", model.debug_ws(st, "cn", D), cap["cn"])
nl = len(spec.expert_dims)
cmp("experts", model.debug_ws(st, f"ex{nl-1}.Act", spec.n_expert * spec.expert_dims[-1]), cap["experts"].flatten(1))
cmp("u", model.debug_ws(st, "In0", spec.n_tower[0] * spec.expert_dims[-1]), cap["u"].flatten(1))
for l in range(spec.n_level - 1):
    cmp(f"tower_out{l}", model.debug_ws(st, f"tw{l}.{len(spec.tower_dims[l])-1}.Act", spec.n_tower[l] * spec.tower_dims[l][-1]), cap[f"tower_out{l}"].flatten(1))
print(st.probs.cpu().numpy()[:, :4]); print(r["probs"].detach().numpy()[:, :4])
# first expert layer pre-activation
W = P["mmoe_experts.0.layers.0.weight"]; b = P["mmoe_experts.0.layers.0.bias"]
h = cap["e"] @ W.t() + b
cmp("ex0.H[g0]", model.debug_ws(st, "ex0.H", spec.n_expert * spec.expert_dims[0])[:, :spec.expert_dims[0]], h)
# ---- level-1 inputs and first layer ----
E = spec.embed_dim
dom = cap["embed"][:, spec.domain_idx, :]
act0 = np.nonzero(masks[d][0].reshape(-1))[0]
grp = P["group_embedding.weight"][torch.from_numpy(act0)].mean(dim=0, keepdim=True)
q = torch.cat([dom, grp.expand(B, -1)], dim=1)
cmp("q", model.debug_ws(st, "q", 2 * E), q)
out0 = cap["tower_out0"]
n0, n1 = spec.n_tower[0], spec.n_tower[1]
w1 = spec.tower_dims[0][-1]
ins = []
gl = []
for t in range(n1):
    lg = q @ P[f"tower_gates.0.{t}.0.weight"].t() + P[f"tower_gates.0.{t}.0.bias"]
    gl.append(lg)
    a = torch.softmax(lg, dim=1)
    col = torch.from_numpy(masks[d][1][:, t].astype(np.float32))
    am = a * col
    ah = am / (am.sum(dim=1, keepdim=True) + 1e-8)
    ins.append((ah.unsqueeze(-1) * out0).sum(dim=1))
ldgt = (sum(spec.n_tower[l] * spec.n_tower[l - 1] for l in range(1, spec.n_level)) + 3) // 4 * 4
cmp("glogT[l1]", model.debug_ws(st, "glogT", ldgt)[:, :n1 * n0], torch.cat(gl, dim=1))
In1 = torch.cat(ins, dim=1)
cmp("In1", model.debug_ws(st, "In1", n1 * w1), In1)
h1 = spec.tower_dims[1][0]
H = torch.cat([ins[t] @ P[f"towers.1.{t}.layers.0.weight"].t() + P[f"towers.1.{t}.layers.0.bias"] for t in range(n1)], dim=1)
cmp("tw1.0.H", model.debug_ws(st, "tw1.0.H", n1 * h1), H)
print(model.debug_ws(st, "tw1.0.H", n1 * h1).cpu().numpy()[:2]); print(H.detach().numpy()[:2])
