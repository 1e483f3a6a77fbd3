"""CPU: HEMP host logic (aread_amd/hemp.py) against mask sequences recorded from the reference under fixed
numpy/torch seeds, plus the snapshot semantics of save/load_model_state."""
import numpy as np
import torch

from oracle import aread_oracle as O
from tests import util as U


def _model():
    spec = U.spec_full()
    model, P = U.build_model(spec, 123, device="cpu")
    model.device = torch.device("cpu")
    return spec, model, P


def test_mask_sequences_match_reference():
    spec, model, _ = _model()
    G = U.load_golden("hemp.npz")
    got = U.hemp_sequence(model, spec)
    assert set(got) == set(G)
    bad = [k for k in G if not np.array_equal(got[k], G[k])]
    assert not bad, bad[:10]


def test_validate_mask_reference_quirk_is_kept():
    """aread.py:601-604: after re-queueing the feeders of a dead last-level tower, the column cut is indexed by
    the last feeder."""
    spec, model, _ = _model()
    m = O.full_mask(spec, False)
    m[0][0, 0] = True
    m[1][0, 2] = True; m[1][0, 4] = True          # level-1 towers 2 and 4 fed by level-0 tower 0
    m[2][2, 7] = True; m[2][4, 7] = True          # level-2 tower 7 fed by level-1 towers 2 and 4 ...
    m[2][4, 4] = True                              # ... and level-2 tower 4 fed by level-1 tower 4
    out = model.validate_mask([a.copy() for a in m], add_output=False)
    # tower (2,7) has no output edge: its feeders (1,2),(1,4) are re-queued and column 4 (= last feeder) is cut
    assert not out[2][:, 4].any()
    assert out[2][2, 7] and out[2][4, 7]


def test_snapshot_omits_mmoe_bottom():
    spec, model, P = _model()
    model.save_model_state()
    keys = set(model.model_state)
    assert not any(k.startswith("mmoe_") for k in keys)
    assert "embedding.embedding_dict.weight" in keys and "towers.0.0.layers.0.weight" in keys
    assert "towers_linear.3.weight" in keys and "tower_gates.1.2.0.bias" in keys and "cn.b.1" in keys
    with torch.no_grad():
        model.dense.add_(1.0)
        model.embedding.embedding_dict.weight.add_(1.0)
    model.load_model_state()
    sd = model.state_dict()
    for k, v in P.items():
        if k.startswith(("atten", "self_attns", "V_res", "final_gate")) or k.endswith("num_batches_tracked"):
            continue
        if k.endswith(("running_mean", "running_var")):
            assert torch.equal(sd[k], v), k                    # untouched by the test's perturbation
        elif k.startswith(("mmoe_", "group_embedding")):
            assert torch.allclose(sd[k], v + 1.0), k           # fast updates leak
        else:
            assert torch.equal(sd[k], v), k                    # restored


def test_error_conventions():
    spec, model, _ = _model()
    import pytest
    with pytest.raises(ValueError):
        model.create_single_full_mask(fill_value=1.5)
    model.reset_for_mask_update()
    with pytest.raises(ValueError):
        model.generate_mask("nope")
    model.tmp_tower_gate_values = [[torch.zeros(spec.n_tower[l - 1] if l else 1) for _ in range(spec.n_tower[l])]
                                   for l in range(spec.n_level)]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()), pytest.raises(ValueError):
        model.prun_single_mask(0, model._to_tensors(O.full_mask(spec)))
