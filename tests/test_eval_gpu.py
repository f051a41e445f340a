"""Evaluation side of the path: metrics.compute_metrics, the N x N similarity assembly and
the reference's multi-device pattern nn.parallel.replicate + one thread per replica (util.py:21-60)."""
import threading

import numpy as np
import pytest
import torch

import univl_oracle as O
from make_golden import case_config

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from univl_amd import metrics as M
    from univl_amd.eval import eval_retrieval
    from test_model_gpu import build

DEV = "cuda"


@pytest.mark.parametrize("n", [1, 7, 300, 1500])
def test_compute_metrics_matches_reference_semantics(n):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, n, generator=g)
    if n >= 7:
        x[3, 5] = x[3, 3]                       # a tie with the diagonal: the reference counts both positions
        x[2, :] = 0.25                          # a constant row: n tied positions
        x[1, 1] = x[1].max() + 1                # a rank-0 row
    want = O.compute_metrics(x.numpy())
    for arg in (x.numpy(), x.to(DEV), x.to(DEV).t().contiguous().t()):     # numpy, device, non-unit-stride device
        got = M.compute_metrics(arg)
        assert got.keys() == want.keys()
        for k in want:
            assert got[k] == want[k], (k, got[k], want[k])
    ind = M.rank_positions(x.to(DEV))
    sx = np.sort(-x.numpy(), axis=1)
    ref = np.where(sx - np.diag(-x.numpy())[:, None] == 0)[1]
    np.testing.assert_array_equal(ind, ref)


def _batches(cfg, rows, seed, nb):
    out = []
    for b in range(nb):
        bt = O.synthetic_batch(cfg, rows, seed=seed + b)
        out.append((bt["input_ids"], bt["attention_mask"], bt["token_type_ids"], bt["video"], bt["video_mask"]))
    return out


@pytest.mark.parametrize("case", ["joint_small", "align_small"])
def test_eval_retrieval_and_replicas(case):
    cfg, rows, dseed = case_config(case)
    model, P = build(cfg, torch.float32)
    model.eval()
    batches = _batches(cfg, 3, dseed, 2)
    metrics, sim = eval_retrieval(model, batches)
    assert sim.shape == (6, 6)
    # oracle: features + similarity block by block, as _run_on_single_gpu does
    feats = []
    for ids, am, tt, video, vm in batches:
        so, vo = O.get_sequence_visual_output(P, cfg, ids, tt, am, video, vm, training=False)
        feats.append((so, vo, am.reshape(-1, am.shape[-1]), vm.reshape(-1, vm.shape[-1])))
    ref = torch.cat([torch.cat([O.similarity_logits(a[0], b[1], a[2], b[3], P, cfg, False) for b in feats], dim=1)
                     for a in feats], dim=0)
    assert float((sim.cpu() - ref).abs().max()) < 1e-3
    assert metrics == O.compute_metrics(sim.cpu().numpy())            # same matrix -> same numbers as metrics.py
    want = O.compute_metrics(ref.numpy())
    assert abs(metrics["R1"] - want["R1"]) <= 1.0 / 6 + 1e-9          # identical unless a near-tie flips one rank

    # util.parallel_apply pattern: replicate, one thread per replica, each on its own share of the text batches
    replicas = torch.nn.parallel.replicate(model, [0, 0], detach=True)
    assert all(r is not model and r._flat is None for r in replicas)
    results, errors = {}, []

    def worker(i, module, share):
        try:
            with torch.cuda.device(0), torch.no_grad():
                rows_ = []
                for ids, am, tt, video, vm in share:
                    so, _ = module.get_sequence_visual_output(ids.to(DEV), tt.to(DEV), am.to(DEV), video.to(DEV), vm.to(DEV))
                    row = []
                    for ids2, am2, tt2, video2, vm2 in batches:
                        _, vo = module.get_sequence_visual_output(ids2.to(DEV), tt2.to(DEV), am2.to(DEV), video2.to(DEV), vm2.to(DEV))
                        row.append(module.get_similarity_logits(so, vo, am.to(DEV), vm2.to(DEV)).cpu().numpy())
                    rows_.append(np.concatenate(row, axis=-1))
                results[i] = np.concatenate(rows_, axis=0)
        except Exception as ex:      # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=worker, args=(i, r, [batches[i]])) for i, r in enumerate(replicas)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    par = np.concatenate([results[0], results[1]], axis=0)
    assert float(np.abs(par - sim.cpu().numpy()).max()) < 1e-5


def test_compute_metrics_matches_reference_golden(golden_dir):
    """GPU rank counts vs values produced by the reference's own metrics.compute_metrics (tests/golden/metrics.npz)."""
    import os
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    for n in (7, 60, 333):
        m = M.compute_metrics(torch.as_tensor(g["x%d" % n]).to(DEV))
        assert [m["R1"], m["R5"], m["R10"], float(m["MR"])] == list(g["m%d" % n])
