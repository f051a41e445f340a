"""Drop-in for the reference's `metrics.py` (compute_metrics :8-20, print_computed_metrics :22-27).

`compute_metrics(x)` accepts what the reference passes (the numpy N x N similarity matrix assembled by
main_task_retrieval.py:_run_on_single_gpu) or a device tensor, and returns the same dict with the same values.  The
N x N row sort of the reference is replaced by two counts per row computed on the GPU (univl_rank_counts): the
positions `ind` at which the diagonal score sits in the descending-sorted row are range(#greater, #greater + #equal).
There is no CPU fallback."""
import numpy as np
import torch

from . import ops


def rank_positions(x):
    """The reference's `ind` array (metrics.py:9-14), ties included, as a sorted-by-row int64 numpy array."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.ascontiguousarray(x))
    if not x.is_cuda:
        x = x.to("cuda")
    x = x.to(torch.float32).contiguous()
    gt, eq = ops.rank_counts(x)
    gt, eq = gt.cpu().numpy().astype(np.int64), eq.cpu().numpy().astype(np.int64)
    if int(eq.max()) == 1:
        return gt
    return np.concatenate([np.arange(g, g + e) for g, e in zip(gt, eq)])


def compute_metrics(x):
    ind = rank_positions(x)
    metrics = {}
    metrics['R1'] = float(np.sum(ind == 0)) / len(ind)
    metrics['R5'] = float(np.sum(ind < 5)) / len(ind)
    metrics['R10'] = float(np.sum(ind < 10)) / len(ind)
    metrics['MR'] = np.median(ind) + 1
    return metrics


def print_computed_metrics(metrics):
    r1 = metrics['R1']
    r5 = metrics['R5']
    r10 = metrics['R10']
    mr = metrics['MR']
    print('R@1: {:.4f} - R@5: {:.4f} - R@10: {:.4f} - Median R: {}'.format(r1, r5, r10, mr))
