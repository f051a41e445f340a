"""Retrieval evaluation helpers: the device-resident equivalent of main_task_retrieval.py:367-450 (`_run_on_single_gpu`,
`eval_epoch`) for callers that can be edited.  The unchanged script keeps working through UniVL.get_* and
nn.parallel.replicate (tests/test_eval_gpu.py); these helpers avoid its per-block D2H copies and numpy concatenations:
all blocks of the N_t x N_v similarity matrix are written into one device tensor and the metrics need 2 N integers
from the GPU (univl_amd.metrics)."""
import torch

from .metrics import compute_metrics


@torch.no_grad()
def similarity_matrix(model, masks_t, masks_v, seq_outs, vis_outs):
    """masks_t[i] / seq_outs[i]: attention mask (b_i, W) and text features (b_i, W, 768) of text batch i; likewise for the
    video batches.  Returns the (sum b_i) x (sum b_j) fp32 device tensor of get_similarity_logits blocks."""
    nt = sum(int(s.shape[0]) for s in seq_outs)
    nv = sum(int(v.shape[0]) for v in vis_outs)
    out = torch.empty(nt, nv, device=seq_outs[0].device, dtype=torch.float32)
    r = 0
    for mt, so in zip(masks_t, seq_outs):
        c = 0
        for mv, vo in zip(masks_v, vis_outs):
            blk = model.get_similarity_logits(so, vo, mt, mv)
            out[r:r + so.shape[0], c:c + vo.shape[0]] = blk
            c += vo.shape[0]
        r += so.shape[0]
    return out


@torch.no_grad()
def eval_retrieval(model, batches, device="cuda"):
    """batches: iterable of (input_ids, input_mask, segment_ids, video, video_mask, ...) tuples in the reference
    loader's order (main_task_retrieval.py:396).  Returns (metrics dict, similarity matrix on the device)."""
    was_training = model.training
    model.eval()
    masks_t, masks_v, seqs, viss = [], [], [], []
    for batch in batches:
        input_ids, input_mask, segment_ids, video, video_mask = [t.to(device) for t in batch[:5]]
        so, vo = model.get_sequence_visual_output(input_ids, segment_ids, input_mask, video, video_mask)
        seqs.append(so)
        viss.append(vo)
        masks_t.append(input_mask.reshape(-1, input_mask.shape[-1]))
        masks_v.append(video_mask.reshape(-1, video_mask.shape[-1]))
    sim = similarity_matrix(model, masks_t, masks_v, seqs, viss)
    model.train(was_training)
    return compute_metrics(sim), sim
