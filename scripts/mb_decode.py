"""Caption decoding: cached beam search (univl_amd.decode) vs the reference's per-step full recompute through
decoder_caption (main_task_caption.py:450-452), cfg4 shapes (max_words 128, max_frames 96, 2 cross + 3 decoder layers)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univl_amd import UniVL  # noqa: E402
from univl_amd.decode import CaptionBeamSearch  # noqa: E402

n_inst, nb, W, F, T = 4, 5, 128, 96, 32
tc = argparse.Namespace(max_words=W, max_frames=F, video_dim=1024, batch_size=n_inst, n_gpu=1, n_pair=1, margin=0.1,
                        negative_weighting=1, hard_negative_rate=0.5, use_mil=False, do_pretrain=False, task_type="caption",
                        stage_two=True, text_num_hidden_layers=12, visual_num_hidden_layers=6, cross_num_hidden_layers=2,
                        decoder_num_hidden_layers=3, local_rank=0, dropout_prob=0.1, compute_dtype="bf16", seed=1)
torch.manual_seed(0)
model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=tc).to("cuda").eval()
g = torch.Generator().manual_seed(1)
ids = torch.randint(1000, 30000, (n_inst, W), generator=g).cuda()
am = torch.ones(n_inst, W, dtype=torch.int64, device="cuda")
vm = torch.ones(n_inst, F, dtype=torch.int64, device="cuda")
video = torch.randn(n_inst, F, 1024, generator=g, dtype=torch.float64).cuda()
with torch.no_grad():
    so, vo = model.get_sequence_visual_output(ids, torch.zeros_like(ids), am, video, vm)
    bs = CaptionBeamSearch(model, n_inst, W, F, n_bm=nb, max_len=T)
    for _ in range(2):
        hyp, sc = bs(so, vo, am, vm, bos=101, eos=-1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hyp, sc = bs(so, vo, am, vm, bos=101, eos=-1)
    torch.cuda.synchronize()
    t_cached = time.perf_counter() - t0
    rep = lambda t: t.repeat_interleave(nb, dim=0)
    so5, vo5, am5, vm5, ids5 = rep(so), rep(vo), rep(am), rep(vm), rep(ids)
    seq = torch.full((n_inst * nb, 1), 101, dtype=torch.int64, device="cuda")
    times = []
    for t in range(T):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lg = model.decoder_caption(so5, vo5, ids5, am5, vm5, seq, torch.ones_like(seq), shaped=True, get_logits=True)
        nxt = lg[:, -1].argmax(-1, keepdim=True)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        seq = torch.cat([seq, nxt], 1)
    # first call of each prefix length builds its plan; time a second pass for the steady state
    seq2 = seq[:, :1]
    t_full = 0.0
    for t in range(T):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lg = model.decoder_caption(so5, vo5, ids5, am5, vm5, seq[:, :t + 1], torch.ones_like(seq[:, :t + 1]), shaped=True, get_logits=True)
        torch.cuda.synchronize()
        t_full += time.perf_counter() - t0
print("cached beam search : %d steps, %d instances x %d beams: %.1f ms (%.2f ms/step)" % (T, n_inst, nb, t_cached * 1e3, t_cached * 1e3 / T))
print("full recompute     : decoder_caption on the growing prefixes: %.1f ms (%.2f ms/step), x%.1f" % (t_full * 1e3, t_full * 1e3 / T, t_full / t_cached))
