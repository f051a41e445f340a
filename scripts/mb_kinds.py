"""Step time of the other three forward() branches (FT-Align bs 4, caption cfg4, pretrain
cfg5) under hipGraph replay -- context for DESIGN.md, not bench lines."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from univl_amd import UniVL, BertAdam
from univl_amd.graphed import GraphedTrainStep
import bench


def run(name, rows, W, F, **over):
    tc = argparse.Namespace(max_words=W, max_frames=F, video_dim=1024, batch_size=rows, n_gpu=1, n_pair=1, margin=0.1,
                            negative_weighting=1, hard_negative_rate=0.5, use_mil=False, do_pretrain=False, task_type="retrieval",
                            stage_two=False, train_sim_after_cross=False, text_num_hidden_layers=12, visual_num_hidden_layers=6,
                            cross_num_hidden_layers=2, decoder_num_hidden_layers=3, local_rank=0, dropout_prob=0.1,
                            compute_dtype="bf16", seed=42)
    for k, v in over.items():
        setattr(tc, k, v)
    torch.manual_seed(0)
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=tc).to("cuda").train()
    opt = bench.make_optimizer(model, BertAdam)
    g = torch.Generator().manual_seed(1)
    dev = "cuda"
    ids = torch.randint(1000, 30522, (rows, 1, W), generator=g).to(dev)
    am = torch.ones(rows, 1, W, dtype=torch.int64, device=dev); tt = torch.zeros_like(am)
    video = torch.randn(rows, 1, F, 1024, generator=g, dtype=torch.float64).to(dev)
    vm = torch.ones(rows, 1, F, dtype=torch.int64, device=dev)
    labels = torch.where(torch.rand(rows, 1, W, generator=g) < 0.15, ids.cpu(), torch.full_like(ids.cpu(), -1)).to(dev)
    vlab = torch.where(torch.rand(rows, 1, F, generator=g) < 0.15, torch.zeros(rows, 1, F, dtype=torch.int64), torch.full((rows, 1, F), -1)).to(dev)
    kw = dict(pairs_masked_text=ids, pairs_token_labels=labels, masked_video=video, video_labels_index=vlab)
    if model.decoder is not None:
        cap = torch.randint(1000, 30522, (rows, 1, W), generator=g).to(dev)
        kw.update(input_caption_ids=cap, decoder_mask=torch.ones_like(cap), output_caption_ids=cap)
    gs = GraphedTrainStep(model, opt, warmup=2, persistent_inputs=True)
    for _ in range(5):
        last = float(gs(ids, tt, am, video, vm, **kw))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 15
    for _ in range(n):
        last = float(gs(ids, tt, am, video, vm, **kw))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    P = sum(p.numel() for p in model.parameters())
    print("%-34s rows %2d  %3dx%-3d  %6.2f ms/step  %7.1f rows/s  (%.1f M params, loss %.4f)" % (name, rows, W, F, ms, rows / ms * 1e3, P / 1e6, last), flush=True)
    del model, opt, gs
    torch.cuda.empty_cache()


run("FT-Joint (headline)", 4, 48, 48)
run("FT-Align (train_sim_after_cross)", 4, 48, 48, train_sim_after_cross=True)
run("caption stage two (cfg4)", 4, 128, 96, stage_two=True, task_type="caption")
run("pretrain stage two (cfg5)", 6, 48, 64, stage_two=True, do_pretrain=True, use_mil=True, n_pair=3, batch_size=6)
