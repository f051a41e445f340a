"""A training script with the SHAPE of the reference's main_task_retrieval.py, written for tests/test_shim_gpu.py: the same
import list, import-time NCCL initialisation, `--local_rank` flag, numpy-1.x dtype aliases in the dataset, 9-tuple batches
with float64 video, parameter grouping by name, stock DistributedDataParallel wrap with find_unused_parameters=True, and
the loop body loss = model(...); loss.backward(); float(loss); torch.nn.utils.clip_grad_norm_; optimizer.step();
optimizer.zero_grad() (main_task_retrieval.py:16-23, 83, 168-198, 318-353).  It only runs through run_univl_amd.py."""
import argparse
import json
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from modules.file_utils import PYTORCH_PRETRAINED_BERT_CACHE
from modules.modeling import UniVL
from modules.optimization import BertAdam
from synthetic_data import Synthetic          # sits next to this script, like the reference's dataloaders package

torch.distributed.init_process_group(backend="nccl")


def get_args():
    p = argparse.ArgumentParser()
    p.add_argument("--output_dir", required=True)
    p.add_argument("--local_rank", default=0, type=int)
    p.add_argument("--steps", default=4, type=int)
    p.add_argument("--batch_size", default=4, type=int)
    p.add_argument("--lr", default=1e-4, type=float)
    p.add_argument("--coef_lr", default=0.1, type=float)
    a = p.parse_args()
    a.__dict__.update(max_words=20, max_frames=12, video_dim=1024, n_gpu=1, n_pair=1, margin=0.1, negative_weighting=1,
                      hard_negative_rate=0.5, use_mil=False, do_pretrain=False, task_type="retrieval", stage_two=False,
                      train_sim_after_cross=False, text_num_hidden_layers=2, visual_num_hidden_layers=1,
                      cross_num_hidden_layers=1, decoder_num_hidden_layers=1, dropout_prob=0.0, compute_dtype="fp32", seed=7)
    return a


def main():
    args = get_args()
    torch.manual_seed(args.seed)
    torch.cuda.set_device(args.local_rank)
    device = torch.device("cuda", args.local_rank)
    cache_dir = os.path.join(str(PYTORCH_PRETRAINED_BERT_CACHE), "distributed")
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", cache_dir=cache_dir,
                                  state_dict=None, task_config=args)
    model.to(device)
    named = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    nd = [(n, p) for n, p in named if not any(x in n for x in no_decay)]
    dc = [(n, p) for n, p in named if any(x in n for x in no_decay)]
    groups = [{"params": [p for n, p in nd if "bert." in n], "weight_decay": 0.01, "lr": args.lr * args.coef_lr},
              {"params": [p for n, p in nd if "bert." not in n], "weight_decay": 0.01},
              {"params": [p for n, p in dc if "bert." in n], "weight_decay": 0.0, "lr": args.lr * args.coef_lr},
              {"params": [p for n, p in dc if "bert." not in n], "weight_decay": 0.0}]
    optimizer = BertAdam(groups, lr=args.lr, warmup=0.1, schedule="warmup_linear", t_total=args.steps * 2, weight_decay=0.01,
                         max_grad_norm=1.0)
    model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[args.local_rank], output_device=args.local_rank,
                                                      find_unused_parameters=True)
    data = Synthetic(args.batch_size * args.steps, args.max_words, args.max_frames, args.video_dim, seed=args.seed)
    loader = DataLoader(data, batch_size=args.batch_size, shuffle=False, num_workers=0, pin_memory=False, drop_last=True)
    model.train()
    losses, lrs = [], []
    for batch in loader:
        batch = tuple(t.to(device=device, non_blocking=True) for t in batch)
        input_ids, input_mask, segment_ids, video, video_mask, masked_text, token_labels, masked_video, video_labels = batch
        loss = model(input_ids, segment_ids, input_mask, video, video_mask, pairs_masked_text=masked_text,
                     pairs_token_labels=token_labels, masked_video=masked_video, video_labels_index=video_labels)
        loss.backward()
        losses.append(float(loss))
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        optimizer.zero_grad()
        lrs.append(sorted(set(optimizer.get_lr())))
    os.makedirs(args.output_dir, exist_ok=True)
    torch.save(model.module.state_dict(), os.path.join(args.output_dir, "pytorch_model.bin.0"))
    with open(os.path.join(args.output_dir, "trace.json"), "w") as f:
        json.dump(dict(losses=losses, lrs=lrs, local_rank=args.local_rank, world=torch.distributed.get_world_size(),
                       backend=torch.distributed.get_backend(), model_class=type(model.module).__module__), f)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
