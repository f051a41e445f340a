"""Benchmark of the UniVL hot path on MI355X: retrieval-finetune TRAINING STEP throughput (video-text pairs/s),
max_words=48 x max_frames=48, BERT-base text encoder + 6-layer visual encoder (BASELINE.json metric / configs[1]).

One step = exactly the reference's loop body (main_task_retrieval.py:333-353):
    loss = model(...); loss.backward(); float(loss); clip_grad_norm_(params, 1.0); optimizer.step(); zero_grad()
dropout 0.1 active, bf16 MFMA operands / fp32 accumulate + fp32 master weights, BertAdam state in fp32.
`value` is measured with the batch resident in HBM (the bench contract); the PCIe-inclusive rate with the batch handed
over as pageable host tensors every step (what the reference's loaders produce) is measured in the same run and printed
as `pcie_inclusive`.

    python bench.py                         one GPU
    python bench.py --gpus N                N ranks on one node: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N      (the driver's form) is used as is

One process per GPU, per-GPU batch fixed (weak scaling), gradient exchange over RCCL overlapped with the backward.
Prints ONE JSON line (rank 0) with
  roofline      the kernel family that owns the step (every gemm_kernel / gemm_pair_kernel / gemm_group_kernel launch of one step, replayed
                alone as a hipGraph and timed with HIP events on the launch stream): algorithmic bytes and flops taken from
                the launch descriptors, fractions of the 8 TB/s HBM and 2.5 PFLOP/s bf16 MFMA peaks; plus `adam` (the fused
                BertAdam update, the largest single kernel) and `step` (parameter-proportional bytes of a whole step / step time)
  cpu_baseline  the oracle's training step (CPU port of the reference's) on this box's host cores, with the port/reference
                time ratio measured in the build container (tests/golden/cpu_port_ratio.json)
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def get_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="pairs per GPU per step (configs[1]: bs=4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--kind", default="joint", choices=["joint", "align", "caption", "pretrain"],
                    help="which UniVL.forward branch to time.  joint (default, the headline): BASELINE cfg1-3, retrieval FT-Joint 48x48; "
                         "align: FT-Align (--train_sim_after_cross) 48x48; caption: cfg4, stage two, 128x96, 3 decoder layers (--batch 4); "
                         "pretrain: cfg5, stage two, five losses, 48x64, n_pair 3 (--batch = rows = videos x 3, use 6)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="BertAdam as the last kernels of its own step instead of riding with the next forward (graphed.GraphedTrainStep)")
    ap.add_argument("--pipeline", action="store_true",
                    help="force the pipelined optimizer (default already on for one bf16 process: the update rides with the next forward)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="time the MAIN loop with the batch handed over as pageable HOST tensors every step")
    ap.add_argument("--loopback", action="store_true",
                    help="single GPU: run the data-parallel schedule (exchange points, segmented hipGraphs) with identity "
                         "exchanges on a communication stream -- exercises the N>1 code path without a second GPU")
    ap.add_argument("--force-dp", action="store_true",
                    help="single rank: still initialise the RCCL process group and run the N > 1 code path (bucketed exchange, "
                         "segmented hipGraphs) with real world-size-1 collectives")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1: reduce-scatter + sharded clip / BertAdam + all-gather of the bf16 shadow instead of all-reduce")
    ap.add_argument("--grad-exchange", default=None, choices=["fp32", "bf16"],
                    help="N > 1: dtype the gradient buckets cross the wire in (default: UNIVL_GRAD_EXCHANGE or fp32, the reference's DDP "
                         "semantics; bf16 halves the bytes -- every exchanged element rounded to 8 mantissa bits, gated in tests/test_ddp_gpu.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / PCIe-inclusive side measurements")
    ap.add_argument("--no-others", action="store_true",
                    help="skip `other_configs` (the other BASELINE.json configurations, each measured by a child process of this run "
                         "after the headline: 16 / 128 pairs, FT-Align, cfg4, cfg5, the data-parallel schedule on one GPU)")
    ap.add_argument("--others-budget", type=float, default=300.0, help="wall-clock budget (s) for all `other_configs` children together")
    ap.add_argument("--measure", default="", choices=["", "eval_joint", "eval_align", "decode"],
                    help="internal (children of the default run): time a CALLER of the hot path instead of the training step -- retrieval "
                         "evaluation FT-Joint / FT-Align (univl_amd.eval.eval_retrieval, main_task_retrieval.py:383-450) or beam-5 caption "
                         "decoding (univl_amd.decode.CaptionBeamSearch, main_task_caption.py:434-522); prints one JSON line")
    ap.add_argument("--eval-items", type=int, default=0, help="--measure eval_*: number of (text, video) items (default 1024 joint / 128 align)")
    ap.add_argument("--cfg3-row", action="store_true",
                    help="also time BASELINE cfg3's share of the global batch of 128 (128 / world pairs per GPU) in THIS run (default at N > 1)")
    ap.add_argument("--watchdog-s", type=float, default=240.0,
                    help="N > 1 / --force-dp: bound (s) on every phase that can hang on a collective (process-group init, communicator "
                         "construction + captured self-test, first captured step, each timed region); on expiry every rank prints a JSON "
                         "error line and exits with status 3 instead of hanging to the driver's timeout")
    ap.add_argument("--dp-safe", action="store_true",
                    help="N > 1: gradient exchange through torch.distributed's own collectives between captured graph segments (model.dp_capture = "
                         "False) instead of the library-held RCCL communicator captured into the step graph; the supervisor's second attempt")
    ap.add_argument("--no-supervise", action="store_true",
                    help="N > 1: run the measurement in the launcher-started process itself (default: that process supervises a worker child "
                         "and retries a crashed / hung attempt in a more conservative exchange mode, see supervise())")
    ap.add_argument("--child", action="store_true", help="internal: one `other_configs` measurement (no CPU leg, no PCIe leg, no children)")
    ap.add_argument("--no-preheat", action="store_true", help="skip the declared, untimed pre-heat in front of the timed steps")
    ap.add_argument("--preheat-max-s", type=float, default=6.0)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--profile-tag", default="")
    return ap.parse_args()


# per --kind: (max_words, max_frames, task-config overrides, algorithmic fwd+bwd GFLOP per row -- SURVEY section 8d / BASELINE.md section 2,
# FlopCounterMode on the reference)
KINDS = {
    "joint": (48, 48, dict(), 37.30),
    "align": (48, 48, dict(train_sim_after_cross=True), 70.61),            # at 4 rows: every one of the 16 pairs through the cross encoder
    "caption": (128, 96, dict(stage_two=True, task_type="caption"), 155.90),
    "pretrain": (48, 64, dict(stage_two=True, do_pretrain=True, use_mil=True, n_pair=3), 185.33),
}


def task_config(args, world):
    kind = getattr(args, "kind", "joint")
    W, F, over, _ = KINDS[kind]
    tc = argparse.Namespace(
        max_words=W, max_frames=F, video_dim=1024, batch_size=args.batch * world, n_gpu=world, n_pair=1, margin=0.1,
        negative_weighting=1, hard_negative_rate=0.5, use_mil=False, do_pretrain=False, task_type="retrieval",
        stage_two=False, train_sim_after_cross=False, text_num_hidden_layers=12, visual_num_hidden_layers=6,
        cross_num_hidden_layers=2, decoder_num_hidden_layers=3, local_rank=int(os.environ.get("LOCAL_RANK", 0)),
        dropout_prob=args.dropout, compute_dtype=args.dtype, seed=42)
    for k, v in over.items():
        setattr(tc, k, v)
    return tc


def make_optimizer(model, BertAdam, lr=3e-5, coef_lr=0.1):
    """prep_optimizer of main_task_retrieval.py:168-195."""
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    named = list(model.named_parameters())
    nd = [(n, p) for n, p in named if not any(x in n for x in no_decay)]
    dc = [(n, p) for n, p in named if any(x in n for x in no_decay)]
    groups = [
        {'params': [p for n, p in nd if "bert." in n], 'weight_decay': 0.01, 'lr': lr * coef_lr},
        {'params': [p for n, p in nd if "bert." not in n], 'weight_decay': 0.01},
        {'params': [p for n, p in dc if "bert." in n], 'weight_decay': 0.0, 'lr': lr * coef_lr},
        {'params': [p for n, p in dc if "bert." not in n], 'weight_decay': 0.0},
    ]
    return BertAdam(groups, lr=lr, warmup=0.1, schedule='warmup_linear', t_total=100000, weight_decay=0.01, max_grad_norm=1.0)


def cpu_baseline(batch_rows, budget_s=20.0):
    """The CPU leg, on the host cores of this box.  Where the reference exists (UNIVL_REFERENCE_ROOT, default /root/reference: the
    build container, or a box a maintainer copied microsoft/UniVL to) the REAL reference loop is timed -- modules.modeling.UniVL +
    modules.optimization.BertAdam through oracle/time_reference.py, `kind: "reference"`.  Otherwise (the GPU boxes of this pool have
    no reference) the oracle's training step (oracle/cpu_step.py: forward, backward, clip, BertAdam -- a CPU port of the reference's
    loop body), `kind: "port"`, with the port/reference time ratio measured in the build container where both run side by side
    (tests/golden/cpu_port_ratio.json)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import _ref_harness
    kind, why_port = "port", None
    if _ref_harness.reference_available():
        try:
            import time_reference
            step = time_reference.reference_step_fn(batch_rows)
            kind = "reference"
        except Exception as ex:      # noqa: BLE001 -- a reference checkout that does not import here: say so, time the port
            why_port = "reference at %s did not load (%s: %s)" % (_ref_harness.REFERENCE_ROOT, type(ex).__name__, str(ex)[:160])
    else:
        why_port = "no reference checkout at %s on this box (UNIVL_REFERENCE_ROOT)" % _ref_harness.REFERENCE_ROOT
    if kind == "port":
        import cpu_step
        step = cpu_step.make_step(batch_rows)
    # pick the OpenMP thread count that runs the step fastest on this host (more threads than ~64 hurt at bs=4:
    # the GEMMs are [192,768]x[768,3072]); every candidate costs one step
    host = os.cpu_count()
    t_end = time.time() + budget_s
    best, best_t = None, None
    for nt in sorted({min(host, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        step()                                 # warm-up for this thread count
        t0 = time.time()
        step()
        dt_ = time.time() - t0
        if best_t is None or dt_ < best_t:
            best, best_t = nt, dt_
        if time.time() > t_end:
            break
    torch.set_num_threads(best)
    times = [best_t]
    while time.time() < t_end and len(times) < 20:
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:   # noqa: BLE001
        phys = None
    what = ("microsoft/UniVL itself (modules.modeling.UniVL + modules.optimization.BertAdam imported from %s)" % _ref_harness.REFERENCE_ROOT
            if kind == "reference" else "oracle/cpu_step.py (CPU port of the reference's loop body)")
    out = dict(value=round(batch_rows / med, 3), unit="pairs/s", cores=best, host_logical_cpus=host, host_physical_cores=phys, kind=kind,
               sample="%d full training steps (fwd+bwd+clip+BertAdam, bs=%d, 48x48, 12+6 layers, fp32, dropout 0.1) "
                      "of %s on %d OpenMP threads (best of 8/16/32/64; host has %d logical CPUs), "
                      "median step %.3f s" % (len(times), batch_rows, what, best, host, med))
    if kind == "port":
        out["why_port"] = why_port
        ratio = os.path.join(ROOT, "tests", "golden", "cpu_port_ratio.json")
        if os.path.exists(ratio):
            r = json.load(open(ratio))
            out["port_over_reference_time"] = r["port_over_reference_time"]
            out["reference_equivalent"] = round(out["value"] * r["port_over_reference_time"], 3)
            out["ratio_source"] = "tests/golden/cpu_port_ratio.json: real reference %.3f s/step vs port %.3f s/step on %d threads " \
                                  "in the build container" % (r["reference_s_per_step"], r["port_s_per_step"], r["threads"])
    return out


# The other BASELINE.json configurations (and the data-parallel SCHEDULE on one GPU), measured by children of THIS run after the
# headline, so that the driver's own clock brackets them: name, extra arguments, extra environment, the parity case that covers the
# configuration (tests/test_model_gpu.py golden fixtures of the real reference, tests/golden/<case>.npz).
OTHER_CONFIGS = [
    ("cfg3_share_of_8_gpus_16_pairs", ["--batch", "16"], {}, "joint_b16"),
    ("cfg3_share_of_4_gpus_32_pairs", ["--batch", "32"], {}, "joint_b32"),
    ("cfg3_share_of_2_gpus_64_pairs", ["--batch", "64", "--steps", "10", "--warmup", "3"], {}, "joint_b64"),
    ("cfg3_on_one_gpu_128_pairs", ["--batch", "128", "--steps", "10", "--warmup", "3"], {}, "joint_b128"),
    # the UNCHANGED training loop of main_task_retrieval.py:333-353 (model(...), loss.backward(), clip_grad_norm_, optimizer.step(),
    # optimizer.zero_grad(), float(loss)) -- no GraphedTrainStep: per-plan graph replay + the BertAdam update riding with the next forward
    ("unchanged_loop_4_pairs", ["--no-graph"], {}, "test_unchanged_training_loop_switches_to_graph_replay"),
    ("ft_align_48x48", ["--kind", "align"], {}, "align_full, align_full_cot"),
    ("cfg4_caption_128x96", ["--kind", "caption"], {}, "caption_full"),
    ("cfg5_pretrain_48x64_6_rows", ["--kind", "pretrain", "--batch", "6"], {}, "pretrain_full, pretrain_full_cot"),
    # the reference's own arithmetic (modules/modeling.py is fp32 end to end): exact-fp32 MFMA products, same plans
    ("fp32_mode_4_pairs", ["--dtype", "fp32"], {}, "joint_full[float32]"),
    # callers either side of the path, timed: retrieval evaluation and beam-search caption decoding
    ("eval_retrieval_ft_joint_1024_items", ["--measure", "eval_joint"], {}, "tests/test_eval_gpu.py"),
    ("eval_retrieval_ft_align_128_items", ["--measure", "eval_align"], {}, "tests/test_eval_gpu.py"),
    ("caption_beam5_decode_16x32", ["--measure", "decode"], {}, "tests/test_decode_gpu.py"),
    ("dp_schedule_dry_run_4_pairs", ["--force-dp"], {"UNIVL_AB": "dp_dryrun=1"}, "test_graphed_and_data_parallel_schedules_match_eager[joint_small]"),
    ("dp_schedule_dry_run_16_pairs", ["--force-dp", "--batch", "16"], {"UNIVL_AB": "dp_dryrun=1"}, "test_graphed_and_data_parallel_schedules_match_eager[joint_small]"),
]


def other_configs(args):
    t_end = time.time() + args.others_budget
    rows = []
    for name, extra, env_extra, parity in OTHER_CONFIGS:
        left = t_end - time.time()
        if left < 20:
            rows.append(dict(name=name, skipped="others budget spent"))
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--dtype", args.dtype, "--dropout", str(args.dropout)] + extra       # later arguments win (argparse)
        env = dict(os.environ)
        env.update(env_extra)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        t0 = time.time()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=min(left, 120.0))
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                rows.append(dict(name=name, error="rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])))
                continue
            j = json.loads(line[-1])
        except Exception as ex:      # noqa: BLE001
            rows.append(dict(name=name, error="%s: %s" % (type(ex).__name__, str(ex)[-300:])))
            continue
        if "--measure" in extra:     # a caller of the path, not the training step: the child's own line is the row
            j.update(name=name, args=" ".join(extra), child_wall_s=round(time.time() - t0, 1))
            rows.append(j)
            continue
        rf = j.get("roofline") or {}
        rows.append(dict(
            name=name, args=" ".join(extra), env=env_extra or None, ms_per_step=j["ms_per_step"], value=j["value"], unit=j["unit"],
            steps=j["steps"], warmup=j["warmup"], preheat_block_ms=(j.get("preheat") or {}).get("block_ms"),
            graph_mode=j["config"].get("graph_mode"), optimizer_riding=j["config"].get("optimizer_riding"), dtype=j.get("dtype"),
            roofline=dict(gemm_family_ms=rf.get("family_ms_per_step"), gemm_hbm_frac=(rf.get("hbm") or {}).get("frac"),
                          gemm_mfma_frac=(rf.get("mfma") or {}).get("frac"), step_hbm_frac=(rf.get("step") or {}).get("hbm_frac"),
                          step_mfma_frac=(rf.get("step") or {}).get("mfma_frac"), adam_frac=(rf.get("adam") or {}).get("frac"))
            if "error" not in rf else dict(error=rf["error"]),
            parity_case=parity, last_loss=j["config"].get("last_loss"), child_wall_s=round(time.time() - t0, 1)))
    return rows


def respawn_under_launcher(args):
    """`python bench.py --gpus N` without a launcher: start N ranks on this node (one per GPU) and relay rank 0's line."""
    n_vis = torch.cuda.device_count()
    if n_vis < args.gpus:
        print("[bench] --gpus %d requested but only %d GPU(s) are visible" % (args.gpus, n_vis), file=sys.stderr)
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def gemm_family(model, dev, reps=10, carried=False):
    """Replays ONLY the GEMM launches of one training step (forward + backward plan of the compiled step), in plan order on
    one stream, as a hipGraph; HIP events around `reps` replays.  Algorithmic work comes from the launch descriptors.
    carried=False: every product alone (no optimizer chunks, no folded LayerNorm: the dense contractions, comparable with rounds
    1-3); carried=True: the launches with the LayerNorm they finish in the step (round 4: univl_gemm_ln / univl_gemm_pair_ln),
    whose rows then count as algorithmic bytes of the launch."""
    from univl_amd import _lib
    st = next(v for v in model._steps.values() if getattr(v, "kind", None) in ("joint", "align", "caption", "pretrain") and v.cx.training)
    items = st.fwd.launches("univl_gemm", carried) + st.backward_plan(True).launches("univl_gemm", carried)
    flops = nbytes = wbytes = 0
    ndesc = nfold = 0
    for _, descs in items:
        for d in descs:
            if isinstance(d, _lib.LayerNorm):
                # rows the folded LayerNorm moves: forward x, residual, y, out32 (fp32) + out16; backward dout, y, dx32 (fp32) + dxd16
                nfold += 1
                nbytes += d.rows * d.N * ((12 + (4 if d.out32 else 0) + (2 if d.out16 else 0)) if not d.dout
                                          else (8 + (4 if d.dx32 else 0) + (2 if d.dxd16 else 0)))
                continue
            esz = 2 if d.dtype == _lib.DT_BF16 else 4
            ndesc += 1
            flops += 2.0 * d.M * d.N * d.K
            opb = (d.M * d.K + d.N * d.K) * esz
            outb = d.M * d.N * ((4 if d.C32 else 0) + (esz if d.C16 else 0))
            if d.flags & _lib.GEMM_ACCUM:
                outb += d.M * d.N * 4
            if d.R:
                outb += d.M * d.N * 4
            if d.aux:
                outb += d.M * d.N * esz
            nbytes += opb + outb
            wbytes += d.N * d.K * esz if not (d.trans_a and d.trans_b) else 0
    from univl_amd.engine import no_gc
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with no_gc(), torch.cuda.graph(g):
        h = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for launch, _ in items:
            rc = launch(h)
            assert rc == 0, rc
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return dict(launches=len(items), gemms=ndesc, folded_layernorms=nfold, family_ms_per_step=round(ms, 4), avg_launch_ms=round(ms / len(items), 5),
                algorithmic_bytes_per_step=int(nbytes), weight_bytes_per_step=int(wbytes), flops_per_step=flops)



def parity_leg(batch, kind, dtype):
    """The parity statistic of the BENCHED configuration, measured in this run: tests/test_model_gpu.py's golden test
    of the library's default mode for the case that covers it (fixtures of the real reference under tests/golden/) runs as a child
    process -- the test module is the checker, this file only reads the numbers it recorded."""
    case = {("joint", 4): "joint_full", ("joint", 16): "joint_b16", ("joint", 32): "joint_b32", ("joint", 64): "joint_b64",
            ("joint", 128): "joint_b128", ("caption", 4): "caption_full", ("pretrain", 6): "pretrain_full", ("align", 4): "align_full"}.get((kind, batch))
    if case is None or dtype != "bf16":
        return None
    import tempfile
    path = os.path.join(tempfile.gettempdir(), "univl_parity_%d.json" % os.getpid())
    env = dict(os.environ, UNIVL_PARITY_OUT=path)
    env.pop("UNIVL_AB", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_model_gpu.py"), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "test_forward_backward_vs_reference_golden_default_mode and %s" % case], env=env, capture_output=True, text=True,
                       timeout=240, cwd=ROOT)
    out = dict(case=case + "@default", test="tests/test_model_gpu.py::test_forward_backward_vs_reference_golden_default_mode[%s]" % case,
               passed=r.returncode == 0, seconds=round(time.time() - t0, 1), north_star_tolerance=1e-2,
               against="gradients / loss / similarity of the real reference in fp32 (tests/golden/%s.npz, oracle/make_golden.py)" % case)
    try:
        e = json.load(open(path))[case + "@default"]["bfloat16"]
        out.update(gglobal=e.get("gglobal"), gate=1.3e-2 * 1.1, gnorm=e.get("gnorm"), gmedian=e.get("gmedian"), loss=e.get("loss"), sim=e.get("sim"),
                   note="gglobal = ||all gradient samples - reference|| / ||reference||; loss / sim relative output errors (gate 1e-2); the "
                        "gradient statistic sits at the north-star tolerance, not under it: profiles/r06_emul_bf16_roundings.txt says which "
                        "operand roundings own it")
        os.remove(path)
    except Exception as ex:      # noqa: BLE001
        out["error"] = "%s: %s | %s" % (type(ex).__name__, ex, (r.stdout or r.stderr)[-200:])
    return out


class Watchdog:
    """A phase that can hang on a collective must not cost the whole scaling record: arm(label, seconds) starts a timer thread; if the
    phase is still running when it fires, this rank prints ONE JSON line naming the phase (rank 0: stdout, so the driver's record holds
    it; others: stderr) and leaves with os._exit(3) -- torch.distributed.run then tears the other ranks down."""

    def __init__(self, rank, world, enabled):
        import threading
        self._threading, self.rank, self.world, self.enabled = threading, rank, world, enabled
        self._timer, self.label = None, None

    def arm(self, label, seconds):
        self.disarm()
        if not self.enabled or seconds <= 0:
            return
        self.label = label
        self._timer = self._threading.Timer(seconds, self._fire, args=(label, seconds))
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def _fire(self, label, seconds):
        line = json.dumps(dict(metric="video-text pairs/sec (retrieval finetune, 48x48)", value=None, error="watchdog",
                               phase=label, bound_s=seconds, rank=self.rank, n_gpus=self.world,
                               what="this phase did not return within its bound; the run was aborted instead of hanging"))
        try:
            print(line, file=(sys.stdout if self.rank == 0 else sys.stderr), flush=True)
        finally:
            os._exit(3)


def _eval_model(args, dev, kind):
    from univl_amd import UniVL
    a2 = argparse.Namespace(**vars(args))
    a2.kind, a2.dropout = kind, 0.0
    tc = task_config(a2, 1)
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=tc)
    model.to(dev).eval()
    return model, tc


def measure_caller(args):
    """`--measure`: the callers either side of the hot path that rounds 1-5 only parity-tested, timed on
    synthetic data of the reference's shapes with the inputs resident in HBM.  One JSON line."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from univl_amd import _ab as _uab
    _uab.allow()
    torch.manual_seed(0)
    g = torch.Generator(device="cpu").manual_seed(99)
    out = dict(measure=args.measure, dtype=args.dtype, data="synthetic", n_gpus=1)
    if args.measure in ("eval_joint", "eval_align"):
        from univl_amd.eval import eval_retrieval
        align = args.measure == "eval_align"
        model, tc = _eval_model(args, dev, "align" if align else "joint")
        W, F = tc.max_words, tc.max_frames
        n = args.eval_items or (128 if align else 1024)
        bs = 64                                      # --batch_size_val 64 of the README's retrieval commands: 64 x 64 similarity blocks
        batches = []
        for lo in range(0, n, bs):
            b = min(bs, n - lo)
            ids = torch.randint(1000, 30522, (b, 1, W), generator=g)
            ids[..., 0] = 101
            batches.append(tuple(t.to(dev) for t in (ids, torch.ones(b, 1, W, dtype=torch.int64), torch.zeros(b, 1, W, dtype=torch.int64),
                                                     torch.randn(b, 1, F, 1024, generator=g, dtype=torch.float64),
                                                     torch.ones(b, 1, F, dtype=torch.int64))))
        eval_retrieval(model, batches)              # builds the plans
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            metrics, sim = eval_retrieval(model, batches)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        # algorithmic forward work: both encoders once per item (12.436 GFLOP: SURVEY section 8a, FlopCounterMode on the reference);
        # FT-Align adds every (text, video) pair through the 2-layer cross encoder over 96 tokens + pooler + similarity_dense
        enc = 12.436e9 * n
        cross_pair = 2 * (W + F) * 14.16e6 + 2 * 4 * (W + F) ** 2 * 768 + 2 * 768 * 768
        flops = enc + (cross_pair * n * n if align else 2.0 * n * n * 768)
        out.update(metric="retrieval evaluation, %s (items/s)" % ("FT-Align: N^2 pairs through the cross encoder" if align else "FT-Joint"),
                   value=round(n / el, 1), unit="items/s", seconds_per_eval=round(el, 4), items=n, block=bs,
                   similarity_pairs_per_s=round(n * n / el, 1),
                   roofline=dict(bound="mfma", flops=flops, achieved_tflops=round(flops / el / 1e12, 2), peak=2500.0,
                                 frac=round(flops / el / 2.5e15, 4)),
                   parity_test="tests/test_eval_gpu.py::test_eval_retrieval_and_replicas[%s]" % ("align_small" if align else "joint_small"),
                   reference="main_task_retrieval.py:383-450 (eval_epoch, _run_on_single_gpu), metrics.py:8-20",
                   R1=float(metrics["R1"]))
    else:
        from univl_amd.decode import CaptionBeamSearch
        model, tc = _eval_model(args, dev, "caption")
        W, F = tc.max_words, tc.max_frames
        n, nb, T = 16, 5, 32
        ids = torch.randint(1000, 30522, (n, 1, W), generator=g)
        ids[..., 0] = 101
        b = [t.to(dev) for t in (ids, torch.zeros(n, 1, W, dtype=torch.int64), torch.ones(n, 1, W, dtype=torch.int64),
                                 torch.randn(n, 1, F, 1024, generator=g, dtype=torch.float64), torch.ones(n, 1, F, dtype=torch.int64))]
        with torch.no_grad():
            so, vo = model.get_sequence_visual_output(*b)
        am, vm = b[2].view(n, -1), b[4].view(n, -1)
        bsr = CaptionBeamSearch(model, n, W, F, n_bm=nb, max_len=T)
        hyp, _ = bsr(so, vo, am, vm, bos=101, eos=-1)                 # captures one graph per position; eos -1: every instance runs T steps
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            hyp, _ = bsr(so, vo, am, vm, bos=101, eos=-1)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        # the reference's procedure (main_task_caption.py:441-452): decoder_caption on the COMPLETE prefixes of all n x 5 beams, once per token
        R = n * nb
        so5, vo5 = so.repeat_interleave(nb, 0), vo.repeat_interleave(nb, 0)
        am5, vm5 = am.repeat_interleave(nb, 0), vm.repeat_interleave(nb, 0)
        cap = torch.randint(1000, 30522, (R, T), generator=g).to(dev)

        def no_cache():
            with torch.no_grad():
                for t in range(1, T + 1):
                    model.decoder_caption(so5, vo5, None, am5, vm5, cap[:, :t], torch.ones(R, t, dtype=torch.int64, device=dev),
                                          shaped=True, get_logits=True)
        no_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        no_cache()
        torch.cuda.synchronize()
        el_nc = time.perf_counter() - t0
        # bytes one cached step has to read: the decoder stack's bf16 weights + the tied vocabulary table (the step is HBM / latency bound)
        esz = 2 if args.dtype == "bf16" else 4
        dec_params = sum(p.numel() for k, p in model.named_parameters() if k.startswith("decoder.decoder.layer"))
        step_bytes = (dec_params + 30522 * 768) * esz
        out.update(metric="beam-%d caption decoding (hypothesis tokens/s)" % nb, value=round(n * T / el, 1), unit="tokens/s",
                   instances=n, beams=nb, max_len=T, ms_per_position=round(el / T * 1e3, 4), seconds_per_batch=round(el, 4),
                   no_cache=dict(value=round(n * T / el_nc, 1), unit="tokens/s", seconds_per_batch=round(el_nc, 4),
                                 what="model.decoder_caption() on the complete prefixes of all %d beams once per position (the reference's loop)" % R),
                   speedup_over_no_cache=round(el_nc / el, 2),
                   roofline=dict(bound="hbm", algorithmic_bytes_per_position=step_bytes, achieved_gbs=round(step_bytes / (el / T) / 1e9, 1),
                                 peak=8000.0, frac=round(step_bytes / (el / T) / 8.0e12, 4)),
                   parity_test="tests/test_decode_gpu.py::test_beam_search_matches_reference_golden",
                   reference="main_task_caption.py:434-522, modules/beam.py", tokens_generated=sum(len(h) for h in hyp))
    print(json.dumps(out), flush=True)


# N > 1 has never run on hardware (no multi-GPU box was ever leased to the build): a segmentation fault inside a replayed graph or a hang in
# the first collective must not cost the whole scaling record.  Every launcher-started rank therefore SUPERVISES a worker child (the
# same script, UNIVL_BENCH_WORKER=1) and, if the worker dies or is stopped by its watchdog, starts the next, more conservative attempt:
#   0  default: the library-held RCCL communicator, collectives captured into the step's hipGraphs
#   1  --dp-safe: torch.distributed's collectives issued by the host between captured segments
#   2  --dp-safe --no-graph: the eager loop
# A failed attempt fails on EVERY rank (a dead rank leaves the others in a collective until their watchdog fires), so the supervisors
# need no agreement protocol: each one just moves on; attempt i rendezvouses on MASTER_PORT + 17 i.  Rank 0's supervisor relays exactly
# ONE JSON line: the first attempt's that carries a value (with `dp_attempts` saying what happened before), else the last error line.
DP_ATTEMPTS = [("captured RCCL exchange", []), ("host-issued collectives between captured segments", ["--dp-safe"]),
               ("eager loop, host-issued collectives", ["--dp-safe", "--no-graph"])]


def supervise(args):
    rank = int(os.environ.get("RANK", "0"))
    base_port = int(os.environ.get("MASTER_PORT", "29500"))
    history, last_line = [], None
    for i, (name, extra) in enumerate(DP_ATTEMPTS):
        if args.dp_safe and i == 0:
            continue                                  # the caller asked for the conservative exchange: start there
        env = dict(os.environ)
        env["UNIVL_BENCH_WORKER"] = "1"
        env["UNIVL_BENCH_ATTEMPT"] = json.dumps(dict(index=i, mode=name, earlier=history))
        env["MASTER_PORT"] = str(base_port + 17 * i)
        done = os.path.join(os.environ.get("TMPDIR", "/tmp"), "univl_bench_%d_r%d_a%d.done" % (base_port, rank, i))
        env["UNIVL_BENCH_DONE_FILE"] = done
        if os.path.exists(done):
            os.remove(done)
        cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a not in extra] + extra
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else None, text=True)
        finished = os.path.exists(done)
        if finished:
            os.remove(done)
        good = None
        if rank == 0:
            for ln in (p.stdout or "").splitlines():
                ln = ln.strip()
                if ln.startswith("{") and ln.endswith("}"):
                    try:
                        d = json.loads(ln)
                    except ValueError:
                        continue
                    last_line = ln
                    if d.get("value") is not None:
                        good = ln
        # `finished`: the worker passed its last barrier (every rank measured): a non-zero status after that (teardown) is not a retry
        if finished and (rank != 0 or good is not None):
            if rank == 0:
                print(good, flush=True)
            return 0
        history.append(dict(mode=name, returncode=p.returncode))
        print("[bench] rank %d: attempt %d (%s) ended with status %s%s" %
              (rank, i, name, p.returncode, "; next: " + DP_ATTEMPTS[i + 1][0] if i + 1 < len(DP_ATTEMPTS) else ""), file=sys.stderr, flush=True)
    if rank == 0:
        print(last_line if last_line is not None else json.dumps(dict(
            metric="video-text pairs/sec (retrieval finetune, 48x48)", value=None, error="every data-parallel attempt failed", dp_attempts=history)),
            flush=True)
    return 3


def main():
    args = get_args()
    if args.measure:
        return measure_caller(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn_under_launcher(args)
    supervised = int(os.environ.get("WORLD_SIZE", "1")) > 1 or (args.force_dp and os.environ.get("UNIVL_BENCH_SUPERVISE") == "1")
    if supervised and not os.environ.get("UNIVL_BENCH_WORKER") and not args.no_supervise:
        sys.exit(supervise(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("[bench] --gpus %d but the launcher started %d rank(s); using %d" % (args.gpus, world, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    wd = Watchdog(rank, world, enabled=(world > 1 or args.force_dp))
    if world > 1 or args.force_dp:
        attempt = json.loads(os.environ.get("UNIVL_BENCH_ATTEMPT", "{}")).get("index", 0)
        # (a later attempt: the other ranks' supervisors may arrive a whole watchdog bound later than this one)
        wd.arm("torch.distributed.init_process_group(nccl)", args.watchdog_s * (3 if attempt > 0 else 1))
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        assert torch.cuda.device_count() > local_rank, "rank %d has no GPU %d" % (rank, local_rank)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        wd.disarm()
    else:
        dist = None
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cpu_base = None

    from univl_amd import _ab as _uab, _lib as _ulib
    _uab.allow()                     # the measurement harness: UNIVL_AB overrides are honoured here (and reported in `config`)
    if world == 1 and not os.path.exists(_ulib.LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        from univl_amd import build as _ubuild          # harness convenience only; the product path never builds
        _ubuild.build(verbose=False)
    from univl_amd import UniVL, BertAdam, clip_grad_norm_
    torch.manual_seed(0)
    tc = task_config(args, world)
    model = UniVL.from_pretrained("bert-base-uncased", "visual-base", "cross-base", "decoder-base", task_config=tc)
    model.to(dev).train()
    if args.grad_exchange:
        os.environ["UNIVL_GRAD_EXCHANGE"] = args.grad_exchange
    if world > 1 or args.force_dp:
        wd.arm("enable_data_parallel: parameter broadcast, ncclCommInitRank of the library's communicator, captured self-test", args.watchdog_s)
        if args.dp_safe:
            model.dp_capture = False     # torch.distributed's collectives between captured segments (the supervisor's second attempt)
        # failure injection for the supervisor's own test (scripts/sessions/r06s_supervisor.sh): UNIVL_BENCH_INJECT="crash:0,hang:1" makes
        # attempt 0 die and attempt 1 hang in front of the first step
        for item in [x for x in os.environ.get("UNIVL_BENCH_INJECT", "").split(",") if x]:
            what, idx = item.split(":")
            if int(idx) == json.loads(os.environ.get("UNIVL_BENCH_ATTEMPT", "{}")).get("index", -1):
                if what == "crash":
                    os._exit(139)
                wd.arm("injected hang", min(args.watchdog_s, 20.0))
                time.sleep(3600)
        model.enable_data_parallel(force=args.force_dp, shard_optimizer=args.shard_optimizer)
        wd.disarm()
    elif args.loopback:
        model.enable_data_parallel(loopback=True)
    opt = make_optimizer(model, BertAdam)
    used = set(model.used_parameter_names(model.step_kind(args.kind in ("caption", "pretrain"))))      # parameters that receive a gradient
    n_params = sum(p.numel() for n, p in model.named_parameters() if n in used)

    W, F, _, gflop_row = KINDS[args.kind]
    params = list(model.parameters())
    import types

    def build_runner(B):
        """Everything that depends on the per-GPU batch: synthetic inputs resident in HBM (+ their pageable host twins), the eager loop
        body, the captured step, and the timing helpers.  The headline uses one runner; at N > 1 a second one (cfg3's share of the
        global batch of 128) is measured in the same N-rank run."""
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        ids = torch.randint(1000, 30522, (B, 1, W), generator=g)
        ids[..., 0] = 101
        host_inputs = dict(input_ids=ids, token_type_ids=torch.zeros(B, 1, W, dtype=torch.int64),
                           attention_mask=torch.ones(B, 1, W, dtype=torch.int64),
                           video=torch.randn(B, 1, F, 1024, generator=g, dtype=torch.float64),
                           video_mask=torch.ones(B, 1, F, dtype=torch.int64))
        if args.kind == "pretrain":      # masked-token / masked-frame labels (15 %), as the pretrain loader builds them
            host_inputs["pairs_token_labels"] = torch.where(torch.rand(B, 1, W, generator=g) < 0.15, ids, torch.full_like(ids, -1))
            host_inputs["video_labels_index"] = torch.where(torch.rand(B, 1, F, generator=g) < 0.15, torch.zeros(B, 1, F, dtype=torch.int64),
                                                            torch.full((B, 1, F), -1, dtype=torch.int64))
        if args.kind in ("caption", "pretrain"):
            cap = torch.randint(1000, 30522, (B, 1, W), generator=g)
            host_inputs.update(input_caption_ids=cap, decoder_mask=torch.ones_like(cap), output_caption_ids=cap.clone())
        inputs = {k: v.to(dev) for k, v in host_inputs.items()}

        def call_args(src):
            kw = dict(pairs_masked_text=src["input_ids"], pairs_token_labels=src.get("pairs_token_labels"), masked_video=src["video"],
                      video_labels_index=src.get("video_labels_index"))
            if "input_caption_ids" in src:
                kw.update(input_caption_ids=src["input_caption_ids"], decoder_mask=src["decoder_mask"], output_caption_ids=src["output_caption_ids"])
            return ((src["input_ids"], src["token_type_ids"], src["attention_mask"], src["video"], src["video_mask"]), kw)

        def step_body():
            a, kw = call_args(inputs)
            loss = model(*a, **kw)
            loss.backward()
            clip_grad_norm_(params, 1.0)
            opt.step()
            opt.zero_grad()
            return loss

        # eager warm-up (builds plans / tables), then hipGraph replay of the whole step (univl_amd.graphed): one graph on a
        # single GPU; with a gradient exchange the collectives stay on the host between captured segments
        wd.arm("first eager training steps at %d pairs per GPU (first collectives of the gradient exchange)" % B, args.watchdog_s)
        for _ in range(3):
            float(step_body())
        torch.cuda.synchronize()
        wd.arm("capture + first replay of the step graph at %d pairs per GPU" % B, args.watchdog_s)
        gstep, mode = None, ("per-plan graphs (UniVL._run_plan)" if model.auto_graph else "eager")
        if not args.no_graph:
            from univl_amd.graphed import GraphedTrainStep
            # one process, bf16: the BertAdam update of iteration t rides with the forward of iteration t + 1 (default; --no-pipeline)
            # ... and, with a captured gradient exchange, as two graphs (forward with riders | backward with collectives + clip)
            pipe = args.pipeline or (not args.no_pipeline and args.dtype == "bf16" and not args.shard_optimizer
                                     and (model._reducer is None or model._reducer.capturable))
            gstep = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=0, pipeline_optimizer=pipe,
                                     persistent_inputs=True)     # the bench contract: the batch is resident in HBM, refilled in place
            ok = 1
            try:
                a, kw = call_args(inputs)
                float(gstep(*a, **kw))
                torch.cuda.synchronize()
            except Exception as ex:      # noqa: BLE001
                print("[bench] rank %d: hipGraph capture failed (%s: %s); running eagerly" % (rank, type(ex).__name__, ex),
                      file=sys.stderr)
                ok = 0
            if dist is not None:         # every rank must take the same path
                flag = torch.tensor([ok], device=dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag)
            if ok:
                mode = "hipGraph"
            else:
                gstep = None
                model.graph_backward = False
                torch.cuda.synchronize()

        wd.disarm()

        def one_step(src):
            a, kw = call_args(src)
            if gstep is not None:
                return float(gstep(*a, **kw))     # D2H sync every step, as main_task_retrieval.py:344
            if src is not inputs:
                for k in inputs:
                    inputs[k].copy_(src[k], non_blocking=True)
            return float(step_body())

        def timed(src, steps, warmup):
            last = None
            wd.arm("timed region: %d warm-up + %d steps at %d pairs per GPU" % (warmup, steps, B), args.watchdog_s)
            for _ in range(warmup):
                last = one_step(src)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                last = one_step(src)
            if gstep is not None:
                gstep.flush()            # a riding optimizer update is still pending: it belongs to the timed steps (conservative: the region
                                         # then holds steps + 1 updates, the first one left over by the warm-up)
            else:
                opt.flush()              # the unchanged loop (--no-graph): BertAdam.step() left its update to the next forward, same accounting
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t)
            wd.disarm()
            return el, last

        def preheat(src, block, tol=0.01, need=3):
            """DECLARED, UNTIMED pre-heat in front of the timed steps.  Measured at the driver in round 3 (BENCH_r03.json): the first 25
            replays after the ~25 s CPU leg + 3 eager steps + capture ran 9 % slower than every later block of the same graph (2.72 ms vs
            2.49-2.51; the PCIe-inclusive block that followed was FASTER than the headline) -- clock / power ramp of a GPU that sat idle,
            first touch of the graph's memory pool.  So: replay blocks of `block` steps until `need` consecutive blocks agree within `tol`
            (or --preheat-max-s is spent), report every block's ms/step, and only then run the W untimed + K timed steps of the contract."""
            blocks, t_begin = [], time.perf_counter()
            stable = False
            wd.arm("pre-heat blocks at %d pairs per GPU" % B, args.watchdog_s + args.preheat_max_s)
            while True:
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(block):
                    one_step(src)
                torch.cuda.synchronize()
                blocks.append((time.perf_counter() - t0) / block * 1e3)
                last_n = blocks[-need:]
                stable = len(last_n) == need and max(last_n) <= min(last_n) * (1.0 + tol)
                spent = time.perf_counter() - t_begin
                if dist is not None:       # every rank must leave the loop in the same iteration
                    f = torch.tensor([1 if (stable or spent > args.preheat_max_s) else 0], device=dev, dtype=torch.int32)
                    dist.all_reduce(f, op=dist.ReduceOp.MAX)
                    if int(f):
                        break
                elif stable or spent > args.preheat_max_s:
                    break
            wd.disarm()
            return dict(seconds=round(time.perf_counter() - t_begin, 3), block_steps=block, block_ms=[round(b, 4) for b in blocks],
                        stable=bool(stable), rule="blocks of %d untimed steps until %d consecutive blocks agree within %.0f %% (cap %.0f s)"
                                                  % (block, need, tol * 100, args.preheat_max_s))

        return types.SimpleNamespace(B=B, inputs=inputs, host_inputs=host_inputs, call_args=call_args, step_body=step_body, gstep=gstep, mode=mode,
                                     one_step=one_step, timed=timed, preheat=preheat)

    R = build_runner(args.batch)
    inputs, host_inputs, call_args, step_body, gstep, mode, timed, preheat = (R.inputs, R.host_inputs, R.call_args, R.step_body, R.gstep, R.mode,
                                                                              R.timed, R.preheat)
    main_src = host_inputs if args.host_inputs else inputs
    pre = None if args.no_preheat else preheat(main_src, max(5, min(args.steps, 50)))
    elapsed, last = timed(main_src, args.steps, args.warmup)
    ms_per_step = elapsed / args.steps * 1e3
    pairs_per_s = args.batch * world / (elapsed / args.steps)

    pcie = None
    if not args.no_extras and not args.host_inputs and not args.child:
        k = max(10, min(args.steps, 50))
        el2, _ = timed(host_inputs, k, 3)
        pcie = dict(value=round(args.batch * world / (el2 / k), 2), ms_per_step=round(el2 / k * 1e3, 4), steps=k,
                    what="same step with the batch handed over as pageable HOST tensors every step (int64 ids/masks + float64 video, "
                         "the reference loaders' output); never the headline value")
        timed(inputs, 2, 0)          # back to the resident buffers
    if gstep is not None:
        gstep.flush()                # a pipelined optimizer step may still be pending

    def side_measurements():
        # ---- fused BertAdam update, HIP events on the launch stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a, kw = call_args(inputs)
        loss = model(*a, **kw)
        loss.backward()
        clip_grad_norm_(params, 1.0)
        opt.step()
        torch.cuda.synchronize()
        reps = 20
        ev0.record()
        for _ in range(reps):
            opt.relaunch_last()             # same descriptor: adam_prep (1 block) + adam_apply, nothing else
        ev1.record()
        torch.cuda.synchronize()
        upd_ms = ev0.elapsed_time(ev1) / reps
        bpp = 30 if args.dtype == "bf16" else 28
        adam_bytes = bpp * n_params
        adam = dict(kernel="adam_apply_kernel (fused BertAdam update, univl_amd/csrc/optim.hip)", bound="hbm",
                    achieved=round(adam_bytes / (upd_ms * 1e-3) / 1e9, 1), peak=8000.0, unit="GB/s",
                    frac=round(adam_bytes / (upd_ms * 1e-3) / 1e9 / 8000.0, 4), algorithmic_bytes_per_launch=adam_bytes,
                    avg_launch_ms=round(upd_ms, 4))
        # ---- the GEMM family of one step, replayed alone
        fam = gemm_family(model, dev)
        # the same launches as they RUN in the step where they also finish the LayerNorm behind the product (round 4, K8 / K10): slower
        # alone than the bare products (the fold's tail -- three dependent round trips -- replaces a kernel boundary and a launch that
        # this replay never contained), with the LayerNorm rows counted as algorithmic bytes; None when the plan folds nothing
        as_run = None
        try:
            fr = gemm_family(model, dev, carried=True)
            if fr["folded_layernorms"]:
                as_run = dict(folded_layernorms=fr["folded_layernorms"], family_ms_per_step=fr["family_ms_per_step"],
                              algorithmic_bytes_per_step=fr["algorithmic_bytes_per_step"],
                              hbm_frac=round(fr["algorithmic_bytes_per_step"] / (fr["family_ms_per_step"] * 1e-3) / 8.0e12, 4),
                              note="the family's launches with the LayerNorms they finish in the step; `frac` above is the bare products, as in rounds 1-3")
        except Exception as ex:   # noqa: BLE001
            as_run = dict(error=str(ex)[:200])
        fam_s = fam["family_ms_per_step"] * 1e-3
        hbm_frac = fam["algorithmic_bytes_per_step"] / fam_s / 8.0e12
        mfma_frac = fam["flops_per_step"] / fam_s / 2.5e15
        # HBM bytes per launch / per step from the PMC passes (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes of
        # scripts/pmc_step.py, both calibrated on cast_kernel's exactly known traffic) -- only while the file was produced by THESE
        # kernel sources (stamp written by scripts/pmc_step_parse.py) and for this batch; otherwise null, never a stale constant
        traffic = traffic_step = traffic_file = None
        if args.batch == 4 and args.kind == "joint":
            import glob
            import hashlib
            hs = hashlib.sha256()
            for name in ("gemm.hip", "gemm256.h", "attn_body.h", "vocab_ce.h", "common.h"):
                hs.update(open(os.path.join(ROOT, "univl_amd", "csrc", name), "rb").read())
            for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_pmc.json")), reverse=True):
                try:
                    pj = json.load(open(pmc))
                    if pj.get("kernel_source_sha16") == hs.hexdigest()[:16]:
                        traffic_step = pj["gemm"]["hbm_read_bytes_per_step"] + pj["gemm"]["hbm_write_bytes_per_step"]
                        traffic = traffic_step / fam["launches"]
                        traffic_file = os.path.relpath(pmc, ROOT)
                        break
                except Exception:   # noqa: BLE001
                    continue
        step_bytes = int((8 + bpp) * n_params + 1.0e8)
        roofline = dict(
            kernel="gemm_kernel / gemm_pair_kernel / gemm_group_kernel / gemm256 family (univl_amd/csrc/gemm.hip, gemm256.h): every dense contraction of one step",
            bound="hbm" if hbm_frac >= mfma_frac else "mfma",
            achieved=round(fam["algorithmic_bytes_per_step"] / fam_s / 1e9, 1) if hbm_frac >= mfma_frac
            else round(fam["flops_per_step"] / fam_s / 1e12, 1),
            peak=8000.0 if hbm_frac >= mfma_frac else 2500.0, unit="GB/s" if hbm_frac >= mfma_frac else "TFLOP/s",
            frac=round(max(hbm_frac, mfma_frac), 4), traffic=traffic, traffic_per_step=traffic_step, traffic_source=traffic_file,
            hbm=dict(achieved_gbs=round(fam["algorithmic_bytes_per_step"] / fam_s / 1e9, 1), frac=round(hbm_frac, 4)),
            mfma=dict(achieved_tflops=round(fam["flops_per_step"] / fam_s / 1e12, 2), frac=round(mfma_frac, 4)),
            launches_per_step=fam["launches"], gemms_per_step=fam["gemms"], family_ms_per_step=fam["family_ms_per_step"],
            avg_launch_ms=fam["avg_launch_ms"], algorithmic_bytes_per_launch=int(fam["algorithmic_bytes_per_step"] / fam["launches"]),
            algorithmic_bytes_per_step=fam["algorithmic_bytes_per_step"], flops_per_step=fam["flops_per_step"],
            how="the step's GEMM launches replayed alone (each product without the optimizer chunks / LayerNorm it carries in the step), "
                "in plan order on one stream, as a hipGraph; HIP events; includes the dependent-launch gaps between them (the "
                "rocprofv3 kernel-trace sum under profiles/ excludes them); `as_run`: the same with the folded LayerNorms",
            traffic_how="FETCH_SIZE / WRITE_SIZE of the family's launches in a step enqueued kernel by kernel with the optimizer update "
                        "as launches of its own (scripts/pmc_step.py), calibrated on cast_kernel; the products carry their folded LayerNorms",
            as_run=as_run,
            adam=adam,
            step=dict(flops_per_pair=gflop_row * 1e9, achieved_tflops=round(pairs_per_s * gflop_row * 1e9 / 1e12 / world, 2),
                      mfma_frac=round(pairs_per_s * gflop_row * 1e9 / world / 2.5e15, 4), hbm_bytes_per_step=step_bytes,
                      achieved_gbs=round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                      hbm_frac=round(step_bytes / 8.0e12 / (ms_per_step * 1e-3), 4)))
        return roofline

    roofline = None
    try:
        roofline = None if args.no_extras else side_measurements()
    except Exception as ex:      # noqa: BLE001 -- a failing side measurement must never cost the headline line
        print("[bench] rank %d: roofline side measurements failed (%s: %s)" % (rank, type(ex).__name__, ex), file=sys.stderr)
        roofline = dict(error="%s: %s" % (type(ex).__name__, ex))
    def exchange_report(R_, ms_step):
        gstep, step_body = R_.gstep, R_.step_body
        exchange = None
        if model._reducer is not None:
            stp = [v for k_, v in model._steps.items() if hasattr(v, "exchange_points") and isinstance(k_, tuple) and len(k_) > 1 and k_[1] == R_.B] \
                or [v for v in model._steps.values() if hasattr(v, "exchange_points")]
            if stp:
                pts = stp[0].exchange_points
                red = model._reducer
                exchange = dict(points=len(pts), dense_mb=round(sum(e - s for c in pts for s, e in c) * 4 / 2 ** 20, 1),
                                sparse_word_embedding=getattr(stp[0], "sparse_exchange", None),
                                backend="loopback" if red.loopback else "rccl", wire_dtype="bf16" if red.bf16 else "fp32",
                                graphs_per_iteration=(2 if (gstep is not None and gstep._g_rest is not None) else 1) if gstep is not None and gstep.mode == "whole" else None,
                                captured_in_step_graph=bool(red.capturable and gstep is not None and gstep.mode == "whole"))
                if red.capturable:
                    # Where the step's time goes once gradients cross xGMI: a few EAGER iterations (events cannot be timed inside a
                    # graph) with HIP events around every collective on the communication stream and around the join that precedes the
                    # clip.  exposed_ms = what the compute stream waits for after its last backward kernel; algbw_gbs = exchanged
                    # bytes / time the collectives themselves took (the all-reduce algorithm bandwidth the ring delivers at this size).
                    try:
                        from univl_amd.parallel import collect_timings
                        if gstep is not None:
                            gstep.flush()
                        auto, model.auto_graph, model.graph_backward = model.auto_graph, False, False
                        red.measure = True
                        for _ in range(5):
                            float(step_body())
                        tm = collect_timings(red)
                        red.measure, model.auto_graph = False, auto
                        n = max(1, tm["steps"])
                        exchange.update(exposed_ms=round(tm["exposed_ms"] / n, 4), collective_ms=round(tm["collective_ms"] / n, 4),
                                        algbw_gbs=round(tm["bytes"] / max(tm["collective_ms"], 1e-9) / 1e6, 1),
                                        measured_on="5 eager iterations after the timed region (HIP events; eager launches are host-bound, so "
                                                    "the backward these collectives hide behind is LONGER than in the graph replay: exposed_ms "
                                                    "is a lower bound for the replayed step)")
                    except Exception as ex:      # noqa: BLE001
                        exchange["timing_error"] = "%s: %s" % (type(ex).__name__, ex)
                if dist is not None and world > 1:
                    # first contact with N > 1: EVERY rank's view of the exchange in the one line rank 0 prints (a slow link or a rank
                    # that fell back shows up here, not in a log nobody collects)
                    mine = dict(rank=rank, backend=exchange["backend"], wire_dtype=exchange["wire_dtype"], captured=exchange["captured_in_step_graph"],
                                exposed_ms=exchange.get("exposed_ms"), collective_ms=exchange.get("collective_ms"), algbw_gbs=exchange.get("algbw_gbs"),
                                timing_error=exchange.get("timing_error"))
                    per_rank = [None] * world
                    dist.all_gather_object(per_rank, mine)
                    exchange["per_rank"] = per_rank
                    ex_ms = [r["exposed_ms"] for r in per_rank if r and r.get("exposed_ms") is not None]
                    if ex_ms:
                        exchange["exposed_ms_max_over_ranks"] = max(ex_ms)
                        exchange["exposed_fraction_of_step"] = round(max(ex_ms) / ms_step, 4)
        return exchange

    exchange = exchange_report(R, ms_per_step)

    # BASELINE cfg3 (MSRVTT retrieval, GLOBAL batch 128 => 128 / N pairs per GPU) in the SAME N-rank run, with its own exchange report:
    # the 4-pair headline is exchange-bound by construction (521 MB of gradients against a 2.2 ms step), cfg3 is the configuration the
    # reference trains at.  `--cfg3-row` forces it on one rank (with --force-dp: the N > 1 code path on world-size-1 RCCL).
    cfg3 = None
    if (world > 1 or args.cfg3_row) and args.kind == "joint" and 128 // world != args.batch and 128 % world == 0:
        b3 = 128 // world
        try:
            if gstep is not None:
                gstep.flush()
            R3 = build_runner(b3)
            k3, w3 = min(args.steps, 10), min(args.warmup, 3)
            pre3 = None if args.no_preheat else R3.preheat(R3.inputs, 5)
            el3, last3 = R3.timed(R3.inputs, k3, w3)
            ms3 = el3 / k3 * 1e3
            if R3.gstep is not None:
                R3.gstep.flush()
            cfg3 = dict(name="cfg3_global_batch_128_in_this_run", per_gpu_batch=b3, global_batch=b3 * world, n_gpus=world,
                        ms_per_step=round(ms3, 4), value=round(b3 * world / (el3 / k3), 2), unit="pairs/s", steps=k3, warmup=w3,
                        preheat_block_ms=(pre3 or {}).get("block_ms"), graph_mode=(R3.gstep.mode if R3.gstep is not None else R3.mode),
                        step_mfma_frac=round(b3 * gflop_row * 1e9 / (ms3 * 1e-3) / 2.5e15, 4), last_loss=round(last3, 6),
                        exchange=exchange_report(R3, ms3), parity_case=("joint_b%d" % b3) if b3 in (16, 32, 64, 128) else "joint_b16")
        except Exception as ex:      # noqa: BLE001 -- never at the price of the headline line (a rank that hangs here meets the watchdog)
            cfg3 = dict(name="cfg3_global_batch_128_in_this_run", per_gpu_batch=b3, error="%s: %s" % (type(ex).__name__, str(ex)[-300:]))
    if rank == 0:
        names = dict(joint=("video-text pairs/sec (retrieval finetune, 48x48)", "YouCookII-shape retrieval finetune (FT-Joint) training step: BERT-base text "
                            "encoder (12 L) + 6-layer visual encoder"),
                     align=("video-text pairs/sec (retrieval finetune FT-Align, 48x48)", "retrieval finetune with --train_sim_after_cross: both "
                            "encoders + all B^2 (text, video) pairs through the 2-layer cross encoder"),
                     caption=("video-caption rows/sec (caption finetune stage two, 128x96)", "BASELINE cfg4 shape: both encoders + 2-layer cross "
                              "encoder + 3-layer decoder + tied 30522-way classifier"),
                     pretrain=("rows/sec (pretrain stage two, 48x64, five losses)", "BASELINE cfg5 shape: clean + masked encoder passes, cross "
                               "encoder x3, MLM + MFM heads, decoder, alignment"))[args.kind]
        out = dict(metric=names[0], value=round(pairs_per_s, 2), unit="pairs/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic",
                   config=dict(workload="%s, max_words=%d, max_frames=%d, bs=%d per GPU, "
                                        "fwd+bwd+clip+BertAdam, dropout %.2f, random-init weights" % (names[1], W, F, args.batch, args.dropout),
                               kind=args.kind, per_gpu_batch=args.batch, global_batch=args.batch * world, max_words=W, max_frames=F,
                               parallelism="dp%d" % world, hip_graph=gstep is not None,
                               graph_mode=(gstep.mode if gstep is not None else mode),
                               optimizer_pipelined=bool(gstep is not None and gstep.pipeline),
                               optimizer_riding=bool(gstep is not None and gstep.ride),
                               host_inputs=bool(args.host_inputs), exchange=exchange, params=n_params,
                               last_loss=round(last, 6), ab_overrides=_uab.overrides()),
                   preheat=pre, pcie_inclusive=pcie, roofline=roofline, cpu_baseline=None)
    if rank == 0 and cfg3 is not None:
        out["other_configs"] = [cfg3]
    wd.disarm()
    if os.environ.get("UNIVL_BENCH_ATTEMPT"):
        out["dp_attempt"] = json.loads(os.environ["UNIVL_BENCH_ATTEMPT"])
    if dist is not None:
        wd.arm("last barrier", args.watchdog_s)
        dist.barrier()
        wd.disarm()
        if os.environ.get("UNIVL_BENCH_DONE_FILE"):          # every rank measured: the supervisor does not retry after this point
            open(os.environ["UNIVL_BENCH_DONE_FILE"], "w").close()
        if rank == 0 and os.environ.get("UNIVL_BENCH_WORKER"):
            # the line goes out BEFORE the teardown of the process group (a crash in there must not lose the measurement)
            try:
                C.CDLL(None).fflush(None)
            except Exception:   # noqa: BLE001
                pass
            print(json.dumps(out), flush=True)
        dist.destroy_process_group()
        if rank == 0 and os.environ.get("UNIVL_BENCH_WORKER"):
            return
    if rank == 0 and world == 1 and not args.child and not args.no_others and args.kind == "joint" and not args.force_dp:
        try:
            out["other_configs"] = out.get("other_configs", []) + other_configs(args)
        except Exception as ex:      # noqa: BLE001 -- never at the price of the headline line
            out["other_configs"] = out.get("other_configs", []) + [dict(error="%s: %s" % (type(ex).__name__, ex))]
    if rank == 0 and world == 1 and not args.child and not args.no_extras and not args.force_dp:
        try:
            out["parity"] = parity_leg(args.batch, args.kind, args.dtype)
        except Exception as ex:      # noqa: BLE001
            out["parity"] = dict(error="%s: %s" % (type(ex).__name__, ex))
    if rank == 0 and world == 1 and not args.child:
        # NOT measured by this run: the builder's alternating comparison of the previous round's HEAD with this one on ONE box (box-to-box
        # variance is as large as a round's gain: VERDICT r5 weak 7), carried in the line so that the record holds it
        try:
            lf = {}
            for ln in open(os.path.join(ROOT, "profiles", "r06_like_for_like_r05_vs_r06.txt")):
                tag, _, val = ln.partition(': "ms_per_step": ')
                rnd, _, cfg = tag.partition("_")
                cfg = cfg.rsplit("_", 1)[0] if cfg.rsplit("_", 1)[-1].isdigit() else cfg
                lf.setdefault(cfg, {}).setdefault(rnd, []).append(float(val))
            out["like_for_like"] = dict(
                source="profiles/r06_like_for_like_r05_vs_r06.txt (scripts/sessions/r06_like_for_like.sh; builder's session, not this run)",
                ms_per_step={c: {r: round(sum(v) / len(v), 4) for r, v in d.items()} for c, d in lf.items()},
                change={c: round(sum(d["r06"]) / len(d["r06"]) / (sum(d["r05"]) / len(d["r05"])) - 1.0, 4) for c, d in lf.items() if "r05" in d and "r06" in d})
        except Exception:   # noqa: BLE001
            pass
    if rank == 0 and world == 1 and not args.child and not args.no_cpu_baseline and args.kind == "joint":
        # after every GPU measurement (round 3 ran it first: ~25 s of idle GPU in front of the timed steps)
        try:
            out["cpu_baseline"] = cpu_baseline(args.batch)
        except Exception as ex:      # noqa: BLE001
            out["cpu_baseline"] = dict(error="%s: %s" % (type(ex).__name__, ex))
    if rank == 0:
        # RCCL writes a version banner to stdout through C stdio: flush that first, so that the JSON line is the LAST line
        try:
            C.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
