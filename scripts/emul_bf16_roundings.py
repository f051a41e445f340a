"""CPU emulation of WHERE the bf16 step's gradient error comes from ("close or formally bound the bf16 gradient gate").

The oracle (oracle/univl_oracle.py, fp32/fp64 torch) is run with straight-through rounding ops inserted at the points where the HIP path
rounds to bf16: the weight shadow, every GEMM A operand (LayerNorm output copy, attention context, gelu(u)),
q | k | v, the softmax probabilities as PV operand, the saved GELU pre-activation, and in the backward every upstream gradient that
is an MFMA operand (dY of each nn.Linear, dO, dS, d(q|k|v)).  Accumulation stays fp32/fp64 everywhere, as on the MFMA.  Each
rounding CLASS can be switched to "hi+lo" (two bf16 terms = 16 mantissa bits: what carrying the operand as a bf16 pair and issuing two
MFMAs into the same accumulators would give) or left exact, so the table says how much of the ~1e-2 global gradient error each class
owns -- before any kernel is written.

    python scripts/emul_bf16_roundings.py [case ...]          # default: joint_full; writes profiles/r06_emul_bf16_roundings.txt

Statistic: gglobal = || g - g_fp32 || / || g_fp32 || over ALL gradient elements (tests/test_model_gpu.py takes the same norm over a
strided sample), dropout off, the golden cases' seeds.  Test infrastructure: imports oracle/, never part of the product path."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import univl_oracle as O          # noqa: E402
import make_golden as MG          # noqa: E402

MODE = {}          # class -> "bf16" | "hilo" | "exact"
CLASSES = ["w", "wb", "x", "xw", "qkv", "p", "u_saved", "dy", "do", "ds", "dqkv"]     # x: A operand of the forward product; xw: the SAME activation as operand of the weight gradient


def _r(t, mode):
    if mode == "exact":
        return t
    hi = t.to(torch.bfloat16).to(t.dtype)
    if mode == "bf16":
        return hi
    return hi + (t - hi).to(torch.bfloat16).to(t.dtype)


class _FwdRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cls):
        return _r(x, MODE[cls])

    @staticmethod
    def backward(ctx, g):
        return g, None


class _BwdRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cls):
        ctx.cls = cls
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g, MODE[ctx.cls]), None


class _Gelu(torch.autograd.Function):
    """erf-GELU of the fp32 accumulator; the backward evaluates GELU' at the SAVED pre-activation (bf16 in the HIP path)."""
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u)
        return u * 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        us = _r(u, MODE["u_saved"])
        return g * (0.5 * (1.0 + torch.erf(us / math.sqrt(2.0))) + us * torch.exp(-0.5 * us * us) / math.sqrt(2.0 * math.pi))


fr, br = _FwdRound.apply, _BwdRound.apply


class _Lin(torch.autograd.Function):
    """y = r_x(x) . r_w(W)^T; backward: dx = r_dy(dy) . r_w(W), dW = r_dy(dy)^T . r_xw(x) -- the activation is rounded separately for its two
    uses (forward A operand / weight-gradient operand), the weight shadow likewise (w: forward B operand, wb: dgrad B operand)."""
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, _r(w, MODE["wb"]))
        return F.linear(_r(x, MODE["x"]), _r(w, MODE["w"]))

    @staticmethod
    def backward(ctx, g):
        x, wr = ctx.saved_tensors
        g = _r(g, MODE["dy"])
        xw = _r(x, MODE["xw"])
        return g.matmul(wr), g.reshape(-1, g.shape[-1]).t().matmul(xw.reshape(-1, xw.shape[-1]))


def linear(x, P, prefix):
    return _Lin.apply(x, P[prefix + ".weight"]) + P[prefix + ".bias"]


def attention_core(q, k, v, add_mask, nh, p_drop=0.0, training=False):
    B, Sq, H = q.shape
    Sk, d = k.shape[1], H // nh
    q, k, v = (br(fr(t, "qkv"), "dqkv") for t in (q, k, v))
    ql = q.view(B, Sq, nh, d).permute(0, 2, 1, 3)
    kl = k.view(B, Sk, nh, d).permute(0, 2, 1, 3)
    vl = v.view(B, Sk, nh, d).permute(0, 2, 1, 3)
    scores = br(torch.matmul(ql, kl.transpose(-1, -2)), "ds") / math.sqrt(d) + add_mask
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(fr(probs, "p"), vl)
    return br(ctx.permute(0, 2, 1, 3).contiguous().view(B, Sq, H), "do")


def grads(case, modes, dtype=torch.float64):
    cfg, rows, dseed = MG.case_config(case)
    cfg.dropout_prob = 0.0
    P = {k: v.to(dtype).requires_grad_(True) for k, v in O.procedural_params(cfg, 0).items()}
    batch = O.synthetic_batch(cfg, rows, seed=dseed)
    batch = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in batch.items()}
    MODE.clear()
    MODE.update({c: "exact" for c in CLASSES})
    MODE.update(modes)
    saved = O.linear, O.attention_core, O.gelu
    if modes is not None and any(v != "exact" for v in MODE.values()):
        O.linear, O.attention_core, O.gelu = linear, attention_core, _Gelu.apply
    try:
        loss = O.univl_forward(P, cfg, batch, training=False)
        loss.backward()
    finally:
        O.linear, O.attention_core, O.gelu = saved
    return float(loss), {k: v.grad.detach() for k, v in P.items() if v.grad is not None}


def gglobal(g, ref):
    num = sum(float((g[k] - ref[k]).double().pow(2).sum()) for k in ref if k in g)
    den = sum(float(ref[k].double().pow(2).sum()) for k in ref)
    return math.sqrt(num / den)


def gglobal_sampled(g, ref):
    """the statistic tests/test_model_gpu.py gates (compare_gradients): the same norm over a 256-element strided sample of EVERY tensor --
    small tensors with large elements (embedding-table sums, LayerNorm vectors, biases) own it, not the matrices"""
    num = den = 0.0
    for k in ref:
        if k in g:
            a, b = MG.sample_exact(g[k].float(), 256).astype("float64"), MG.sample_exact(ref[k].float(), 256).astype("float64")
            num += float(((a - b) ** 2).sum())
            den += float((b ** 2).sum())
    return math.sqrt(num / den)


TT = "bert.embeddings.token_type_embeddings.weight"


def variants():
    allb = {c: "bf16" for c in CLASSES}
    bwd = ["dy", "do", "ds", "dqkv"]
    out = [("all operands bf16 (the HIP path's roundings)", dict(allb))]
    out.append(("dY of every nn.Linear as hi+lo (the verdict's proposal)", dict(allb, dy="hilo")))
    out.append(("every BACKWARD operand hi+lo (dY, dO, dS, dqkv)", dict(allb, **{c: "hilo" for c in bwd})))
    out.append(("every backward operand EXACT (lower bound of any backward-side fix)", dict(allb, **{c: "exact" for c in bwd})))
    out.append(("weights hi+lo, everything else bf16", dict(allb, w="hilo", wb="hilo")))
    out.append(("weights hi+lo in the FORWARD products only", dict(allb, w="hilo")))
    out.append(("weights hi+lo in the DGRAD products only", dict(allb, wb="hilo")))
    out.append(("forward weights and dY hi+lo", dict(allb, w="hilo", dy="hilo")))
    out.append(("forward weights and forward x hi+lo", dict(allb, w="hilo", x="hilo")))
    out.append(("every FORWARD-product operand hi+lo (w, x, qkv, p)", dict(allb, w="hilo", x="hilo", qkv="hilo", p="hilo")))
    out.append(("forward activations (x, qkv, p, saved u) hi+lo, rest bf16", dict(allb, x="hilo", xw="hilo", qkv="hilo", p="hilo", u_saved="hilo")))
    out.append(("weights AND backward operands hi+lo", dict(allb, w="hilo", wb="hilo", **{c: "hilo" for c in bwd})))
    out.append(("activations AND backward operands hi+lo (weights bf16)", dict(allb, x="hilo", xw="hilo", qkv="hilo", p="hilo", u_saved="hilo", **{c: "hilo" for c in bwd})))
    out.append(("only the weights bf16", dict({c: "exact" for c in CLASSES}, w="bf16", wb="bf16")))
    out.append(("only forward activations bf16", dict({c: "exact" for c in CLASSES}, x="bf16", xw="bf16", qkv="bf16", p="bf16", u_saved="bf16")))
    out.append(("only backward operands bf16", dict({c: "exact" for c in CLASSES}, **{c: "bf16" for c in bwd})))
    out.append(("x (forward A operand) AND xw (weight-gradient operand) hi+lo", dict(allb, x="hilo", xw="hilo")))
    out.append(("x, xw and dy hi+lo", dict(allb, x="hilo", xw="hilo", dy="hilo")))
    out.append(("xw and dy hi+lo (both weight-gradient operands; off the critical path)", dict(allb, xw="hilo", dy="hilo")))
    out.append(("operand pairs as built (forward w and x hi+lo), rest bf16", dict(allb, w="hilo", x="hilo")))
    out.append(("operand pairs as built + dY hi+lo", dict(allb, w="hilo", x="hilo", dy="hilo")))
    out.append(("operand pairs as built + every backward operand and the dgrad weights hi+lo", dict(allb, w="hilo", x="hilo", wb="hilo", **{c: "hilo" for c in bwd})))
    for c in CLASSES:                      # leave-one-out: which single class is worth carrying as a pair
        out.append(("leave-one-out: %s hi+lo, rest bf16" % c, dict(allb, **{c: "hilo"})))
    if os.environ.get("EMUL_ONLY"):
        keep = os.environ["EMUL_ONLY"].split(",")
        out = [v for v in out if any(k in v[0] for k in keep)]
    return out


def main():
    cases = sys.argv[1:] or ["joint_full"]
    lines = ["emulated bf16 roundings on the CPU oracle (scripts/emul_bf16_roundings.py); gglobal = ||g - g_exact|| / ||g_exact|| over all gradient elements, dropout off",
             "north_star gate: 1.0e-2.  Measured on the MI355X for these cases (profiles/r05_final2_parity_errors.json): joint_full 0.98e-2 (deterministic), joint_b16 0.99e-2", ""]
    for case in cases:
        l0, ref = grads(case, {})
        lines.append("case %s   (exact loss %.6f)" % (case, l0))
        for name, modes in variants():
            l, g = grads(case, modes)
            tt = float((g[TT] - ref[TT]).norm() / ref[TT].norm()) if TT in ref else float("nan")
            lines.append("  %-78s gglobal %.3e   sampled (the gated statistic) %.3e   token-type table %.3e   loss err %.1e" %
                         (name, gglobal(g, ref), gglobal_sampled(g, ref), tt, abs(l - l0) / max(1e-12, abs(l0))))
            print(lines[-1], flush=True)
        lines.append("")
    out = os.path.join(ROOT, "profiles", os.environ.get("EMUL_OUT", "r06_emul_bf16_roundings.txt"))
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
