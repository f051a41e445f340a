"""profiles/r02_gemm_pmc.json from the three PMC passes of scripts/pmc_step.py.
    python scripts/pmc_step_parse.py <fetch_dir> <write_dir> <mfma_dir> <flat_elements> <steps> <out.json>
HBM bytes: FETCH_SIZE / WRITE_SIZE of every gemm_kernel / gemm_group_kernel dispatch, scaled by the bytes-per-unit factors
calibrated on cast_kernel in the same pass (/opt/skills/guides/MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE tallies
wide coalesced reads at half their size; both counters are calibrated on a kernel with exactly known traffic), summed over
the measured steps and divided by the number of GEMM launches.  MFMA pass: counter sums over the same dispatches."""
import collections
import csv
import glob
import json
import sys


def rows(d):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def family(name):
    # gemm_ln_kernel / the FOLD forms of gemm_pair_kernel (round 4) also finish a LayerNorm: its rows (~2.4 MB per launch at 192 tokens)
    # are counted with the family -- the reported traffic-over-algorithmic ratio is an upper bound
    # round 5: the 256 x 256 body (gemm256_*) and the two attention launches that compute a projection inside (attn_fwd_qkv_kernel: the
    # q | k | v product; attn_bwd_odgrad_kernel: the attention-output dgrad + its riding weight gradient) belong to the family as run
    if ("gemm_group_kernel" in name or "gemm_kernel" in name or "gemm_pair_kernel" in name or "gemm_ln_kernel" in name
            or "gemm_adam_kernel" in name or "gemm_adam_rect_kernel" in name or "gemm256" in name or "attn_fwd_qkv_kernel" in name or "attn_bwd_odgrad_kernel" in name):
        return "gemm"
    if "adam_apply" in name:
        return "adam"
    if "cast_kernel" in name:
        return "cast"
    if "ln_" in name:
        return "layernorm"
    if "attn_" in name:
        return "attention"
    return "other"


def sums(d, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in rows(d):
        if r.get("Counter_Name") != counter:
            continue
        fam = family(r["Kernel_Name"])
        tot[fam] += float(r["Counter_Value"])
        n[fam] += 1
    return tot, n


def main(fetch_dir, write_dir, mfma_dir, elements, steps, out):
    f, nf = sums(fetch_dir, "FETCH_SIZE")
    w, nw = sums(write_dir, "WRITE_SIZE")
    kf = 4.0 * elements * nf["cast"] / f["cast"]          # true bytes per counter unit, reads
    kw = 2.0 * elements * nw["cast"] / w["cast"]          # writes
    res = dict(steps=steps, calibration=dict(kernel="cast_kernel (4 B read + 2 B written per element, %d elements)" % elements,
                                             read_bytes_per_unit=kf, write_bytes_per_unit=kw, launches=nf["cast"]))
    for fam in ("gemm", "adam", "layernorm", "attention", "other"):
        if nf[fam] == 0:
            continue
        rd, wr = f[fam] * kf, w[fam] * kw
        res[fam] = dict(launches_per_step=nf[fam] / steps, hbm_read_bytes_per_step=rd / steps, hbm_write_bytes_per_step=wr / steps,
                        hbm_bytes_per_launch=(rd + wr) / nf[fam])
    res["hbm_bytes_per_launch"] = res["gemm"]["hbm_bytes_per_launch"]          # what bench.py reports as roofline.traffic
    res["hbm_bytes_per_step_all_kernels"] = sum(v["hbm_read_bytes_per_step"] + v["hbm_write_bytes_per_step"]
                                                for k, v in res.items() if isinstance(v, dict) and "hbm_read_bytes_per_step" in v)
    if mfma_dir:
        m = {}
        for c in ("SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"):
            t, n = sums(mfma_dir, c)
            if n["gemm"]:
                m[c] = dict(gemm_per_step=t["gemm"] / steps, all_kernels_per_step=sum(t.values()) / steps)
        res["mfma_counters"] = m
    # stamp: the kernel sources these counters belong to -- bench.py reports roofline.traffic from this file only while they match
    import hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in ("gemm.hip", "gemm256.h", "attn_body.h", "vocab_ce.h", "common.h"):
        h.update(open(os.path.join(root, "univl_amd", "csrc", name), "rb").read())
    res["kernel_source_sha16"] = h.hexdigest()[:16]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if sys.argv[3] != "-" else None, int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
