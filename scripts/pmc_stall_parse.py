"""Stall breakdown per kernel family from one SQ PMC pass (scripts/r03_final.sh): sums of every collected counter per kernel name
prefix, as fractions of SQ_WAVE_CYCLES (MI355X_MICROARCH.md, rocprofv3 PMC slots: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~
WAVE_CYCLES).   python scripts/pmc_stall_parse.py <dir>"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(gemm_group_kernel|gemm_pair_kernel|gemm_ln_kernel|gemm_adam_kernel|gemm_kernel|attn_fwd_kernel|attn_bwd_kernel|ln_fwd_kernel|ln_bwd_kernel|adam_apply_kernel)", name)
    if not m:
        return "other"
    k = m.group(1)
    if k.startswith("gemm"):
        # operand layouts and tile of the instantiation: gemm_kernel<T, TA, TB, BM, BN, ...>
        t = re.search(r"Lb([01])ELb([01])ELi(\d+)ELi(\d+)", name) or re.search(r"(true|false), (true|false), (\d+), (\d+)", name) or \
            re.search(r"bool, E, (true|false), (\d+), (\d+)", name)
        if t:
            k += "<" + ",".join(t.groups()) + ">"
    return k


def main(d):
    tot = collections.defaultdict(collections.Counter)
    n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                n[k] += 1
    for k in sorted(tot, key=lambda x: -tot[x]["SQ_WAVE_CYCLES"]):
        c = tot[k]
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        print("%-48s launches %5d  wave-cycles %.3e  wait_any %.2f  wait_inst_any %.2f  active_inst %.2f  wait_inst_lds %.3f  "
              "lds_bank_conflict/lds_active %.3f  mfma_busy_cycles %.3e" %
              (k, n[k], c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc,
               c["SQ_WAIT_INST_LDS"] / wc, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), c["SQ_VALU_MFMA_BUSY_CYCLES"]))


if __name__ == "__main__":
    main(sys.argv[1])
