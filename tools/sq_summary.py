#!/usr/bin/env python3
"""Summarise the --pmc passes of tools/sq_counters.sh: per pipeline kernel, the median over its
full-size launches of every counter, plus a few ratios (share of wave cycles spent waiting, VALU
and LDS instructions per wave, LDS bank-conflict share)."""
import collections
import csv
import glob
import json
import sys

KEYS = {"k1_cols_fwd": "k1_cols_fwd_", "k2_rows": ("k2_rows_r16<false", "k2_rows_r16_planes"), "k2_rows_h16": "k2_rows_h16", "k2_rows_m16": "k2_rows_m16", "k3_cols_inv": "k3_cols_inv_",
        "k2_rows_group": "k2_rows_r16_group"}


def main():
    root, out = sys.argv[1], sys.argv[2]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/pass*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    res = {}
    for key, pat in KEYS.items():
        agg, agg_grid = {}, -1
        for name, counters in vals.items():
            pats = (pat,) if isinstance(pat, str) else pat
            if not any(p_ in name for p_ in pats) or name.split("(")[0].rstrip().endswith(", 1>"):   # (skip the device-side redo's K3 instantiations)
                continue
            # several kernels can match a class (the 256-row forms build the tail plan's needle spectrum): the one
            # with the largest grid is the pipeline's
            grid = max(g for lst in counters.values() for _, g in lst)
            if grid <= agg_grid:
                continue
            agg, agg_grid = {}, grid
            for cname, lst in counters.items():
                gmax = max(g for _, g in lst)
                full = sorted(v for v, g in lst if g == gmax)
                agg[cname] = full[len(full) // 2]
            agg["kernel"] = name.split("(")[0]
        if not agg:
            continue
        w = agg.get("SQ_WAVES", 0) or 1
        wc = agg.get("SQ_WAVE_CYCLES", 0) or 1
        agg["derived"] = {
            "frac_wave_cycles_waiting_any": agg.get("SQ_WAIT_ANY", 0) / wc,
            "frac_wave_cycles_waiting_for_issue": agg.get("SQ_WAIT_INST_ANY", 0) / wc,
            "frac_wave_cycles_issuing": agg.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            "valu_insts_per_wave": agg.get("SQ_INSTS_VALU", 0) / w,
            # share of the kernel's cycles in which a SIMD's VALU is busy if every VALU instruction takes four
            # cycles (1024 SIMDs; SQ_BUSY_CYCLES is summed over the chip's 32 shader engines)
            "valu_busy_at_4_cycles": agg.get("SQ_INSTS_VALU", 0) * 4.0 / (max(agg.get("SQ_BUSY_CYCLES", 0), 1) / 32.0 * 1024.0),
            "lds_insts_per_wave": agg.get("SQ_INSTS_LDS", 0) / w,
            "vmem_insts_per_wave": (agg.get("SQ_INSTS_VMEM_RD", 0) + agg.get("SQ_INSTS_VMEM_WR", 0)) / w,
            "lds_bank_conflict_share_of_lds_active": agg.get("SQ_LDS_BANK_CONFLICT", 0) / (agg.get("SQ_LDS_IDX_ACTIVE", 0) or 1),
        }
        res[key] = agg
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["derived"] for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
