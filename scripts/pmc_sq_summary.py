"""Fold a rocprofv3 --pmc SQ pass (gpurun_out/pmc_sq) into per-kernel fractions: MFMA busy cycles per SQ busy cycle,
wait buckets per wave cycle (SQ_WAIT_ANY: parked at s_waitcnt / barrier; SQ_WAIT_INST_ANY: issue stalls;
SQ_ACTIVE_INST_ANY: issuing), LDS bank-conflict cycles per LDS-active cycle.  SQ_*_CYCLES of waves count quad-cycles and
SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md): only the ratios inside one unit are used."""
import collections, csv, json, sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"]
        key = next((k for k in ("gemm_conv_fat_kernel", "wgrad_gang_kernel", "gemm_conv_dma_kernel", "wgrad_kernel", "dw_bwd_fused_kernel", "dw_ring_kernel", "dw_s1_kernel", "dw_bwd_weight", "norm_act_fwd",
                                "norm_act_bwd_apply", "colreduce") if k in name), None)
        if key is None:
            continue
        if key == "gemm_conv_fat_kernel":     # split by tile variant: <T, BKB, MI, NJ, WM, WN, NBUF, PW1>
            for tag, label in (("Li6ELi7ELi4ELi2E", "<384x224>"), ("Li4ELi7ELi4ELi2E", "<256x224>"), ("Li2ELi7ELi8ELi1E", "<256x112>"),
                               ("Li2ELi7ELi4ELi2E", "<128x224>")):
                if tag in name:
                    key += label
                    break
        if key == "gemm_conv_dma_kernel":     # split by tile variant
            for tag in ("Li256ELi128ELb0", "Li256ELi256ELb0", "Li128ELi256ELb0", "Li128ELi128ELb0", "Lb1"):
                if tag in name:
                    key += {"Li256ELi128ELb0": "<256x128>", "Li256ELi256ELb0": "<256x256>", "Li128ELi256ELb0": "<128x256>",
                            "Li128ELi128ELb0": "<128x128>", "Lb1": "<split-K>"}[tag]
                    break
        agg[key][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVE_CYCLES":
            calls[key] += 1
out = {}
for k, c in agg.items():
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    out[k] = {"dispatches": calls[k],
              "mfma_busy_cycles_per_sq_busy_cycle_relative": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] or 1.0),
              "wait_any_per_wave_cycle": c["SQ_WAIT_ANY"] / wc, "wait_inst_per_wave_cycle": c["SQ_WAIT_INST_ANY"] / wc,
              "active_inst_per_wave_cycle": c["SQ_ACTIVE_INST_ANY"] / wc,
              "lds_conflict_per_lds_active": c["SQ_LDS_BANK_CONFLICT"] / (c["SQ_LDS_IDX_ACTIVE"] or 1.0)}
json.dump({"source": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES "
                     "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES, bench.py --steps 1 --warmup 1, 1152x768x16 batch 8 bf16",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items()):
    print(f"{k:34s} x{v['dispatches']:5d}  mfma/busy {v['mfma_busy_cycles_per_sq_busy_cycle_relative']:.3f}  wait {v['wait_any_per_wave_cycle']:.2f}  "
          f"stall {v['wait_inst_per_wave_cycle']:.2f}  active {v['active_inst_per_wave_cycle']:.2f}  lds-conflict {v['lds_conflict_per_lds_active']:.3f}")
