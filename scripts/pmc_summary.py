"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gpurun_out/pmc_{fetch,write}) into
per-kernel HBM traffic per launch.  Units and the gfx950 correction follow
MI355X_MICROARCH.md (HBM): counters are in KiB; FETCH_SIZE reports exactly half of the
bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, json, sys

FAMILY = {  # kernel symbol fragment -> C-ABI entry points it implements
    "gemm_conv_fat_kernel": "bg_conv2d_fwd+bg_conv2d_bwd_data (fat tiles: one 384/256-row tile per CU)",
    "wgrad_gang_kernel": "bg_conv2d_bwd_weight_grouped",
    "gemm_conv_kernel": "bg_conv2d_fwd+bg_conv2d_bwd_data",
    "gemm_conv_dma_kernel": "bg_conv2d_fwd+bg_conv2d_bwd_data (64x64-per-wave tiles)",
    "wgrad_kernel": "bg_conv2d_bwd_weight",
    "dgrad_s2_smallc_kernel": "bg_conv2d_bwd_data(first layer: 3x3 stride 2, <= 16 channels)",
    "dw_fork_bwd_kernel": "bg_dwconv3x3_bwd_fork (one pass: skip gradient + depthwise data / weight gradient + the producer's BatchNorm-backward statistics)",
    "dw_bwd_fused_kernel": "bg_dwconv3x3_bwd_fused (one pass: data gradient + weight gradient + BatchNorm-backward statistics)",
    "dw_ring_kernel": "bg_dwconv3x3_fwd(+_pre)+bg_dwconv3x3_bwd_data(stride 1, dilation 1/2: LDS prefetch ring)",
    "dw_s1_kernel": "bg_dwconv3x3_fwd+bg_dwconv3x3_bwd_data(stride 1, dilation 1)",
    "dw_fwd": "bg_dwconv3x3_fwd(+stride-1 bwd_data, other strides/dilations)",
    "dw_bwd_data": "bg_dwconv3x3_bwd_data(stride 2)",
    "dw_bwd_weight": "bg_dwconv3x3_bwd_weight",
    "norm_act_fwd": "bg_norm_act_fwd",
    "norm_act_bwd_apply": "bg_norm_act_bwd_apply",
    "colreduce": "bg_norm_stats+bg_norm_act_bwd_reduce+bg_colsum",
}


def fold(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            fam = next((v for k, v in FAMILY.items() if k in name), None)
            if fam is None:
                continue
            a = agg[fam]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return agg


fetch, write = fold(sys.argv[1]), fold(sys.argv[2])
out = {}
for fam in fetch:
    n = fetch[fam][0]
    rd = 2.0 * fetch[fam][1] * 1024.0          # gfx950: FETCH_SIZE counts 128-B requests at 64 B
    wr = write.get(fam, [0, 0.0])[1] * 1024.0
    out[fam] = {"launches": n, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
                "hbm_bytes_per_launch": (rd + wr) / n}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1, "
                     "1152x768x16 batch 8 bf16; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
    print(f"{k:50s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch "
          f"(read {v['read_bytes_per_launch'] / 1e6:8.2f}, write {v['write_bytes_per_launch'] / 1e6:8.2f})")
