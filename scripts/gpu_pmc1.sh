R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr "\n" " " | head -c 6000 > $R/gpurun_out/sq_counters.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_conv_$tag -o c -- python3 $R/scripts/one_conv.py > /dev/null 2> $R/gpurun_out/pmc_conv_$tag.err || tail -3 $R/gpurun_out/pmc_conv_$tag.err
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_conv_*/c_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        if "gemm_conv" in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, (n, v) in agg.items(): print(f"{k:28s} {v / n:14.1f} per launch ({n} launches)")
PY
