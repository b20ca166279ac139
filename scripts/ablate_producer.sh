# Ablation builds of the fat-tile GEMM with a synthetic in-kernel depthwise PRODUCER beside the MFMAs (igemm_fat.h,
# ABL_FAT_PRODUCER = vector instructions and ABL_FAT_PRODUCER_LDS = 16-byte LDS reads per MFMA group; 12 groups per 64
# reduction channels).  The real producer of the fused [BatchNorm + LeakyReLU -> depthwise -> pointwise] unit needs about
# 530 vector instructions and 48 LDS reads per thread and 64 channels = 44 + 4 per group.  Not shipped.
set -e
cd "$(dirname "$0")/../bias-gan_amd/csrc"
mkdir -p /tmp/abl ../../abl_build
for v in ${ABL_VARIANTS:-8:1 24:2 40:0 0:4 40:4 64:4}; do
  p=${v%%:*}; l=${v##*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DABL_FAT_PRODUCER=$p -DABL_FAT_PRODUCER_LDS=$l -c igemm_conv.hip -o /tmp/abl/igemm_P${p}L${l}.o 2>/dev/null &
done
wait
for v in ${ABL_VARIANTS:-8:1 24:2 40:0 0:4 40:4 64:4}; do
  p=${v%%:*}; l=${v##*:}
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 api.o /tmp/abl/igemm_P${p}L${l}.o fp8_conv.o dwconv.o dw_fused_bwd.o norm_act.o resample.o head_loss.o optim.o staging_ring.o volume.o partial.o -lpthread -o ../../abl_build/libbgamd_P${p}L${l}.so
done
ls -la ../../abl_build/libbgamd_P*.so
