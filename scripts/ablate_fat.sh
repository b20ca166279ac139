# Ablation build of the fat-tile GEMM (not shipped): ABL_FAT_NO_A = no A-operand DMA and no A fragment reads from LDS.
set -e
cd "$(dirname "$0")/../bias-gan_amd/csrc"
mkdir -p /tmp/abl ../../abl_build
for v in ${ABL_VARIANTS:-FAT_NO_A}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DABL_$v -c igemm_conv.hip -o /tmp/abl/igemm_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 api.o /tmp/abl/igemm_$v.o fp8_conv.o dwconv.o dw_fused_bwd.o norm_act.o resample.o head_loss.o optim.o staging_ring.o volume.o partial.o -lpthread -o ../../abl_build/libbgamd_$v.so
done
ls -la ../../abl_build/libbgamd_*.so
