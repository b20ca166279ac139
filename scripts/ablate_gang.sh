# Ablation builds of the grouped weight-gradient kernel (not shipped): what its K loop costs without one of its parts.
set -e
cd "$(dirname "$0")/../bias-gan_amd/csrc"
mkdir -p /tmp/abl ../../abl_build
for v in ${ABL_VARIANTS:-G_SKIP_DMA G_NO_DMA G_HOT G_NO_FRAG G_NO_MFMA}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DABL_$v -c igemm_conv.hip -o /tmp/abl/igemm_$v.o 2>/dev/null &
done
wait
for v in ${ABL_VARIANTS:-G_SKIP_DMA G_NO_DMA G_HOT G_NO_FRAG G_NO_MFMA}; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 api.o /tmp/abl/igemm_$v.o fp8_conv.o dwconv.o dw_fused_bwd.o norm_act.o resample.o head_loss.o optim.o staging_ring.o volume.o partial.o -lpthread -o ../../abl_build/libbgamd_$v.so
done
ls -la ../../abl_build/libbgamd_*.so
