# Build ablation variants of the library (not shipped): K loop without MFMAs, without loads, with hot loads.
set -e
cd "$(dirname "$0")/../bias-gan_amd/csrc"
mkdir -p /tmp/abl ../../abl_build
for v in ${ABL_VARIANTS:-NO_MMA NO_LOAD HOT_LOAD}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DABL_$v -c igemm_conv.hip -o /tmp/abl/igemm_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 api.o /tmp/abl/igemm_$v.o dwconv.o norm_act.o resample.o head_loss.o optim.o staging_ring.o volume.o partial.o -lpthread -o ../../abl_build/libbgamd_$v.so
done
ls -la ../../abl_build/libbgamd_*.so
