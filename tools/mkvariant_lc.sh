# usage: bash tools/mkvariant_lc.sh <name> [extra hipcc flags...]   (like tools/mkvariant.sh, for light_cluster.hip)
name=$1; shift
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
make -C orbit_amd/csrc -s -j4 || exit 1
mkdir -p tools/variants
/opt/rocm/bin/hipcc $flags "$@" -c orbit_amd/csrc/light_cluster.hip -o tools/variants/$name.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/$name.so tools/variants/$name.o \
    orbit_amd/csrc/abi.o orbit_amd/csrc/scan.o orbit_amd/csrc/entity_cull.o orbit_amd/csrc/depth_reduce.o orbit_amd/csrc/meshlet_cull.o || exit 1
rm -f tools/variants/$name.o
echo "built tools/variants/$name.so"
