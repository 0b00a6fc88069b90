# usage: bash tools/mkvariant.sh <name> [extra hipcc flags...]   (run here; builds a library variant for tools/ab*.sh)
# Compiles meshlet_cull.hip with the extra flags into a scratch object and links tools/variants/<name>.so from it and
# the current objects of the other translation units.  orbit_amd/liborbit_cull.so and its objects are NOT touched.
name=$1; shift
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
make -C orbit_amd/csrc -s -j4 || exit 1
mkdir -p tools/variants
/opt/rocm/bin/hipcc $flags "$@" -c orbit_amd/csrc/meshlet_cull.hip -o tools/variants/$name.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/$name.so tools/variants/$name.o \
    orbit_amd/csrc/abi.o orbit_amd/csrc/scan.o orbit_amd/csrc/entity_cull.o orbit_amd/csrc/depth_reduce.o orbit_amd/csrc/light_cluster.o || exit 1
rm -f tools/variants/$name.o
echo "built tools/variants/$name.so"
