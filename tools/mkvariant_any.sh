# usage: bash tools/mkvariant_any.sh <name> <unit> [extra hipcc flags...]   unit = meshlet_cull | entity_cull | light_cluster | ...
# Compiles orbit_amd/csrc/<unit>.hip with the extra flags into a scratch object and links tools/variants/<name>.so from
# it and the current objects of the other translation units.  The product library and its objects are NOT touched.
name=$1; unit=$2; shift 2
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
make -C orbit_amd/csrc -s -j4 || exit 1
mkdir -p tools/variants
/opt/rocm/bin/hipcc $flags "$@" -c orbit_amd/csrc/$unit.hip -o tools/variants/$name.o || exit 1
others=""
for u in abi scan entity_cull meshlet_cull depth_reduce light_cluster; do [ $u != $unit ] && others="$others orbit_amd/csrc/$u.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/$name.so tools/variants/$name.o $others || exit 1
rm -f tools/variants/$name.o
echo "built tools/variants/$name.so"
