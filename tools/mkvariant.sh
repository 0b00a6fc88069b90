# usage: bash tools/mkvariant.sh <name> [extra hipcc flags...]   (run here; builds a library variant for tools/ab.sh)
# Only meshlet_cull.o is rebuilt with the extra flags.
name=$1; shift
base="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
rm -f orbit_amd/csrc/meshlet_cull.o
make -C orbit_amd/csrc -s -j4 HIPFLAGS="$base $*" || exit 1
cp orbit_amd/liborbit_cull.so tools/variants/$name.so
echo "built tools/variants/$name.so"
