# usage: bash tools/mkvariant_any.sh <name> <unit>[,<unit>...] [extra hipcc flags...]
#        unit = meshlet_eval | meshlet_emit | meshlet_lists | meshlet_stream | meshlet_cull | cull_fused | entity_cull | ...
# Compiles orbit_amd/csrc/<unit>.hip with the extra flags into scratch objects and links tools/variants/<name>.so from
# them and the current objects of the other translation units (the list is the Makefile's SRCS).  The product library
# and its objects are NOT touched.  The tuning macros of the meshlet path (meshlet_common.h) are read by meshlet_eval
# (ORBIT_EVAL_LOAD_AUX, ORBIT_EV_WPS0) and meshlet_emit (ORBIT_EMIT_CNT_AUX, ORBIT_EMIT_STORE_AUX).
name=$1; units=$2; shift 2
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
make -C orbit_amd/csrc -s -j4 || exit 1
mkdir -p tools/variants
all=$(make -C orbit_amd/csrc -s print-units)
objs=""
for u in $all; do
  case ",$units," in
    *,$u,*) /opt/rocm/bin/hipcc $flags "$@" -c orbit_amd/csrc/$u.hip -o tools/variants/${name}_$u.o || exit 1
            objs="$objs tools/variants/${name}_$u.o" ;;
    *) objs="$objs orbit_amd/csrc/$u.o" ;;
  esac
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/$name.so $objs || exit 1
rm -f tools/variants/${name}_*.o
echo "built tools/variants/$name.so"
